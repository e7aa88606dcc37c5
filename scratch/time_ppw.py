"""Cycle breakdown of conv_ppw.hip (TIMING instantiation: SP_CONV_PP_PRIO bit 2, SP_CONV_PPW = 2)."""
import sys; sys.path.insert(0, '/root/repo')
import ctypes, torch
from semantic_pyramid_for_image_generation_amd import ops, _lib as L
lib = L.lib(); dt = torch.bfloat16
names = ["L_A reads+requests", "L_B reads+vmcnt+lgkm", "lgkm(L_A)+barriers after L", "mfma segments", "barriers after M", "epilogue+switch"]
for cin, cout, hw, B in [(128, 128, 128, 40), (256, 256, 64, 40), (512, 512, 32, 20)]:
    x = ops.nhwc_empty(B, cin, hw, hw, dt, 'cuda'); x.normal_()
    w = (torch.randn(cout * 9 * cin, device='cuda') * 0.05).to(dt)
    bias = torch.randn(cout, device='cuda'); y = ops.nhwc_empty(B, cout, hw, hw, dt, 'cuda')
    ws = torch.zeros(256 * 8 * 16, device='cuda')
    lib.sp_set_tuning(22, 1 | 4); lib.sp_set_tuning(26, 2)
    p = L.SpConvParams()
    p.x, p.w, p.bias, p.y = x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr()
    p.n, p.h, p.w_, p.cin_p, p.cout, p.ldy, p.ksize, p.act, p.dtype = B, hw, hw, cin, cout, cout, 3, 1, 1
    p.workspace, p.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    for _ in range(3):
        L.call("sp_conv2d_igemm", ctypes.byref(p), ops.stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); L.call("sp_conv2d_igemm", ctypes.byref(p), ops.stream()); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    t = ws.view(256, 8, 16)[:, :, :6]
    items = B * (hw // 16) * (hw // 32) * ((cout + 127) // 128) / 256
    stages = items * ((cin + 31) // 32) * 3
    for half, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        m = t[:, sl, :].mean(dim=(0, 1)); tot = m.sum().item()
        print("%d->%d @%d N=%d %s: total %.0f cycles | " % (cin, cout, hw, B, half, tot) + " | ".join("%s %.0f/stage (%.1f%%)" % (n, v / stages, 100 * v / tot) for n, v in zip(names, m.tolist())))
    cyc = t.sum(dim=2).mean().item()
    print("   stages per block %.1f -> cycles per stage %.0f; launch %.1f us = %.0f TFLOP/s -> shader clock under this load %.2f GHz, matrix pipe busy %.0f %% of the cycles" % (
        stages, cyc / stages, us, 2.0 * B * hw * hw * cin * cout * 9 / us / 1e6, cyc / us / 1e3, 100 * 2 * t[:, :, 3].mean().item() / cyc))
lib.sp_set_tuning(22, -1); lib.sp_set_tuning(26, -1)
