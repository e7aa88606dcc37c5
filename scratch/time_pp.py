"""Cycle breakdown of the ping-pong kernel (TIMING instantiation: SP_CONV_PP_PRIO bit 2)."""
import sys; sys.path.insert(0, '.')
import ctypes, torch
from semantic_pyramid_for_image_generation_amd import ops, _lib as L
lib = L.lib(); dt = torch.bfloat16; B = 20
names = ["frag reads issue", "vmcnt wait", "barrier after L", "mfma segment", "barrier after M", "epilogue+init", "dma issue", "lgkm wait"]
for cin, cout, hw in [(64, 64, 256), (128, 64, 256), (128, 128, 128), (256, 256, 64), (64, 128, 128)]:
    x = ops.nhwc_empty(B, cin, hw, hw, dt, 'cuda'); x.normal_()
    w = (torch.randn(cout * 9 * cin, device='cuda') * 0.05).to(dt)
    bias = torch.randn(cout, device='cuda'); y = ops.nhwc_empty(B, cout, hw, hw, dt, 'cuda')
    ws = torch.zeros(256 * 8 * 16, device='cuda')
    lib.sp_set_tuning(21, 8); lib.sp_set_tuning(26, 0); lib.sp_set_tuning(22, int(sys.argv[1]) if len(sys.argv) > 1 else 5)
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
    t16 = ws.view(256, 8, 16); t = t16[:, :, :8]
    for half, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        m = t[:, sl, :].mean(dim=(0, 1)); tot = m.sum().item()
        print("%d->%d @%d %s: total %.0f cycles | " % (cin, cout, hw, half, tot) + " | ".join("%s %.1f%%" % (n, 100 * v / tot) for n, v in zip(names, m.tolist())))
    e = t16[:, :, 8:12].mean(dim=(0, 1)).tolist(); th = 16 if cout <= 64 else 8; items = B * (hw // th) * (hw // 32) * ((cout + 127) // 128) / 256
    print("   per item: acc init %.0f | item-end setup %.0f | fragments %.0f | next coordinates %.0f cycles" % tuple(v / items for v in e))
    stages = items * ((cin + 31) // 32) * 3
    if (int(sys.argv[1]) if len(sys.argv) > 1 else 5) & 512:
        mid = items * max((cin + 31) // 32 - 2, 0) * 3
        m = t.mean(dim=(0, 1)).tolist()
        print("   MID chunks only (%d stages): per stage " % mid + " | ".join("%s %.0f" % (n, v / max(mid, 1)) for n, v in zip(names, m)))
    cyc = t.sum(dim=2).mean().item()
    print("   stages per block %.1f -> cycles per stage %.0f; launch %.1f us = %.0f TFLOP/s -> shader clock under this load %.2f GHz, matrix pipe busy %.0f %% of the cycles" % (
        stages, cyc / stages, us, 2.0 * B * hw * hw * cin * cout * 9 / us / 1e6, cyc / us / 1e3, 100 * 2 * t[:, :, 3].mean().item() / cyc))
lib.sp_set_tuning(21, -1); lib.sp_set_tuning(22, -1); lib.sp_set_tuning(26, -1)
