"""Cycle breakdown of conv_wgrad_pp3_kernel (TIMING instantiation: SP_WGRAD_PP=3)."""
import sys, ctypes; sys.path.insert(0, '.')
import torch
from semantic_pyramid_for_image_generation_amd import ops, _lib as L
lib = L.lib(); dt = torch.bfloat16; B = 20
names = ["LOAD (reads + requests)", "vmcnt wait", "lgkm wait", "barrier after L", "MFMA segment", "barrier after M"]
for cin, cout, hw in [(256, 256, 64), (128, 128, 128), (64, 64, 256)]:
    x = ops.nhwc_empty(B, cin, hw, hw, dt, 'cuda').normal_(); dy = ops.nhwc_empty(B, cout, hw, hw, dt, 'cuda').normal_()
    ndw = cout * 9 * cin
    buf = torch.zeros(ndw + cout + 8, dtype=torch.float32, device='cuda')
    wsf = ops.wgrad_workspace_floats(B, hw, hw, cin, cout, 3, dt); ws = torch.zeros(max(wsf, 1), dtype=torch.float32, device='cuda')
    lib.sp_set_tuning(23, 3)
    for _ in range(2):
        L.call("sp_conv2d_wgrad_accum", ops.ptr(x), ops.ptr(dy), ops.ptr(buf), ctypes.c_void_p(buf.data_ptr() + 4 * (ndw + 4)), ops.ptr(ws), wsf, B, hw, hw, cin, cout, cout, 3, L.SP_BF16, ops.stream())
    torch.cuda.synchronize(); lib.sp_set_tuning(23, -1)
    t = ws[:256 * 8 * 8].view(256, 8, 8)
    for half, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        m = t[:, sl, :6].mean(dim=(0, 1)); steps = t[:, sl, 6].mean().item(); tot = m.sum().item()
        print("%d->%d @%d %s: %.0f cycles over %.1f steps = %.0f per step | " % (cin, cout, hw, half, tot, steps, tot / max(steps, 1)) + " | ".join("%s %.0f" % (n, v / max(steps, 1)) for n, v in zip(names, m.tolist())))
