"""Cycle stamps of conv1x1_direct_kernel (debug build with SP_DBG1X1=16): fill, barrier, K loop, epilogue, per wave."""
import sys, os, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
from semantic_pyramid_for_image_generation_amd import ops, _lib as L
B = 20
dt = torch.bfloat16
for cin, cout, hw in [(512, 512, 4), (256, 256, 16), (256, 128, 32), (128, 256, 32), (768, 512, 2), (32, 256, 16)]:
    x = ops.nhwc_empty(B, cin, hw, hw, dt, 'cuda'); x.normal_()
    w = (torch.randn(cout * cin, device='cuda') * 0.05).to(dt)
    bias = torch.randn(cout, device='cuda')
    y = ops.nhwc_empty(B, cout, hw, hw, dt, 'cuda')
    dbg = torch.zeros(1024 * 4 * 8, dtype=torch.int64, device='cuda')
    def launch(ws):
        p = L.SpConvParams()
        p.x, p.w, p.bias, p.y = x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr()
        p.n, p.h, p.w_, p.cin_p, p.cout, p.ldy, p.ksize, p.act, p.dtype = B, hw, hw, cin, cout, cout, 1, 0, L.SP_BF16
        if ws is not None:
            p.workspace, p.workspace_bytes = ws.data_ptr(), ws.numel() * 8
        L.call("sp_conv2d_igemm", ctypes.byref(p), ops.stream())
    for _ in range(3): launch(None)
    launch(dbg)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(-1, 8).astype(np.float64)
    d = d[d[:, 5] > 0]
    m = d.mean(0)
    print("%4d->%4d @%3d: %4d waves; cycles: fill %.0f, barrier %.0f, K loop %.0f, epilogue %.0f, total %.0f (max %.0f)" %
          (cin, cout, hw, len(d), m[0], m[1], m[2], m[3], m[4], d[:, 4].max()))
