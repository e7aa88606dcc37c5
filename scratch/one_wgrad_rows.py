import sys, ctypes; sys.path.insert(0, "/root/repo")
import torch
from semantic_pyramid_for_image_generation_amd import ops, _lib as L
cin, cout, hw, k, B = [int(a) for a in sys.argv[1:6]]
dt = torch.bfloat16
x = ops.nhwc_empty(B, cin, hw, hw, dt, 'cuda'); x.normal_()
dy = ops.nhwc_empty(B, cout, hw, hw, dt, 'cuda'); dy.normal_()
ndw = cout * k * k * cin
buf = torch.zeros(ndw + cout + 8, dtype=torch.float32, device='cuda')
wsf = ops.wgrad_workspace_floats(B, hw, hw, cin, cout, k, dt)
ws = torch.empty(max(wsf, 1), dtype=torch.float32, device='cuda')
for _ in range(3):
    L.call("sp_conv2d_wgrad_accum", ops.ptr(x), ops.ptr(dy), ops.ptr(buf), ctypes.c_void_p(buf.data_ptr() + 4 * (ndw + 4)), ops.ptr(ws) if wsf else None, wsf,
           B, hw, hw, cin, cout, cout, k, L.SP_BF16, ops.stream())
torch.cuda.synchronize()
