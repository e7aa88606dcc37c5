"""Randomised shapes through the kernels added in the second half of round 2, against fp32 torch references built from the same
bf16-rounded operands: 1x1 forward (direct / split-K), 1x1 weight gradient (streaming), 3x3 on 8-channel input (forward + weight
gradient), 3x3 with <= 4 outputs."""
import sys, ctypes, random
sys.path.insert(0, '.')
import torch
import torch.nn.functional as F
from semantic_pyramid_for_image_generation_amd import ops, _lib as L
dt = torch.bfloat16
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = {}
def note(kind, err, shape):
    if err > worst.get(kind, (0, None))[0]: worst[kind] = (err, shape)
def fwd(cin, cout, k, n, h, w):
    x = ops.nhwc_empty(n, cin, h, w, dt, 'cuda'); x.normal_()
    wt = (torch.randn(cout, k, k, cin, device='cuda') * 0.05).to(dt)
    bias = torch.randn(cout, device='cuda')
    ldy = (cout + 7) // 8 * 8
    y = ops.nhwc_empty(n, ldy, h, w, dt, 'cuda'); y.zero_()
    ops.conv_launch(x, wt.data_ptr(), bias, y, None, None, None, 0.0, n, h, w, cin, cout, ldy, k, 0, dt)
    ref = F.conv2d(x.float().contiguous(), wt.float().permute(0, 3, 1, 2).contiguous(), bias, padding=k // 2)
    got = y[:, :cout].float()
    return float((got - ref).abs().max() / ref.abs().max())
def wgrad(cin, cout, k, n, h, w):
    x = ops.nhwc_empty(n, cin, h, w, dt, 'cuda'); x.normal_()
    cp = (cout + 7) // 8 * 8
    dy = ops.nhwc_empty(n, cp, h, w, dt, 'cuda'); dy.normal_()
    ndw = cout * k * k * cin
    buf = torch.zeros(ndw + cout + 8, dtype=torch.float32, device='cuda')
    wsf = ops.wgrad_workspace_floats(n, h, w, cin, cout, k, dt)
    ws = torch.empty(max(wsf, 1), dtype=torch.float32, device='cuda')
    L.call("sp_conv2d_wgrad_accum", ops.ptr(x), ops.ptr(dy), ops.ptr(buf), ctypes.c_void_p(buf.data_ptr() + 4 * (ndw + 4)),
           ops.ptr(ws) if wsf else None, wsf, n, h, w, cin, cout, cp, k, L.SP_BF16, ops.stream())
    xf, dyf = x.float().contiguous().requires_grad_(False), dy[:, :cout].float().contiguous()
    wt = torch.zeros(cout, cin, k, k, device='cuda', requires_grad=True)
    F.conv2d(xf, wt, None, padding=k // 2).backward(dyf)
    ref = wt.grad.permute(0, 2, 3, 1).reshape(cout, k * k * cin)          # [co][tap][ci]
    got = buf[:ndw].view(cout, k * k * cin)
    e = float((got - ref).abs().max() / ref.abs().max())
    bref = dyf.sum((0, 2, 3))
    be = float((buf[ndw + 4:ndw + 4 + cout] - bref).abs().max() / bref.abs().max())
    return max(e, be)
for it in range(40):
    n = random.choice([1, 2, 3, 5, 20])
    h = random.choice([1, 2, 3, 4, 6, 8, 16, 24, 32, 64]); w = random.choice([1, 2, 4, 5, 8, 16, 24, 32, 64])
    cin = random.choice([8, 32, 64, 72, 128, 136, 256, 264, 512, 520, 768]); cout = random.choice([3, 8, 30, 32, 64, 72, 128, 136, 256, 512])
    note("fwd 1x1", fwd(cin, cout, 1, n, h, w), (cin, cout, n, h, w))
    note("wgrad 1x1", wgrad(cin, cout, 1, n, h, w), (cin, cout, n, h, w))
for it in range(12):
    n = random.choice([1, 2, 4]); h = random.choice([16, 32, 64, 128]); w = random.choice([32, 64, 128, 256])
    cout = random.choice([16, 32, 48, 64])
    note("fwd 3x3 cin8", fwd(8, cout, 3, n, h, w), (8, cout, n, h, w))
    note("wgrad 3x3 cin8", wgrad(8, random.choice([16, 48, 64, 72, 128]), 3, n, h, w), (n, h, w))
    note("fwd 3x3 thin cout", fwd(random.choice([32, 64]), random.choice([1, 3, 4]), 3, n, random.choice([8, 16, 64]), w), (n, h, w))
torch.cuda.synchronize()
for k, (e, s) in worst.items(): print("%-20s worst relative error %.2e at %s" % (k, e, s))
assert all(e < 2e-2 for e, _ in worst.values())
print("ok")
