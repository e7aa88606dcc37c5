"""Randomised shapes across ALL convolution routes (bf16): 3x3 / 1x1 forward on any map size (small-spatial split-K igemm, 8-channel
inputs, thin outputs, tall / ping-pong kernels) and the weight gradient (per-tap, narrow-map row walker, streaming 1x1, row walker),
against fp32 torch references.  Usage: stress_conv_all.py [seed]"""
import sys, ctypes, random
sys.path.insert(0, '.')
import torch, torch.nn.functional as F
from semantic_pyramid_for_image_generation_amd import ops, _lib as L
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == 'f32') else torch.bfloat16
SPD = L.SP_F32 if dt == torch.float32 else L.SP_BF16
TOLF, TOLW = (3e-4, 3e-4) if dt == torch.float32 else (8e-3, 2e-3)
PAD = 4 if dt == torch.float32 else 8
random.seed(seed); torch.manual_seed(seed)
worst = {}
def note(kind, err, shape):
    if err > worst.get(kind, (0, None))[0]: worst[kind] = (err, shape)
def fwd(cin, cout, k, n, h, w, act, res):
    x = ops.nhwc_empty(n, cin, h, w, dt, 'cuda').normal_()
    wt = (torch.randn(cout, k, k, cin, device='cuda') * 0.05).to(dt)
    bias = torch.randn(cout, device='cuda')
    ldy = (cout + PAD - 1) // PAD * PAD
    y = ops.nhwc_empty(n, ldy, h, w, dt, 'cuda').fill_(-7.0)
    r1 = ops.nhwc_empty(n, ldy, h, w, dt, 'cuda').normal_() if res else None
    ops.conv_launch(x, wt.data_ptr(), bias, y, r1, None, None, 0.0, n, h, w, cin, cout, ldy, k, act, dt)
    ref = F.conv2d(x.float(), wt.float().permute(0, 3, 1, 2), bias, padding=k // 2)
    if res: ref = ref + r1.float()[:, :cout]
    if act == 1: ref = F.leaky_relu(ref, 0.2)
    return float((y[:, :cout].float() - ref).abs().max() / ref.abs().max())
def wgrad(cin, cout, k, n, h, w):
    x = ops.nhwc_empty(n, cin, h, w, dt, 'cuda').normal_()
    cp = (cout + PAD - 1) // PAD * PAD
    dy = ops.nhwc_empty(n, cp, h, w, dt, 'cuda').normal_()
    ndw = cout * k * k * cin
    buf = torch.zeros(ndw + cout + 8, dtype=torch.float32, device='cuda')
    wsf = ops.wgrad_workspace_floats(n, h, w, cin, cout, k, dt)
    ws = torch.empty(max(wsf, 1), dtype=torch.float32, device='cuda')
    L.call("sp_conv2d_wgrad_accum", ops.ptr(x), ops.ptr(dy), ops.ptr(buf), ctypes.c_void_p(buf.data_ptr() + 4 * (ndw + 4)),
           ops.ptr(ws) if wsf else None, wsf, n, h, w, cin, cout, cp, k, SPD, ops.stream())
    dyf = dy.float()[:, :cout].contiguous()
    wt = torch.zeros(cout, cin, k, k, device='cuda', requires_grad=True)
    F.conv2d(x.float(), wt, padding=k // 2).backward(dyf)
    ref = wt.grad.permute(0, 2, 3, 1).reshape(-1)
    e = float((buf[:ndw] - ref).abs().max() / ref.abs().max())
    bref = dyf.sum((0, 2, 3))
    return max(e, float((buf[ndw + 4:ndw + 4 + cout] - bref).abs().max() / bref.abs().max().clamp_min(1e-6)))
CH_IN = [8, 16, 32, 40, 64, 72, 128, 136, 256, 264, 512, 520, 768]
CH_OUT = [3, 8, 16, 24, 32, 40, 64, 72, 128, 136, 256, 512, 768]
SZ = [1, 2, 3, 4, 5, 7, 8, 12, 16, 24, 32, 40, 64, 96, 128]
for it in range(70):
    k = random.choice([1, 3, 3]); cin = random.choice(CH_IN); cout = random.choice(CH_OUT)
    h = random.choice(SZ); w = random.choice(SZ); n = random.randint(1, 5)
    if n * h * w * max(cin, cout) > 3e7: continue
    act = random.choice([0, 1]); res = random.random() < 0.3
    e = fwd(cin, cout, k, n, h, w, act, res)
    note("fwd %dx%d" % (k, k), e, (cin, cout, n, h, w, act, res))
    if e > TOLF: print("FAIL fwd", (k, cin, cout, n, h, w, act, res), e, flush=True)
for it in range(50):
    k = random.choice([1, 3, 3]); cin = random.choice(CH_IN); cout = random.choice(CH_OUT)
    h = random.choice(SZ); w = random.choice(SZ); n = random.randint(1, 5)
    if n * h * w * max(cin, cout) > 3e7: continue
    e = wgrad(cin, cout, k, n, h, w)
    note("wgrad %dx%d" % (k, k), e, (cin, cout, n, h, w))
    if e > TOLW: print("FAIL wgrad", (k, cin, cout, n, h, w), e, flush=True)
for kk, (e, sh) in sorted(worst.items()): print("%-12s worst rel err %.2e at %s" % (kk, e, sh))
print("seed %d done" % seed)
