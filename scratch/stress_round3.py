"""Randomised shapes through the kernels of round 3 against fp32 torch references on the same bf16 operands: 3x3 forward on the
ping-pong kernel (32-wide tiles with Cout > 64 / <= 64, 16-wide tiles; bias, activation, mask, one / two residuals, pooling) and the
3x3 weight gradient (register-carried rows; odd row ranges, several columns per block, pooled gradients).  Usage: stress_round3.py [seed]"""
import sys, ctypes, random
sys.path.insert(0, '.')
import torch, torch.nn.functional as F
from semantic_pyramid_for_image_generation_amd import ops, _lib as L
dt = torch.bfloat16
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
random.seed(seed); torch.manual_seed(seed)
worst = {}
def note(kind, err, shape):
    if err > worst.get(kind, (0, None))[0]: worst[kind] = (err, shape)
def conv(n, cin, cout, h, w, act, res, mask, pool2, bias, up=False):
    x = ops.nhwc_empty(n, cin, h // 2 if up else h, w // 2 if up else w, dt, 'cuda').normal_()
    wt = (torch.randn(cout, 3, 3, cin, device='cuda') * 0.05).to(dt)
    b = torch.randn(cout, device='cuda') if bias else None
    ho, wo = (h // 2, w // 2) if pool2 else (h, w)
    mk = lambda: ops.nhwc_empty(n, cout, ho, wo, dt, 'cuda').normal_()
    r1 = mk() if res >= 1 else None; r2 = mk() if res >= 2 else None; ms = mk() if mask else None
    y = ops.nhwc_empty(n, cout, ho, wo, dt, 'cuda').fill_(-7.0)
    ops._conv_launch(x, wt.data_ptr(), b, y, r1, r2, ms, 0.2, n, h, w, cin, cout, cout, 3, act, dt, pool2, up)
    xin = 0.25 * F.interpolate(x.float(), scale_factor=2, mode='nearest') if up else x.float()
    ref = F.conv2d(xin, wt.float().permute(0, 3, 1, 2), b, padding=1)
    if pool2 == 1: ref = F.avg_pool2d(ref, 2)
    elif pool2 == 2: ref = F.max_pool2d(ref, 2)
    if ms is not None: ref = ref * torch.where(ms.float() > 0, 1.0, 0.2)
    if r1 is not None: ref = ref + r1.float()
    if r2 is not None: ref = ref + r2.float()
    ref = {0: lambda t: t, 1: lambda t: F.leaky_relu(t, 0.2), 2: F.relu, 3: torch.tanh}[act](ref)
    return float((y.float() - ref).abs().max() / ref.abs().max())
def wgrad(n, cin, cout, h, w, up):
    x = ops.nhwc_empty(n, cin, h, w, dt, 'cuda').normal_()
    cp = (cout + 7) // 8 * 8
    hd, wd = (h // 2, w // 2) if up else (h, w)
    dy = ops.nhwc_empty(n, cp, hd, wd, dt, 'cuda').normal_()
    ndw = cout * 9 * cin
    buf = torch.zeros(ndw + cout + 8, dtype=torch.float32, device='cuda')
    wsf = ops.wgrad_workspace_floats(n, h, w, cin, cout, 3, dt)
    ws = torch.empty(max(wsf, 1), dtype=torch.float32, device='cuda')
    L.call("sp_conv2d_wgrad_accum_pooled" if up else "sp_conv2d_wgrad_accum", ops.ptr(x), ops.ptr(dy), ops.ptr(buf), ctypes.c_void_p(buf.data_ptr() + 4 * (ndw + 4)),
           ops.ptr(ws) if wsf else None, wsf, n, h, w, cin, cout, cp, 3, L.SP_BF16, ops.stream())
    dyf = dy.float()[:, :cout]
    if up: dyf = F.interpolate(dyf, scale_factor=2, mode='nearest') * 0.25
    wt = torch.zeros(cout, cin, 3, 3, device='cuda', requires_grad=True)
    F.conv2d(x.float(), wt, padding=1).backward(dyf.contiguous())
    ref = wt.grad.permute(0, 2, 3, 1).reshape(-1)
    e = float((buf[:ndw] - ref).abs().max() / ref.abs().max())
    bref = dyf.sum((0, 2, 3))
    eb = float((buf[ndw + 4:ndw + 4 + cout] - bref).abs().max() / bref.abs().max())
    return max(e, eb)
for it in range(60):
    kind = random.choice(["wide", "wide", "thin", "w16"])
    if kind == "w16":
        cout = random.choice([128, 256, 512, 192, 320]); cin = random.choice([32, 64, 72, 128, 264, 512])
        h = 16 * random.choice([1, 1, 2]); w = 16; pool2 = 0
        per = ((cout + 127) // 128) * (h // 16)
        n = (64 + per - 1) // per + random.randint(0, 6)
    else:
        cout = random.choice([128, 256, 136, 192, 80]) if kind == "wide" else random.choice([64, 32, 40, 24])
        cin = random.choice([32, 64, 72, 128, 136, 256])
        n = random.randint(1, 6); h = (8 if kind == "wide" else 16) * random.randint(1, 6); w = 32 * random.randint(1, 3)
        pool2 = random.choice([0, 0, 0, 1, 2]) if cout % 16 == 0 and cout > 32 and h % 16 == 0 else 0
    act = random.choice([0, 1, 2, 3] if pool2 == 0 else [0, 2]); res = random.choice([0, 0, 1, 2]) if pool2 != 2 else 0; mask = random.random() < 0.25 and pool2 == 0; bias = random.random() < 0.8
    up = kind != "w16" and cout > 32 and pool2 == 0 and random.random() < 0.2
    shape = (kind, n, cin, cout, h, w, act, res, mask, pool2, bias, up)
    e = conv(n, cin, cout, h, w, act, res, mask, pool2, bias, up)
    note("conv3x3 " + kind, e, shape)
    if e > 8e-3: print("FAIL conv", shape, e, flush=True)
for it in range(30):
    cin = random.choice([64, 72, 128, 136, 256]); cout = random.choice([64, 128, 136, 256, 40])
    n = random.randint(1, 7); h = 2 * random.randint(2, 40); w = 32 * random.randint(1, 3); up = random.random() < 0.25 and h % 2 == 0
    e = wgrad(n, cin, cout, h, w, up)
    note("wgrad3x3", e, (n, cin, cout, h, w, up))
    if e > 2e-3: print("FAIL wgrad", (n, cin, cout, h, w, up), e, flush=True)
for k, (e, sh) in sorted(worst.items()): print("%-16s worst rel err %.2e at %s" % (k, e, sh))
print("seed %d done" % seed)
