"""Randomised-shape stress of every kernel route against fp32 torch references on the same operands (round-3 VERDICT, missing #5:
the harness that found three real bugs lived in scratch/ and nobody would ever run it again).  Fixed seeds, both storage types,
partial channel tiles (Cout 80, 192, 136 ...), odd maps, every epilogue operand - and every case is launched TWICE into fresh
outputs (the round-3 register race only showed on the second and later launches of a process).  ~45 s on one MI355X.

Routes covered: 3x3 / 1x1 forward on any map size (small-spatial split-K igemm, 8-channel inputs, thin outputs, tall and ping-pong
kernels in their three tile forms, pooled / up-sampled-gradient variants), the weight gradient (per-tap, narrow-map row walker,
streaming 1x1, register-carried row walker, pooled gradients), BatchNorm plain / conditional / fused with the bilinear pass,
bilinear forward / backward, the pooling family, the attention core."""
import ctypes
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from semantic_pyramid_for_image_generation_amd import _lib as L  # noqa: E402
from semantic_pyramid_for_image_generation_amd import ops  # noqa: E402


@pytest.fixture(autouse=True)
def _dtype_reset():
    yield
    ops.set_compute_dtype(torch.float32)


def _rel(a, b):
    return float((a.detach() - b.detach()).abs().max() / b.detach().abs().max().clamp_min(1e-6))


def _seed(seed):
    random.seed(seed)
    torch.manual_seed(seed)


_ACT = {0: lambda t: t, 1: lambda t: F.leaky_relu(t, 0.2), 2: F.relu, 3: torch.tanh}


def _conv_case(dt, n, cin, cout, k, h, w, act, res, mask, pool2, bias, up, ldy=None):
    """One forward launch (twice) of sp_conv2d_igemm vs F.conv2d on the same operands; returns the relative error."""
    x = ops.nhwc_empty(n, cin, h // 2 if up else h, w // 2 if up else w, dt, 'cuda').normal_()
    wt = (torch.randn(cout, k, k, cin, device='cuda') * 0.05).to(dt)
    b = torch.randn(cout, device='cuda') if bias else None
    ho, wo = (h // 2, w // 2) if pool2 else (h, w)
    ldy = cout if ldy is None else ldy
    mk = lambda: ops.nhwc_empty(n, ldy, ho, wo, dt, 'cuda').normal_()          # noqa: E731
    r1 = mk() if res >= 1 else None
    r2 = mk() if res >= 2 else None
    ms = mk() if mask else None
    xin = 0.25 * F.interpolate(x.float(), scale_factor=2, mode='nearest') if up else x.float()
    ref = F.conv2d(xin, wt.float().permute(0, 3, 1, 2), b, padding=k // 2)
    if pool2 == 1:
        ref = F.avg_pool2d(ref, 2)
    elif pool2 == 2:
        ref = F.max_pool2d(ref, 2)
    if ms is not None:
        ref = ref * torch.where(ms.float()[:, :cout] > 0, 1.0, 0.2)
    if r1 is not None:
        ref = ref + r1.float()[:, :cout]
    if r2 is not None:
        ref = ref + r2.float()[:, :cout]
    ref = _ACT[act](ref)
    worst = 0.0
    for _ in range(2):
        y = ops.nhwc_empty(n, ldy, ho, wo, dt, 'cuda').fill_(-7.0)
        ops._conv_launch(x, wt.data_ptr(), b, y, r1, r2, ms, 0.2, n, h, w, cin, cout, ldy, k, act, dt, pool2, up)
        worst = max(worst, _rel(y[:, :cout].float(), ref))
    return worst


def _wgrad_case(dt, n, cin, cout, k, h, w, up):
    spd = ops.sp_dtype(dt)
    pad = 4 if dt == torch.float32 else 8
    x = ops.nhwc_empty(n, cin, h, w, dt, 'cuda').normal_()
    cp = (cout + pad - 1) // pad * pad
    hd, wd = (h // 2, w // 2) if up else (h, w)
    dy = ops.nhwc_empty(n, cp, hd, wd, dt, 'cuda').normal_()
    ndw = cout * k * k * cin
    wsf = ops.wgrad_workspace_floats(n, h, w, cin, cout, k, dt)
    dyf = dy.float()[:, :cout]
    if up:
        dyf = F.interpolate(dyf, scale_factor=2, mode='nearest') * 0.25
    wt = torch.zeros(cout, cin, k, k, device='cuda', requires_grad=True)
    F.conv2d(x.float(), wt, padding=k // 2).backward(dyf.contiguous())
    ref = wt.grad.permute(0, 2, 3, 1).reshape(-1)
    bref = dyf.sum((0, 2, 3))
    worst = 0.0
    for _ in range(2):
        buf = torch.zeros(ndw + cout + 8, dtype=torch.float32, device='cuda')
        ws = torch.empty(max(wsf, 1), dtype=torch.float32, device='cuda')
        L.call("sp_conv2d_wgrad_accum_pooled" if up else "sp_conv2d_wgrad_accum", ops.ptr(x), ops.ptr(dy), ops.ptr(buf),
               ctypes.c_void_p(buf.data_ptr() + 4 * (ndw + 4)), ops.ptr(ws) if wsf else None, wsf, n, h, w, cin, cout, cp, k, spd, ops.stream())
        worst = max(worst, _rel(buf[:ndw], ref), _rel(buf[ndw + 4:ndw + 4 + cout], bref))
    return worst


CH_IN = [8, 16, 32, 40, 64, 72, 128, 136, 256, 264, 512, 520, 768]
CH_OUT = [3, 8, 16, 24, 32, 40, 64, 72, 80, 128, 136, 192, 256, 512, 768]
SZ = [1, 2, 3, 4, 5, 7, 8, 12, 16, 24, 32, 40, 64, 96, 128]


@pytest.mark.parametrize("dt,seed,tol_f,tol_w", [(torch.bfloat16, 0, 8e-3, 2e-3), (torch.bfloat16, 1, 8e-3, 2e-3), (torch.float32, 2, 3e-4, 3e-4)])
def test_every_convolution_route_random_shapes(dt, seed, tol_f, tol_w):
    """scratch/stress_conv_all.py: 3x3 / 1x1 forward and weight gradient on maps 1 x 1 ... 128 x 128, channels 3 ... 768."""
    _seed(seed)
    pad = 4 if dt == torch.float32 else 8
    fails = []
    for _ in range(36):
        k = random.choice([1, 3, 3])
        cin, cout = random.choice(CH_IN), random.choice(CH_OUT)
        if dt == torch.float32:
            cin = (cin + 3) // 4 * 4
        h, w, n = random.choice(SZ), random.choice(SZ), random.randint(1, 5)
        if n * h * w * max(cin, cout) > 2e7:
            continue
        act, res = random.choice([0, 1]), int(random.random() < 0.3)
        e = _conv_case(dt, n, cin, cout, k, h, w, act, res, False, 0, True, False, ldy=(cout + pad - 1) // pad * pad)
        if e > tol_f:
            fails.append(("fwd", k, cin, cout, n, h, w, act, res, e))
    for _ in range(26):
        k = random.choice([1, 3, 3])
        cin, cout = random.choice(CH_IN), random.choice(CH_OUT)
        h, w, n = random.choice(SZ), random.choice(SZ), random.randint(1, 5)
        if n * h * w * max(cin, cout) > 2e7:
            continue
        e = _wgrad_case(dt, n, cin, cout, k, h, w, False)
        if e > tol_w:
            fails.append(("wgrad", k, cin, cout, n, h, w, e))
    assert not fails, fails


@pytest.mark.parametrize("seed", [0, 3])
def test_ping_pong_kernels_random_shapes(seed):
    """scratch/stress_round3.py: the ping-pong 3x3 in its three tile forms with every epilogue operand, pooling, pooled-gradient
    inputs, partial channel tiles (Cout 80, 136, 192, 320); the register-carried weight gradient incl. pooled gradients."""
    _seed(seed)
    dt = torch.bfloat16
    fails = []
    for _ in range(40):
        kind = random.choice(["wide", "wide", "thin", "w16"])
        if kind == "w16":
            cout = random.choice([128, 256, 512, 192, 320])
            cin = random.choice([32, 64, 72, 128, 264, 512])
            h, w, pool2 = 16 * random.choice([1, 1, 2]), 16, 0
            per = ((cout + 127) // 128) * (h // 16)
            n = (64 + per - 1) // per + random.randint(0, 6)
        else:
            cout = random.choice([128, 256, 136, 192, 80]) if kind == "wide" else random.choice([64, 32, 40, 24])
            cin = random.choice([32, 64, 72, 128, 136, 256])
            n, h, w = random.randint(1, 6), (8 if kind == "wide" else 16) * random.randint(1, 6), 32 * random.randint(1, 3)
            pool2 = random.choice([0, 0, 0, 1, 2]) if cout % 16 == 0 and cout > 32 and h % 16 == 0 else 0
        act = random.choice([0, 1, 2, 3] if pool2 == 0 else [0, 2])
        res = random.choice([0, 0, 1, 2]) if pool2 != 2 else 0
        mask = random.random() < 0.25 and pool2 == 0
        bias = random.random() < 0.8
        up = kind != "w16" and cout > 32 and pool2 == 0 and random.random() < 0.2
        e = _conv_case(dt, n, cin, cout, 3, h, w, act, res, mask, pool2, bias, up)
        if e > 8e-3:
            fails.append(("conv", kind, n, cin, cout, h, w, act, res, mask, pool2, bias, up, e))
    for _ in range(20):
        cin, cout = random.choice([64, 72, 128, 136, 256]), random.choice([64, 128, 136, 256, 40])
        n, h, w = random.randint(1, 7), 2 * random.randint(2, 40), 32 * random.randint(1, 3)
        up = random.random() < 0.25
        e = _wgrad_case(dt, n, cin, cout, 3, h, w, up)
        if e > 2e-3:
            fails.append(("wgrad", n, cin, cout, h, w, up, e))
    assert not fails, fails


@pytest.mark.parametrize("dt,seed,ppw", [(torch.bfloat16, 0, 1), (torch.bfloat16, 7, 2), (torch.float16, 2, 1), (torch.float16, 4, 2)])
def test_tail_split_random_shapes_beyond_one_round(dt, seed, ppw):
    """Round 6: launches of MORE than one round of the 256 persistent blocks with a partial last round (the K-split of conv_pp.hip and,
    with SP_TUNE_CONV_PPW = 2, of conv_ppw.hip): random batch / map / channel counts with 257 ... 1200 work items, partial K chunks
    and channel tiles, every epilogue operand, both 16-bit storage types; each case twice into dirty outputs, against fp32 arithmetic."""
    _seed(seed)
    ops.set_compute_dtype(dt)
    fails, split_like = [], 0
    ops.set_tuning(L.TUNE_KEYS["SP_CONV_PPW"], ppw)
    try:
        for _ in range(28):
            cout = random.choice([128, 256, 192, 136, 320, 64, 48])
            cin = random.choice([64, 72, 128, 136, 256, 264, 512])
            h, w = 16 * random.randint(1, 4), 32 * random.randint(1, 2)
            rows = 16 if (cout <= 64 or ppw == 2) else 8
            per = (h // rows) * (w // 32) * ((cout + 127) // 128)
            items = random.randint(257, 1200)
            n = max(1, min(48, (items + per - 1) // per))
            if n * h * w * max(cin, cout) > 6e7:
                continue
            split_like += (n * per) % 256 != 0 and n * per > 256
            pool2 = random.choice([0, 0, 0, 1, 2]) if cout % 16 == 0 and cout > 32 else 0
            act = random.choice([0, 1, 2] if pool2 == 0 else [0, 2])
            res = random.choice([0, 0, 1, 2]) if pool2 != 2 else 0
            mask = random.random() < 0.25 and pool2 == 0
            bias = random.random() < 0.8
            up = cout > 32 and pool2 == 0 and random.random() < 0.2
            e = _conv_case(dt, n, cin, cout, 3, h, w, act, res, mask, pool2, bias, up)
            if e > (8e-3 if dt == torch.bfloat16 else 2e-3):
                fails.append((n, cin, cout, h, w, act, res, mask, pool2, bias, up, e))
    finally:
        ops.set_tuning(L.TUNE_KEYS["SP_CONV_PPW"], -1)
    assert not fails, fails
    assert split_like >= 12, split_like
    for t in ops._SPLIT_SYNC.values():
        assert int(t.abs().sum()) == 0           # the counters are clean after every launch


@pytest.mark.parametrize("dt,seed", [(torch.bfloat16, 0), (torch.bfloat16, 5), (torch.float16, 1)])
def test_four_row_wave_kernel_random_shapes(dt, seed):
    """conv_ppw.hip forced onto every launch it covers (SP_TUNE_CONV_PPW = 2): random channel counts (partial K chunks: 72, 136, 264;
    partial channel tiles: 80 ... 320), 16-row multiples, every epilogue operand, pooled epilogues, pooled-gradient inputs, one to
    many items per block - both 16-bit storage types, each case launched twice into dirty outputs, against fp32 arithmetic."""
    _seed(seed)
    ops.set_compute_dtype(dt)
    fails, hits = [], 0
    ops.set_tuning(L.TUNE_KEYS["SP_CONV_PPW"], 2)
    try:
        for _ in range(36):
            cout = random.choice([128, 256, 136, 192, 80, 320, 512])
            cin = random.choice([32, 64, 72, 128, 136, 256, 264])
            n, h, w = random.randint(1, 9), 16 * random.randint(1, 4), 32 * random.randint(1, 3)
            pool2 = random.choice([0, 0, 0, 1, 2]) if cout % 16 == 0 else 0
            act = random.choice([0, 1, 2] if pool2 == 0 else [0, 2])
            res = random.choice([0, 0, 1, 2]) if pool2 != 2 else 0
            mask = random.random() < 0.25 and pool2 == 0
            bias = random.random() < 0.8
            up = pool2 == 0 and random.random() < 0.2
            e = _conv_case(dt, n, cin, cout, 3, h, w, act, res, mask, pool2, bias, up)
            hits += "conv3x3_ppw" in L.lib().sp_last_route().decode()
            if e > (8e-3 if dt == torch.bfloat16 else 2e-3):
                fails.append((n, cin, cout, h, w, act, res, mask, pool2, bias, up, e))
    finally:
        ops.set_tuning(L.TUNE_KEYS["SP_CONV_PPW"], -1)
    assert not fails, fails
    assert hits >= 24, hits                      # (Cout % 16 != 0 cases fall to the other kernels)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-4), (torch.bfloat16, 3e-2)])
def test_batch_norm_and_bilinear_random_shapes(dtype, tol):
    """scratch/stress_norm_resample.py: BatchNorm plain / conditional / fused with the bilinear pass, bilinear x2, vs autograd."""
    _seed(4)
    ops.set_compute_dtype(dtype)
    fails = []
    for _ in range(20):
        c = random.choice([4, 8, 12, 16, 24, 40, 64, 72, 128, 136, 256, 512])
        n = random.randint(1, 6)
        h, w = random.choice([1, 2, 4, 7, 8, 16, 30, 32, 64]), random.choice([1, 2, 4, 8, 9, 16, 32, 48, 64])
        act, cond, fuse_up = random.choice([0, 1]), random.random() < 0.4, random.random() < 0.3
        if n * h * w < 4:                # a channel with 1-3 samples: variance ~ 0, the gradient is 0/0-conditioned in any arithmetic
            continue
        x0 = torch.randn(n, c, h, w, device='cuda').to(dtype).float()
        gy = torch.randn(n, c, h * (2 if fuse_up else 1), w * (2 if fuse_up else 1), device='cuda').to(dtype).float()
        emb = torch.randn(7, 2 * c, device='cuda') if cond else None
        cls = torch.randint(0, 7, (n,), device='cuda') if cond else None
        gamma = None if cond else torch.randn(c, device='cuda')
        beta = None if cond else torch.randn(c, device='cuda')
        xr = x0.clone().requires_grad_(True)
        er = emb.clone().requires_grad_(True) if cond else None
        gr = gamma.clone().requires_grad_(True) if not cond else None
        br = beta.clone().requires_grad_(True) if not cond else None
        mean, var = xr.mean((0, 2, 3), keepdim=True), xr.var((0, 2, 3), unbiased=False, keepdim=True)
        xh = (xr - mean) / torch.sqrt(var + 1e-5)
        sc, bi = (er[cls][:, :c, None, None], er[cls][:, c:, None, None]) if cond else (gr[None, :, None, None], br[None, :, None, None])
        yr = sc * xh + bi
        if act:
            yr = F.leaky_relu(yr, 0.2)
        if fuse_up:
            yr = F.interpolate(yr, scale_factor=2, mode='bilinear', align_corners=True)
        yr.backward(gy)
        for rep in range(2):
            xd = ops.as_nhwc(x0, dtype).requires_grad_(True)
            ed = emb.clone().requires_grad_(True) if cond else None
            gd = gamma.clone().requires_grad_(True) if not cond else None
            bd = beta.clone().requires_grad_(True) if not cond else None
            rm, rv = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
            y = ops.batch_norm(xd, gd, bd, ed, cls, rm, rv, 0.1, 1e-5, True, act, fuse_up)
            y.backward(ops.as_nhwc(gy, dtype))
            errs = [_rel(y.float(), yr), _rel(xd.grad.float(), xr.grad)]
            errs += [_rel(ed.grad, er.grad)] if cond else [_rel(gd.grad, gr.grad), _rel(bd.grad, br.grad)]
            if max(errs) > tol * (4 if n * h * w < 8 else 1):
                fails.append(("bn", rep, n, c, h, w, act, cond, fuse_up, errs))
    for _ in range(10):
        c, n = random.choice([4, 8, 16, 40, 64, 128, 256]), random.randint(1, 5)
        h, w = random.choice([1, 2, 4, 8, 16, 32, 64]), random.choice([1, 2, 4, 8, 16, 32, 64])
        x0 = torch.randn(n, c, h, w, device='cuda').to(dtype).float()
        gy = torch.randn(n, c, 2 * h, 2 * w, device='cuda').to(dtype).float()
        xr = x0.clone().requires_grad_(True)
        yr = F.interpolate(xr, scale_factor=2, mode='bilinear', align_corners=True)
        yr.backward(gy)
        xd = ops.as_nhwc(x0, dtype).requires_grad_(True)
        y = ops.upsample2(xd)
        y.backward(ops.as_nhwc(gy, dtype))
        e = max(_rel(y.float(), yr), _rel(xd.grad.float(), xr.grad))
        if e > tol:
            fails.append(("upsample2", n, c, h, w, e))
    assert not fails, fails


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-4), (torch.bfloat16, 3e-2)])
def test_pooling_and_attention_random_shapes(dtype, tol):
    """scratch/stress_pool_attn.py: avg / max / adaptive / activation + avg pooling, the attention core, vs autograd."""
    _seed(5)
    ops.set_compute_dtype(dtype)
    fails = []
    for _ in range(24):
        c, n = random.choice([4, 8, 16, 40, 64, 72, 128, 256, 512]), random.randint(1, 5)
        h, w = 2 * random.randint(1, 24), 2 * random.randint(1, 24)
        x0 = torch.randn(n, c, h, w, device='cuda').to(dtype).float()
        op = random.choice(["avg", "max", "actavg", "adaptive"])
        xr = x0.clone().requires_grad_(True)
        xd = ops.as_nhwc(x0, dtype).requires_grad_(True)
        if op == "avg":
            yr, y = F.avg_pool2d(xr, 2), ops.avgpool2(xd)
        elif op == "max":
            yr, y = F.max_pool2d(xr, 2), ops.maxpool2(xd)
        elif op == "actavg":
            a_r = F.leaky_relu(xr, 0.2)
            yr = F.avg_pool2d(xr, 2)
            ya, y = ops.act_avgpool2(xd, ops.ACT_LRELU)
            gy2 = torch.randn_like(a_r).to(dtype).float()
        else:
            oh, ow = random.choice([1, 2, 3, 7]), random.choice([1, 2, 3, 7])
            yr, y = F.adaptive_avg_pool2d(xr, (oh, ow)), ops.adaptive_avgpool(xd, oh, ow)
        gy = torch.randn_like(yr).to(dtype).float()
        if op == "actavg":
            ((yr * gy).sum() + (a_r * gy2).sum()).backward()
            ((y.float() * ops.as_nhwc(gy, dtype).float()).sum() + (ya.float() * ops.as_nhwc(gy2, dtype).float()).sum()).backward()
            e = max(_rel(y.float(), F.avg_pool2d(x0, 2)), _rel(ya.float(), F.leaky_relu(x0, 0.2)), _rel(xd.grad.float(), xr.grad))
        else:
            yr.backward(gy)
            y.backward(ops.as_nhwc(gy, dtype))
            e = max(_rel(y.float(), yr), _rel(xd.grad.float(), xr.grad))
        if e > tol:
            fails.append((op, n, c, h, w, e))
    for _ in range(10):
        b, d, dv = random.randint(1, 4), random.choice([32, 64]), random.choice([32, 64, 128, 256])
        hq = random.choice([4, 8, 16, 32])
        hk = hq // 2
        q0 = (torch.randn(b, d, hq, hq, device='cuda') * 0.5).to(dtype).float()
        k0 = (torch.randn(b, d, hk, hk, device='cuda') * 0.5).to(dtype).float()
        v0 = torch.randn(b, dv, hk, hk, device='cuda').to(dtype).float()
        go = torch.randn(b, dv, hq, hq, device='cuda').to(dtype).float()
        qr, kr, vr = (t.clone().requires_grad_(True) for t in (q0, k0, v0))
        att = torch.softmax(torch.bmm(qr.flatten(2).transpose(1, 2), kr.flatten(2)), dim=-1)
        outr = torch.bmm(vr.flatten(2), att.transpose(1, 2)).view(b, dv, hq, hq)
        outr.backward(go)
        for rep in range(2):
            qd, kd, vd = (ops.as_nhwc(t, dtype).requires_grad_(True) for t in (q0, k0, v0))
            o = ops.attention_core(qd, kd, vd)
            o.backward(ops.as_nhwc(go, dtype))
            e = max(_rel(o.float(), outr), _rel(qd.grad.float(), qr.grad), _rel(kd.grad.float(), kr.grad), _rel(vd.grad.float(), vr.grad))
            if e > (tol if dtype == torch.float32 else 4e-2):
                fails.append(("attention", rep, b, d, dv, hq, e))
    assert not fails, fails


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-4), (torch.bfloat16, 3e-2)])
def test_batch_norm_of_the_bilinear_expansion_random_shapes(dtype, tol):
    """The generator's final block (models.py:52-54): UpsamplingBilinear2d -> BatchNorm2d -> LeakyReLU computed from the
    low-resolution tensor (sp_bn_*_up2, the expansion is never written) - against torch autograd, and against the library's own two
    separate passes, which it must reproduce almost exactly (same interpolation, same rounding of the expansion)."""
    _seed(9)
    ops.set_compute_dtype(dtype)
    fails = []
    for it in range(14):
        c = random.choice([4, 8, 16, 24, 64, 72, 128, 256])
        n = random.randint(1, 5)
        h, w = random.choice([1, 2, 3, 4, 8, 16, 31, 32, 64]), random.choice([1, 2, 4, 5, 8, 16, 32, 33, 64])
        if n * h * w < 2:
            continue
        act, cond = random.choice([0, 1]), random.random() < 0.3
        x0 = torch.randn(n, c, h, w, device='cuda').to(dtype).float()
        gy = torch.randn(n, c, 2 * h, 2 * w, device='cuda').to(dtype).float()
        emb = torch.randn(7, 2 * c, device='cuda') if cond else None
        cls = torch.randint(0, 7, (n,), device='cuda') if cond else None
        gamma = None if cond else torch.randn(c, device='cuda')
        beta = None if cond else torch.randn(c, device='cuda')
        xr = x0.clone().requires_grad_(True)
        er = emb.clone().requires_grad_(True) if cond else None
        gr = gamma.clone().requires_grad_(True) if not cond else None
        br = beta.clone().requires_grad_(True) if not cond else None
        u = F.interpolate(xr, scale_factor=2, mode='bilinear', align_corners=True)
        mean, var = u.mean((0, 2, 3), keepdim=True), u.var((0, 2, 3), unbiased=False, keepdim=True)
        uh = (u - mean) / torch.sqrt(var + 1e-5)
        sc, bi = (er[cls][:, :c, None, None], er[cls][:, c:, None, None]) if cond else (gr[None, :, None, None], br[None, :, None, None])
        yr = sc * uh + bi
        if act:
            yr = F.leaky_relu(yr, 0.2)
        yr.backward(gy)
        outs = []
        for mode in ("before", "separate"):
            xd = ops.as_nhwc(x0, dtype).requires_grad_(True)
            ed = emb.clone().requires_grad_(True) if cond else None
            gd = gamma.clone().requires_grad_(True) if not cond else None
            bd = beta.clone().requires_grad_(True) if not cond else None
            rm, rv = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
            if mode == "before":
                y = ops.batch_norm(xd, gd, bd, ed, cls, rm, rv, 0.1, 1e-5, True, act, "before")
            else:
                y = ops.batch_norm(ops.upsample2(xd), gd, bd, ed, cls, rm, rv, 0.1, 1e-5, True, act)
            y.backward(ops.as_nhwc(gy, dtype))
            pg = [ed.grad] if cond else [gd.grad, bd.grad]
            outs.append((y.detach().float(), xd.grad.float(), pg, rm, rv))
        y, dx, pg, rm, rv = outs[0]
        errs = [_rel(y, yr), _rel(dx, xr.grad)] + ([_rel(pg[0], er.grad)] if cond else [_rel(pg[0], gr.grad), _rel(pg[1], br.grad)])
        if max(errs) > tol * (4 if n * h * w < (8 if dtype == torch.float32 else 64) else 1):      # few samples: 16-bit rounding of u weighs in
            fails.append(("vs torch", n, c, h, w, act, cond, errs))
        y2, dx2, pg2, rm2, rv2 = outs[1]
        same = [_rel(y, y2), _rel(dx, dx2), _rel(rm, rm2 + 1e-12), _rel(rv, rv2)] + [_rel(a, b) for a, b in zip(pg, pg2)]
        if max(same) > (1e-5 if dtype == torch.float32 else 2e-2):
            fails.append(("vs separate passes", n, c, h, w, act, cond, same))
    assert not fails, fails


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_maxpool_epilogue_with_positions_random_shapes(dtype):
    """Round 5: conv3x3 -> ReLU -> MaxPool with recorded window positions (sp_conv_params.pool_idx / sp_maxpool2_bwd_idx) on random
    shapes - channel counts that leave partial 64 / 128-channel tiles, Cin with a padded tail, every tile height, ties and not, the
    ping-pong kernels switched off for a third of the cases (the LDS-DMA tall kernels then), fp32 / bf16 / fp16 storage - always BIT-
    IDENTICAL to the unpooled tensor -> sp_maxpool2_fwd -> sp_maxpool2_bwd path (tests/test_gpu_ops.py holds the fixed cases)."""
    import random
    import test_gpu_ops as T
    _seed(11)
    for k in range(18):
        cout = random.choice([48, 64, 80, 128, 144, 192, 256])
        cin = random.choice([16, 24, 64, 72, 128])
        if dtype == torch.float32:
            cin = (cin + 3) // 4 * 4
        n, h, w = random.randint(1, 4), random.choice([8, 16, 24, 32, 48]), random.choice([32, 64])
        T.test_conv_maxpool_epilogue_records_the_window_positions((cin, cout, n, h, w, int(k % 3 == 2)), bool(k % 2), dtype)
