"""Randomised shapes through the BatchNorm / bilinear passes rewritten in round 3 (16-byte parameter loads, row-per-block
resampling), fp32 storage (the parity mode's tolerances) against torch autograd.  Usage: stress_norm_resample.py [seed]"""
import sys, random
sys.path.insert(0, '.')
import torch, torch.nn.functional as F
from semantic_pyramid_for_image_generation_amd import ops
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
random.seed(seed); torch.manual_seed(seed)
worst = {}
def note(kind, err, shape):
    if err > worst.get(kind, (0, None))[0]: worst[kind] = (err, shape)
def rel(a, b): return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))
for dtype, tol in ((torch.float32, 3e-4), (torch.bfloat16, 3e-2)):
    ops.set_compute_dtype(dtype)
    for it in range(25):
        c = random.choice([4, 8, 12, 16, 24, 40, 64, 72, 128, 136, 256, 512]); n = random.randint(1, 6)
        h = random.choice([1, 2, 4, 7, 8, 16, 30, 32, 64]); w = random.choice([1, 2, 4, 8, 9, 16, 32, 48, 64])
        act = random.choice([0, 1]); cond = random.random() < 0.4; fuse_up = random.random() < 0.3
        x0 = torch.randn(n, c, h, w, device='cuda').to(dtype).float()
        gy = torch.randn(n, c, h * (2 if fuse_up else 1), w * (2 if fuse_up else 1), device='cuda').to(dtype).float()
        classes = 7
        emb = torch.randn(classes, 2 * c, device='cuda') if cond else None
        cls = torch.randint(0, classes, (n,), device='cuda') if cond else None
        gamma = None if cond else torch.randn(c, device='cuda'); beta = None if cond else torch.randn(c, device='cuda')
        # reference
        xr = x0.clone().requires_grad_(True)
        er = emb.clone().requires_grad_(True) if cond else None
        gr = gamma.clone().requires_grad_(True) if not cond else None; br = beta.clone().requires_grad_(True) if not cond else None
        mean = xr.mean((0, 2, 3), keepdim=True); var = xr.var((0, 2, 3), unbiased=False, keepdim=True)
        xh = (xr - mean) / torch.sqrt(var + 1e-5)
        if cond: sc, bi = er[cls][:, :c, None, None], er[cls][:, c:, None, None]
        else: sc, bi = gr[None, :, None, None], br[None, :, None, None]
        yr = sc * xh + bi
        if act: yr = F.leaky_relu(yr, 0.2)
        if fuse_up: yr = F.interpolate(yr, scale_factor=2, mode='bilinear', align_corners=True)
        yr.backward(gy)
        # kernels
        xd = ops.as_nhwc(x0, dtype).requires_grad_(True)
        ed = emb.clone().requires_grad_(True) if cond else None
        gd = gamma.clone().requires_grad_(True) if not cond else None; bd = beta.clone().requires_grad_(True) if not cond else None
        rm, rv = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
        y = ops.batch_norm(xd, gd, bd, ed, cls, rm, rv, 0.1, 1e-5, True, act, fuse_up)
        y.backward(ops.as_nhwc(gy, dtype))
        shape = (str(dtype)[6:], n, c, h, w, act, cond, fuse_up)
        errs = [rel(y.float(), yr), rel(xd.grad.float(), xr.grad)]
        errs += [rel(ed.grad, er.grad)] if cond else [rel(gd.grad, gr.grad), rel(bd.grad, br.grad)]
        e = max(errs)
        note("batch_norm " + str(dtype)[6:], e, shape)
        if e > tol * (4 if n * h * w < 8 else 1): print("FAIL bn", shape, errs, flush=True)
    for it in range(15):
        c = random.choice([4, 8, 16, 40, 64, 128, 256]); n = random.randint(1, 5); h = random.choice([1, 2, 4, 8, 16, 32, 64]); w = random.choice([1, 2, 4, 8, 16, 32, 64])
        x0 = torch.randn(n, c, h, w, device='cuda').to(dtype).float(); gy = torch.randn(n, c, 2 * h, 2 * w, device='cuda').to(dtype).float()
        xr = x0.clone().requires_grad_(True)
        yr = F.interpolate(xr, scale_factor=2, mode='bilinear', align_corners=True); yr.backward(gy)
        xd = ops.as_nhwc(x0, dtype).requires_grad_(True)
        y = ops.upsample2(xd); y.backward(ops.as_nhwc(gy, dtype))
        e = max(rel(y.float(), yr), rel(xd.grad.float(), xr.grad))
        note("upsample2 " + str(dtype)[6:], e, (n, c, h, w))
        if e > tol: print("FAIL upsample", (str(dtype)[6:], n, c, h, w), e, flush=True)
ops.set_compute_dtype(torch.float32)
for k, (e, sh) in sorted(worst.items()): print("%-22s worst rel err %.2e at %s" % (k, e, sh))
print("seed %d done" % seed)
