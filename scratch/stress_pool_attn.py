"""Randomised shapes through pooling (avg / max / adaptive / activation + avg), bilinear, scale_add and the attention core against
torch autograd, fp32 and bf16.  Usage: stress_pool_attn.py [seed]"""
import sys, random
sys.path.insert(0, '.')
import torch, torch.nn.functional as F
from semantic_pyramid_for_image_generation_amd import ops
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
random.seed(seed); torch.manual_seed(seed)
worst = {}
def note(kind, err, shape):
    if err > worst.get(kind, (0, None))[0]: worst[kind] = (err, shape)
def rel(a, b): return float((a.detach() - b.detach()).abs().max() / b.detach().abs().max().clamp_min(1e-6))
for dtype, tol in ((torch.float32, 3e-4), (torch.bfloat16, 3e-2)):
    ops.set_compute_dtype(dtype)
    tag = str(dtype)[6:]
    for it in range(30):
        c = random.choice([4, 8, 16, 40, 64, 72, 128, 256, 512]); n = random.randint(1, 5)
        h = 2 * random.randint(1, 24); w = 2 * random.randint(1, 24)
        x0 = torch.randn(n, c, h, w, device='cuda').to(dtype).float()
        op = random.choice(["avg", "max", "actavg", "adaptive"])
        xr = x0.clone().requires_grad_(True); xd = ops.as_nhwc(x0, dtype).requires_grad_(True)
        if op == "avg":
            yr = F.avg_pool2d(xr, 2); y = ops.avgpool2(xd)
        elif op == "max":
            yr = F.max_pool2d(xr, 2); y = ops.maxpool2(xd)
        elif op == "actavg":
            a_r = F.leaky_relu(xr, 0.2); yr = a_r.sum() * 0 + F.avg_pool2d(xr, 2)
            ya, y = ops.act_avgpool2(xd, ops.ACT_LRELU)
            gy2 = torch.randn_like(a_r).to(dtype).float()
        else:
            oh, ow = random.choice([1, 2, 3, 7]), random.choice([1, 2, 3, 7])
            yr = F.adaptive_avg_pool2d(xr, (oh, ow)); y = ops.adaptive_avgpool(xd, oh, ow)
        gy = torch.randn_like(yr).to(dtype).float()
        if op == "actavg":
            (yr * gy).sum().backward(retain_graph=True); (a_r * gy2).sum().backward()
            (y.float() * ops.as_nhwc(gy, dtype).float()).sum().backward(retain_graph=True); (ya.float() * ops.as_nhwc(gy2, dtype).float()).sum().backward()
            e = max(rel(y.float(), F.avg_pool2d(x0, 2)), rel(ya.float(), F.leaky_relu(x0, 0.2)), rel(xd.grad.float(), xr.grad))
        else:
            yr.backward(gy); y.backward(ops.as_nhwc(gy, dtype))
            e = max(rel(y.float(), yr), rel(xd.grad.float(), xr.grad))
        note("%s %s" % (op, tag), e, (n, c, h, w))
        if e > tol: print("FAIL", op, tag, (n, c, h, w), e, flush=True)
    for it in range(12):
        b = random.randint(1, 4); d = random.choice([32, 64]); dv = random.choice([32, 64, 128, 256])
        hq = random.choice([4, 8, 16, 32]); wq = hq; hk = hq // 2; wk = wq // 2
        q0 = (torch.randn(b, d, hq, wq, device='cuda') * 0.5).to(dtype).float(); k0 = (torch.randn(b, d, hk, wk, device='cuda') * 0.5).to(dtype).float()
        v0 = torch.randn(b, dv, hk, wk, device='cuda').to(dtype).float(); go = torch.randn(b, dv, hq, wq, device='cuda').to(dtype).float()
        qr, kr, vr = (t.clone().requires_grad_(True) for t in (q0, k0, v0))
        att = torch.softmax(torch.bmm(qr.flatten(2).transpose(1, 2), kr.flatten(2)), dim=-1)          # [b, n, nk]
        outr = torch.bmm(vr.flatten(2), att.transpose(1, 2)).view(b, dv, hq, wq)
        outr.backward(go)
        qd, kd, vd = (ops.as_nhwc(t, dtype).requires_grad_(True) for t in (q0, k0, v0))
        o = ops.attention_core(qd, kd, vd); o.backward(ops.as_nhwc(go, dtype))
        e = max(rel(o.float(), outr), rel(qd.grad.float(), qr.grad), rel(kd.grad.float(), kr.grad), rel(vd.grad.float(), vr.grad))
        note("attention " + tag, e, (b, d, dv, hq))
        if e > (tol if dtype == torch.float32 else 4e-2): print("FAIL attention", tag, (b, d, dv, hq), e, flush=True)
ops.set_compute_dtype(torch.float32)
for k, (e, sh) in sorted(worst.items()): print("%-20s worst rel err %.2e at %s" % (k, e, sh))
print("seed %d done" % seed)
