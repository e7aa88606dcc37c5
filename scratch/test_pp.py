"""Ping-pong 3x3 kernel (conv_pp.hip) vs the round-2 kernels: bit-exact comparison over shapes / epilogue variants (several
repeats: race screen), then an interleaved A/B timing on the step's fat shapes (bf16, B = 20)."""
import sys, os, itertools
sys.path.insert(0, '.')
import torch
from semantic_pyramid_for_image_generation_amd import ops, _lib as L
dt = torch.bfloat16
lib = L.lib()
PP, PRIO = 21, 22
def setpp(v, prio=-1):
    lib.sp_set_tuning(PP, v); lib.sp_set_tuning(PRIO, prio)

def run(x, w, bias, y, r1, r2, ms, n, hw, cin, cout, ldy, act, pool2=0, up=False):
    ops._conv_launch(x, w.data_ptr(), bias, y, r1, r2, ms, 0.2, n, hw, hw, cin, cout, ldy, 3, act, dt, pool2, up)

def check(n, cin, cout, hw, act=1, res=0, mask=False, pool2=0, up=False, bias=True, modes=(8,), reps=3, seed=0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    hin = hw // 2 if up else hw
    x = ops.nhwc_empty(n, cin, hin, hin, dt, 'cuda'); x.normal_(generator=g)
    w = (torch.randn(cout * 9 * cin, device='cuda', generator=g) * 0.05).to(dt)
    b = torch.randn(cout, device='cuda', generator=g) if bias else None
    ho = hw // 2 if pool2 else hw
    def mk():
        t = ops.nhwc_empty(n, cout, ho, ho, dt, 'cuda'); t.normal_(generator=g); return t
    r1 = mk() if res >= 1 else None
    r2 = mk() if res >= 2 else None
    ms = mk() if mask else None
    y0 = ops.nhwc_zeros(n, cout, ho, ho, dt, 'cuda')
    setpp(0)
    run(x, w, b, y0, r1, r2, ms, n, hw, cin, cout, cout, act, pool2, up)
    torch.cuda.synchronize()
    ok = True
    for m in modes:
        for prio in tuple(int(a) for a in os.environ.get("PP_PRIOS", "1,17").split(",")):
            for rep in range(reps):
                y1 = ops.nhwc_zeros(n, cout, ho, ho, dt, 'cuda')
                setpp(m, prio)
                run(x, w, b, y1, r1, r2, ms, n, hw, cin, cout, cout, act, pool2, up)
                torch.cuda.synchronize()
                if not torch.equal(y0, y1):
                    d = (y0.float() - y1.float()).abs()
                    bad = (d > 0).sum().item()
                    print("MISMATCH mode %d prio %d rep %d: n=%d %d->%d @%d act=%d res=%d mask=%d pool2=%d up=%d bias=%d: %d elements differ, max %.4g, nan %d"
                          % (m, prio, rep, n, cin, cout, hw, act, res, mask, pool2, up, bias, bad, d.max().item(), torch.isnan(y1.float()).sum().item()))
                    ok = False
                    break
    setpp(-1)
    return ok

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

if __name__ == "__main__":
    modes = tuple(int(a) for a in os.environ.get("PP_MODES", "8").split(","))
    allok = True
    cases = [
        dict(n=2, cin=128, cout=128, hw=64), dict(n=20, cin=128, cout=128, hw=128), dict(n=20, cin=256, cout=256, hw=64),
        dict(n=20, cin=512, cout=512, hw=32), dict(n=20, cin=64, cout=128, hw=128, res=2, act=0), dict(n=3, cin=136, cout=128, hw=64),
        dict(n=3, cin=264, cout=256, hw=32, act=2), dict(n=5, cin=32, cout=192, hw=32, act=3), dict(n=2, cin=64, cout=72, hw=32),
        dict(n=1, cin=8, cout=96, hw=32, bias=False), dict(n=7, cin=128, cout=128, hw=64, mask=True, act=0, bias=False),
        dict(n=4, cin=128, cout=256, hw=64, pool2=1, res=2, act=0), dict(n=4, cin=64, cout=128, hw=64, pool2=2, act=2),
        dict(n=4, cin=256, cout=128, hw=64, up=True, act=0, bias=False, mask=True), dict(n=20, cin=256, cout=128, hw=128, up=True, act=0, bias=False),
        dict(n=20, cin=256, cout=512, hw=32, res=1, act=1), dict(n=20, cin=512, cout=256, hw=32), dict(n=1, cin=32, cout=128, hw=32),
        dict(n=13, cin=96, cout=130, hw=32),
        # Cout <= 64: the 8-row-pair-wave form (WCO = 1, 16 x 32 patches)
        dict(n=2, cin=64, cout=64, hw=64), dict(n=20, cin=64, cout=64, hw=256), dict(n=3, cin=128, cout=64, hw=128, res=2, act=0),
        dict(n=2, cin=72, cout=64, hw=32, mask=True, act=0, bias=False), dict(n=2, cin=64, cout=64, hw=64, pool2=2, act=2),
        dict(n=2, cin=64, cout=64, hw=64, pool2=1, res=1, act=0), dict(n=2, cin=128, cout=64, hw=64, up=True, act=0, bias=False),
        dict(n=3, cin=32, cout=40, hw=32), dict(n=1, cin=64, cout=24, hw=32, act=3),
    ]
    for c in cases:
        ok = check(modes=modes, **c)
        print("case %s: %s" % (c, "ok" if ok else "FAIL"), flush=True)
        allok &= ok
    print("ALL OK" if allok else "SOME FAILED", flush=True)
    # ---- timing: interleaved rounds, old vs pp modes
    B = 20
    SHAPES = [(64, 64, 256), (128, 64, 128), (72, 64, 128), (128, 128, 128), (256, 256, 64), (512, 512, 32), (64, 128, 128), (128, 256, 64), (256, 512, 32), (256, 256, 32), (264, 256, 32), (136, 128, 64)]
    variants = [("old", 0, -1)] + [("pp%d/p%d" % (m, pr), m, pr) for m in modes for pr in tuple(int(a) for a in os.environ.get("PP_TPRIOS", "1,17").split(","))]
    tot = {v[0]: 0.0 for v in variants}
    for cin, cout, hw in SHAPES:
        x = ops.nhwc_empty(B, cin, hw, hw, dt, 'cuda'); x.normal_()
        w = (torch.randn(cout * 9 * cin, device='cuda') * 0.05).to(dt)
        bias = torch.randn(cout, device='cuda')
        y = ops.nhwc_empty(B, cout, hw, hw, dt, 'cuda')
        flops = 2.0 * B * hw * hw * cin * cout * 9
        res = {v[0]: [] for v in variants}
        for rnd in range(3):
            for name, m, pr in variants:
                setpp(m, pr)
                res[name].append(timeit(lambda: run(x, w, bias, y, None, None, None, B, hw, cin, cout, cout, 1)))
        line = "%4d->%4d @%3d " % (cin, cout, hw)
        for name, _, _ in variants:
            t = min(res[name]); tot[name] += t
            line += "| %s %6.1f us %6.0f TF " % (name, t * 1e3, flops / t / 1e9)
        print(line, flush=True)
    print("sum: " + " | ".join("%s %.3f ms" % (k, v) for k, v in tot.items()))
    setpp(-1)
