"""Bandwidth of the memory-bound kernels (BatchNorm family, bilinear resampling) at the step's shapes vs a plain device copy of
the same bytes.  Replayed HIP graphs of 20 back-to-back launches on rotating buffers (no L2 / MALL reuse between launches)."""
import sys; sys.path.insert(0, '.')
import torch
from semantic_pyramid_for_image_generation_amd import ops, _lib as L
lib = L.lib(); dt = torch.bfloat16; B = 20
NBUF = 6

def timed(fn_of_i, iters=18):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(3): fn_of_i(i)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for i in range(iters): fn_of_i(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best * 1e3   # us

import os
for c, hw in [(64, 256), (128, 128), (256, 64), (512, 32), (64, 128), (128, 64), (512, 16)]:
    xs = [ops.nhwc_empty(B, c, hw, hw, dt, 'cuda').normal_() for _ in range(NBUF)]
    ys = [ops.nhwc_empty(B, c, hw, hw, dt, 'cuda') for _ in range(NBUF)]
    dys = [ops.nhwc_empty(B, c, hw, hw, dt, 'cuda').normal_() for _ in range(NBUF)]
    y2 = [ops.nhwc_empty(B, c, 2 * hw, 2 * hw, dt, 'cuda') for _ in range(2)] if hw <= 128 else None
    mean = torch.zeros(c, device='cuda'); invstd = torch.ones(c, device='cuda'); gamma = torch.ones(c, device='cuda'); beta = torch.zeros(c, device='cuda')
    sums = torch.empty(1024 * 2 * c, device='cuda'); rm = torch.zeros(c, device='cuda'); rv = torch.ones(c, device='cuda')
    red = torch.empty(1024 * 2 * c, device='cuda'); ctmp = torch.empty(2 * c, device='cuda'); dg = torch.empty(c, device='cuda'); db = torch.empty(c, device='cuda')
    P = ops.ptr; st = ops.stream; sd = ops.sp_dtype(dt)
    nbytes = B * c * hw * hw * 2
    res = {}
    res["copy (r+w)"] = (timed(lambda i: ys[i % NBUF].copy_(xs[i % NBUF])), 2 * nbytes)
    res["bn_stats (r)"] = (timed(lambda i: L.call("sp_bn_stats", P(xs[i % NBUF]), B, hw * hw, c, P(sums), 1e-5, 0.1, P(rm), P(rv), 1, P(mean), P(invstd), sd, st())), nbytes)
    res["bn_apply (r+w)"] = (timed(lambda i: L.call("sp_bn_apply", P(xs[i % NBUF]), P(ys[i % NBUF]), B, hw * hw, c, P(mean), P(invstd), P(gamma), P(beta), None, None, 1, sd, st())), 2 * nbytes)
    res["bn_backward (2r + 2r+w)"] = (timed(lambda i: L.call("sp_bn_backward", P(dys[i % NBUF]), P(xs[i % NBUF]), P(ys[i % NBUF]), B, hw * hw, c, P(mean), P(invstd), P(gamma), P(beta), None, None, 1, P(red), P(ctmp), P(dg), P(db), None, 0, sd, st())), 5 * nbytes)
    if y2 is not None:
        res["upsample2_fwd (r+4w)"] = (timed(lambda i: L.call("sp_upsample2_fwd", P(xs[i % NBUF]), P(y2[i % 2]), B, hw, hw, c, sd, st())), 5 * nbytes)
    print("C=%d %dx%d (%.0f MB per tensor): " % (c, hw, hw, nbytes / 1e6) + " | ".join("%s %.1f us %.2f TB/s" % (k, v[0], v[1] / v[0] / 1e6) for k, v in res.items()), flush=True)
