"""us per launch of the 1x1 convolution shapes of the step (bf16) under the library SEMPYR_LIB points at; the buffers are rotated so
that operands come from HBM like in the step (8 copies)."""
import sys
sys.path.insert(0, '.')
import torch
from semantic_pyramid_for_image_generation_amd import ops
dt = torch.bfloat16
# (cin_p, cout, h, w, n, act)
SHAPES = [(64, 3, 256, 256, 20, ops.ACT_TANH), (8, 64, 256, 256, 20, 0), (8, 64, 128, 128, 40, 0), (64, 128, 64, 64, 40, 0), (128, 256, 32, 32, 40, 0),
          (128, 64, 256, 256, 20, 0), (256, 128, 128, 128, 20, 0), (64, 128, 128, 128, 20, 0), (256, 32, 64, 64, 20, 0), (256, 256, 32, 32, 20, 0), (512, 256, 64, 64, 20, 0)]
def timeit(fns, rounds=5):
    for f in fns: f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for f in fns: f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / len(fns) * 1e3)
    return best
tot = 0.0
for cin, cout, h, w, n, act in SHAPES:
    wt = (torch.randn(cout, 1, 1, cin, device='cuda') * 0.05).to(dt)
    b = torch.randn(cout, device='cuda')
    cp = ops.pad_channels(cout, dt) if cout > 3 else cout
    fns = []
    for _ in range(8):
        x = ops.nhwc_empty(n, cin, h, w, dt, 'cuda').normal_()
        y = ops.nhwc_empty(n, cp, h, w, dt, 'cuda')
        fns.append(lambda x=x, y=y: ops._conv_launch(x, wt.data_ptr(), b, y, None, None, None, 0.0, n, h, w, cin, cout, cp, 1, act, dt))
    t = timeit(fns)
    mb = n * h * w * (cin + cp) * 2 / 1e6
    print("%4d->%4d @%3d^2 n=%2d: %6.1f us  %5.2f TB/s" % (cin, cout, h, n, t, mb / t / 1e6 * 1e6 / 1e6), flush=True)
    tot += t
print("sum %.1f us" % tot)
