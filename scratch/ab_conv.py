"""A/B micro-benchmark of sp_conv2d_igemm on the step's fat shapes (bf16, B=20) WITH the bias (and, second column, both residuals):
run once per build (SEMPYR_LIB selects the library)."""
import sys, os
sys.path.insert(0, '.')
import torch
from semantic_pyramid_for_image_generation_amd import ops, _lib as L
B = 20
dt = torch.bfloat16
SHAPES = [(64, 64, 256), (128, 128, 128), (256, 256, 64), (512, 512, 32), (64, 128, 128), (128, 256, 64), (256, 512, 32), (256, 256, 32),
          (512, 512, 16), (8, 64, 256), (64, 3, 256)]
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
tag = os.path.basename(os.environ.get("SEMPYR_LIB", "current"))
tot = [0.0, 0.0]
for cin, cout, hw in SHAPES:
    x = ops.nhwc_empty(B, cin, hw, hw, dt, 'cuda'); x.normal_()
    w = (torch.randn(cout * 9 * cin, device='cuda') * 0.05).to(dt)
    bias = torch.randn(cout, device='cuda')
    r1 = ops.nhwc_empty(B, cout, hw, hw, dt, 'cuda'); r1.normal_()
    r2 = ops.nhwc_empty(B, cout, hw, hw, dt, 'cuda'); r2.normal_()
    y = ops.nhwc_empty(B, cout, hw, hw, dt, 'cuda')
    flops = 2.0 * B * hw * hw * cin * cout * 9
    t0 = timeit(lambda: ops.conv_launch(x, w.data_ptr(), bias, y, None, None, None, 0.0, B, hw, hw, cin, cout, cout, 3, 1, dt))
    t1 = timeit(lambda: ops.conv_launch(x, w.data_ptr(), bias, y, r1, r2, None, 0.0, B, hw, hw, cin, cout, cout, 3, 0, dt))
    tot[0] += t0; tot[1] += t1
    print("%-22s %4d->%4d @%3d  bias+lrelu %7.1f us %7.1f TF | bias+2res %7.1f us %7.1f TF" % (tag, cin, cout, hw, t0 * 1e3, flops / t0 / 1e9, t1 * 1e3, flops / t1 / 1e9))
print("%-22s sum %.3f ms | %.3f ms" % (tag, tot[0], tot[1]))
