"""conv1x1_direct at the step's large-map shapes: microseconds per launch and the rate of its compulsory traffic (x + y)."""
import sys
sys.path.insert(0, '/root/repo')
import torch
from semantic_pyramid_for_image_generation_amd import ops
dt = torch.bfloat16
SHAPES = [(8, 64, 256, 20), (8, 64, 128, 40), (64, 128, 64, 40), (128, 256, 32, 40), (64, 128, 64, 20), (128, 64, 64, 20), (64, 8, 256, 20), (256, 128, 32, 20), (256, 512, 16, 40), (128, 64, 128, 20)]
def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for cin, cout, hw, B in SHAPES:
    x = ops.nhwc_empty(B, cin, hw, hw, dt, 'cuda'); x.normal_()
    w = (torch.randn(cout * cin, device='cuda') * 0.05).to(dt)
    bias = torch.randn(cout, device='cuda')
    cp = max(cout, 8)
    y = ops.nhwc_empty(B, cp, hw, hw, dt, 'cuda')
    f = lambda: ops.conv_launch(x, w.data_ptr(), bias, y, None, None, None, 0.0, B, hw, hw, cin, cout, cp, 1, 0, dt)
    t = timeit(f)
    ref = torch.nn.functional.conv2d(x.float(), w.float().view(cout, cin, 1, 1), bias)
    err = float((y.float()[:, :cout] - ref).abs().max() / ref.abs().max())
    mb = B * hw * hw * (cin + cp) * 2 / 1e6
    print("%4d->%4d @%3d N=%2d: %6.1f us  %6.1f MB  %5.2f TB/s  err %.1e" % (cin, cout, hw, B, t * 1e3, mb, mb / t / 1e3, err), flush=True)
