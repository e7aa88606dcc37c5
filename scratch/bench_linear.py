"""Split-K MFMA linear kernel at the step's big shapes (VGG-16 classifier forward over 40 rows, its input gradients over 20, the
generator's 4096-wide mapping): microseconds per call (kernel + finalize) and the rate the packed weights are streamed at."""
import sys
sys.path.insert(0, '/root/repo')
import torch
from semantic_pyramid_for_image_generation_amd import ops
dt = torch.bfloat16
SHAPES = [(40, 25088, 4096), (40, 4096, 4096), (40, 4096, 1000), (20, 4096, 25088), (20, 4096, 4096), (20, 1000, 4096), (20, 4096, 2048), (20, 365, 2048), (20, 2048, 4096)]
def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
big = torch.empty(256 << 20, dtype=torch.uint8, device='cuda')          # flushed between calls: the weights come from HBM as in the step
for b, k, n in SHAPES:
    kp = (k + 7) // 8 * 8
    x = (torch.randn(b, k, device='cuda')).to(dt)
    w = (torch.randn(n, kp, device='cuda') * 0.02).to(dt)
    if kp != k: w[:, k:] = 0
    bias = torch.randn(n, device='cuda')
    y = torch.empty(b, n, dtype=dt, device='cuda')
    f = lambda: ops.linear_launch(x, w.data_ptr(), kp, bias, None, y, b, k, n, 0)
    ref = x.float() @ w[:, :k].float().t() + bias
    row = []
    for ks in (0, 1024, 512, 256, 128):
        ops.set_tuning(27, ks)
        y.zero_(); f(); torch.cuda.synchronize()
        err = float((y.float() - ref).abs().max() / ref.abs().max())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(10):
            big.zero_()
            e0.record(); f(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort(); cold = ts[len(ts) // 2]
        row.append("%s %5.1f us%s" % ("auto" if ks == 0 else "ks%d" % ks, cold * 1e3, "" if err < 8e-3 else " ERR %.1e" % err))
        if ks == 0: auto = cold
    ops.set_tuning(27, -1)
    print("b=%2d k=%5d n=%5d  weights %6.1f MB  cold: %s   (auto %4.2f TB/s)" % (b, k, n, n * kp * 2 / 1e6, "  ".join(row), n * kp * 2 / auto / 1e9), flush=True)
