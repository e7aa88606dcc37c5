"""A/B micro-benchmark of the step's 1x1 convolutions (bf16, B=20): run once per build (SEMPYR_LIB selects the library).
Also checks the result against torch (fp32 matmul of the bf16 operands)."""
import sys, os
sys.path.insert(0, '.')
import torch
from semantic_pyramid_for_image_generation_amd import ops, _lib as L
B = 20
dt = torch.bfloat16
# (cin, cout, hw, count per step)
SHAPES = [(256, 128, 32, 9), (64, 3, 256, 2), (128, 256, 32, 9), (256, 256, 16, 6), (768, 512, 2, 3), (256, 256, 8, 6), (8, 64, 128, 3),
          (128, 64, 64, 5), (64, 128, 64, 4), (256, 32, 32, 5), (512, 768, 2, 3), (256, 128, 16, 5), (512, 256, 4, 3), (512, 512, 4, 3),
          (512, 512, 8, 3), (256, 32, 16, 5), (256, 512, 4, 3), (512, 256, 16, 2), (8, 64, 256, 1), (128, 256, 16, 4), (32, 256, 32, 4),
          (32, 256, 16, 4), (512, 128, 4, 1), (128, 512, 4, 2), (64, 3, 128, 1), (256, 512, 16, 1)]
def timeit(fn, iters=40):
    """Device time per launch: the launches are replayed from a captured graph (the eager call costs ~8 us of host time)."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(iters): fn()
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(3): gr.replay()
        e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters / 3
tag = os.path.basename(os.environ.get("SEMPYR_LIB", "current"))
tot = 0.0
for cin, cout, hw, cnt in SHAPES:
    torch.manual_seed(cin * 7 + cout + hw)
    x = ops.nhwc_empty(B, cin, hw, hw, dt, 'cuda'); x.normal_()
    w = (torch.randn(cout, cin, device='cuda') * 0.05).to(dt)
    bias = torch.randn(cout, device='cuda')
    ldy = (cout + 7) // 8 * 8
    y = ops.nhwc_empty(B, ldy, hw, hw, dt, 'cuda'); y.zero_()
    fn = lambda: ops.conv_launch(x, w.data_ptr(), bias, y, None, None, None, 0.0, B, hw, hw, cin, cout, ldy, 1, 0, dt)
    t = timeit(fn)
    xr = x.permute(0, 2, 3, 1).reshape(-1, cin).float()
    ref = xr @ w.float().t() + bias
    got = y.permute(0, 2, 3, 1).reshape(-1, ldy)[:, :cout].float()
    err = float((got - ref).abs().max() / ref.abs().max())
    gb = (B * hw * hw * (cin + cout) * 2 + cin * cout * 2) / 1e9
    tot += t * cnt
    print("%-22s %4d->%4d @%3d x%d  %7.1f us  %7.1f GB/s  err %.1e" % (tag, cin, cout, hw, cnt, t * 1e3, gb / t * 1e3, err))
print("%-22s weighted sum %.3f ms/step" % (tag, tot))
