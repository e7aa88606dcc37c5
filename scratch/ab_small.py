"""Small-spatial 3x3 layers (4x4 .. 16x16, bf16, B=20) through sp_conv2d_igemm, device time from a replayed graph; checks the
result against a plain fp32 convolution of the same bf16 operands.  SP_DMA_TILE / SP_SPLITK_TARGET select variants."""
import sys, os
sys.path.insert(0, '.')
import torch
import torch.nn.functional as F
from semantic_pyramid_for_image_generation_amd import ops
B = 20
dt = torch.bfloat16
SHAPES = [(512, 512, 16, 15), (256, 256, 16, 12), (512, 512, 8, 12), (768, 768, 4, 6), (520, 512, 16, 2), (512, 768, 4, 2), (256, 512, 16, 2)]
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(iters): fn()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(3): gr.replay()
        e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters / 3
tag = "tile=%s target=%s" % (os.environ.get("SP_DMA_TILE", "-"), os.environ.get("SP_SPLITK_TARGET", "-"))
tot = 0.0
for cin, cout, hw, cnt in SHAPES:
    torch.manual_seed(1)
    x = ops.nhwc_empty(B, cin, hw, hw, dt, 'cuda'); x.normal_()
    w4 = (torch.randn(cout, 3, 3, cin, device='cuda') * 0.02).to(dt)       # packed layout [co][tap][ci]
    bias = torch.randn(cout, device='cuda')
    y = ops.nhwc_empty(B, cout, hw, hw, dt, 'cuda')
    flops = 2.0 * B * hw * hw * cin * cout * 9
    fn = lambda: ops.conv_launch(x, w4.data_ptr(), bias, y, None, None, None, 0.0, B, hw, hw, cin, cout, cout, 3, 0, dt)
    t = timeit(fn)
    ref = F.conv2d(x.float().contiguous(), w4.float().permute(0, 3, 1, 2).contiguous(), bias, padding=1)
    err = float((y.float() - ref).abs().max() / ref.abs().max())
    tot += t * cnt
    print("%-22s %4d->%4d @%3d x%-2d %7.1f us %6.1f TF  err %.1e" % (tag, cin, cout, hw, cnt, t * 1e3, flops / t / 1e9, err))
print("%-22s weighted sum %.3f ms/step" % (tag, tot))
