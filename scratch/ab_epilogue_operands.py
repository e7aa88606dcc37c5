"""What the epilogue operands of the big 3x3 launches cost (round 5): the same launch with no extra operand, with the
activation-derivative mask (mask_src), with a residual (res1) and with both - the input-gradient launches of the residual blocks
carry the mask (and the shortcut's gradient) and run ~25 % below the forward launches of the same shape."""
import sys
sys.path.insert(0, '/root/repo')
import torch
from semantic_pyramid_for_image_generation_amd import ops
dt = torch.bfloat16
SHAPES = [(128, 128, 128, 40), (64, 64, 256, 40), (64, 64, 256, 20), (256, 256, 64, 40), (128, 128, 128, 20), (512, 512, 32, 40), (128, 64, 256, 20)]
def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for cin, cout, hw, B in SHAPES:
    x = ops.nhwc_empty(B, cin, hw, hw, dt, 'cuda'); x.normal_()
    w = (torch.randn(cout * 9 * cin, device='cuda') * 0.02).to(dt)
    m = ops.nhwc_empty(B, cout, hw, hw, dt, 'cuda'); m.normal_()
    r = ops.nhwc_empty(B, cout, hw, hw, dt, 'cuda'); r.normal_()
    y = ops.nhwc_empty(B, cout, hw, hw, dt, 'cuda')
    flops = 2.0 * B * hw * hw * cin * cout * 9
    row = []
    for tag, res, mask in (("plain", None, None), ("mask", None, m), ("res", r, None), ("mask+res", r, m)):
        f = lambda: ops.conv_launch(x, w.data_ptr(), None, y, res, None, mask, 0.2, B, hw, hw, cin, cout, cout, 3, 0, dt)
        t = timeit(f)
        row.append("%s %6.1f us %6.0f TF" % (tag, t * 1e3, flops / t / 1e9))
    print("%4d->%4d @%3d N=%2d: %s" % (cin, cout, hw, B, "   ".join(row)), flush=True)
