"""A/B (round 5): output tile and split-K target of the LDS-DMA igemm on the small-spatial 3x3 layers (SP_TUNE_IGEMM_TILE / SP_TUNE_SPLITK_TARGET)."""
import sys
sys.path.insert(0, '/root/repo')
import torch
from semantic_pyramid_for_image_generation_amd import ops, _lib as L
dt = torch.bfloat16
SHAPES = [(512, 512, 8, 20), (512, 512, 8, 40), (520, 512, 8, 20), (256, 512, 8, 40), (768, 768, 4, 20), (768, 768, 4, 40), (512, 768, 4, 40), (520, 512, 4, 20)]
def timeit(fn, iters=30):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for cin, cout, hw, B in SHAPES:
    x = ops.nhwc_empty(B, cin, hw, hw, dt, 'cuda'); x.normal_()
    w = (torch.randn(cout * 9 * cin, device='cuda') * 0.02).to(dt)
    bias = torch.randn(cout, device='cuda')
    flops = 2.0 * B * hw * hw * cin * cout * 9
    ref = None
    row = []
    for tile in (0, 1, 2, 3):
        for target in (640, 384, 256):
            ops.set_tuning(L.TUNE_KEYS["SP_IGEMM_TILE"], tile)
            ops.set_tuning(L.TUNE_KEYS["SP_SPLITK_TARGET"], target)
            y = ops.nhwc_empty(B, cout, hw, hw, dt, 'cuda')
            f = lambda: ops.conv_launch(x, w.data_ptr(), bias, y, None, None, None, 0.0, B, hw, hw, cin, cout, cout, 3, 1, dt)
            t = timeit(f)
            if ref is None:
                ref = y.float().clone()
            err = float((y.float() - ref).abs().max() / ref.abs().max())
            row.append("t%d/%d %5.1fus%s" % (tile, target, t * 1e3, "" if err < 2e-2 else " ERR%.1e" % err))
    ops.set_tuning(L.TUNE_KEYS["SP_IGEMM_TILE"], -1); ops.set_tuning(L.TUNE_KEYS["SP_SPLITK_TARGET"], -1)
    print("%4d->%4d @%2d N=%2d (%5.1f GF): %s" % (cin, cout, hw, B, flops / 1e9, "  ".join(row)), flush=True)
