"""pp8 (8-row ping-pong tiles) vs tall16 (16-row lockstep tiles) per layer shape and batch: calibrates conv_igemm.hip's round rule."""
import sys
sys.path.insert(0, '.')
import torch
from semantic_pyramid_for_image_generation_amd import ops
dt = torch.bfloat16
SHAPES = [(64, 128, 128), (128, 128, 128), (128, 256, 64), (256, 256, 64), (256, 128, 64), (256, 256, 32), (512, 512, 32), (256, 512, 32), (512, 256, 32), (128, 128, 64)]
def timeit(fn, reps=20, rounds=4):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
for n in (20, 40):
    for cin, cout, hw in SHAPES:
        x = ops.nhwc_empty(n, cin, hw, hw, dt, 'cuda').normal_()
        wt = (torch.randn(cout, 3, 3, cin, device='cuda') * 0.05).to(dt)
        b = torch.randn(cout, device='cuda')
        y = ops.nhwc_empty(n, cout, hw, hw, dt, 'cuda')
        fn = lambda: ops._conv_launch(x, wt.data_ptr(), b, y, None, None, None, 0.0, n, hw, hw, cin, cout, cout, 3, 1, dt)
        ops.set_tuning(21, 8); ops.set_tuning(0, -1)
        t8 = timeit(fn)
        ops.set_tuning(21, 0); ops.set_tuning(0, 2)
        t16 = timeit(fn)
        ops.set_tuning(21, -1); ops.set_tuning(0, -1)
        td = timeit(fn)
        cot = (cout + 127) // 128
        i16 = n * (hw // 16) * (hw // 32) * cot
        print("n=%2d %3d->%3d @%3d: pp8 %6.1f us (%4d items) tall16 %6.1f us (%4d items) default %6.1f  ratio t16/t8 per round %.2f" %
              (n, cin, cout, hw, t8, 2 * i16, t16, i16, td, (t16 / ((i16 + 255) // 256)) / (t8 / ((2 * i16 + 255) // 256))), flush=True)
