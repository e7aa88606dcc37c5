"""Same-box A/B of two builds of libsempyr.so (SEMPYR_LIB=...) on the step's fat 3x3 launches: forward / input gradient through
sp_conv2d_igemm and the weight gradient, bf16, batch 20.  Prints us per launch (best of 5 x 20 back-to-back launches)."""
import sys, ctypes
sys.path.insert(0, '.')
import torch
from semantic_pyramid_for_image_generation_amd import ops, _lib as L
dt = torch.bfloat16
SHAPES = [(256, 256, 64, 64), (128, 128, 128, 128), (64, 64, 256, 256), (512, 512, 32, 32), (128, 256, 64, 64), (64, 128, 128, 128), (512, 512, 16, 16), (256, 256, 32, 32)]
def timeit(fn, reps=20, rounds=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
tot_f = tot_w = 0.0
for cin, cout, h, w in SHAPES:
    n = 20
    x = ops.nhwc_empty(n, cin, h, w, dt, 'cuda').normal_()
    wt = (torch.randn(cout, 3, 3, cin, device='cuda') * 0.05).to(dt)
    b = torch.randn(cout, device='cuda')
    y = ops.nhwc_empty(n, cout, h, w, dt, 'cuda')
    r1 = ops.nhwc_empty(n, cout, h, w, dt, 'cuda').normal_()
    tf = timeit(lambda: ops._conv_launch(x, wt.data_ptr(), b, y, None, None, None, 0.0, n, h, w, cin, cout, cout, 3, 1, dt))
    tr = timeit(lambda: ops._conv_launch(x, wt.data_ptr(), b, y, r1, None, None, 0.0, n, h, w, cin, cout, cout, 3, 0, dt))
    dy = ops.nhwc_empty(n, cout, h, w, dt, 'cuda').normal_()
    ndw = cout * 9 * cin
    buf = torch.zeros(ndw + cout + 8, dtype=torch.float32, device='cuda')
    wsf = ops.wgrad_workspace_floats(n, h, w, cin, cout, 3, dt)
    ws = torch.empty(max(wsf, 1), dtype=torch.float32, device='cuda')
    tw = timeit(lambda: L.call("sp_conv2d_wgrad_accum", ops.ptr(x), ops.ptr(dy), ops.ptr(buf), ctypes.c_void_p(buf.data_ptr() + 4 * (ndw + 4)),
                               ops.ptr(ws) if wsf else None, wsf, n, h, w, cin, cout, cout, 3, L.SP_BF16, ops.stream()))
    gf = 2.0 * n * h * w * cin * cout * 9 / 1e9
    print("%4d->%4d @%3d^2: fwd(lrelu) %6.1f us %6.0f TF | fwd(+res) %6.1f us | wgrad %6.1f us %6.0f TF" % (cin, cout, h, tf, gf / tf * 1e3, tr, tw, gf / tw * 1e3), flush=True)
    tot_f += tf + tr; tot_w += tw
print("sum fwd %.1f us, wgrad %.1f us" % (tot_f, tot_w))
