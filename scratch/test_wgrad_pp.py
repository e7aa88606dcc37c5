"""Ping-pong row walker vs the round-2 row walker (SP_WGRAD_PP 0 / 1): agreement of dW / dbias (incl. fp32 torch reference on small
shapes), repeated runs (race screen: bit-identical with slabs), timing on the step's 3x3 shapes (bf16, B = 20)."""
import sys, os, ctypes
sys.path.insert(0, '.')
import torch, torch.nn.functional as F
from semantic_pyramid_for_image_generation_amd import ops, _lib as L
lib = L.lib(); dt = torch.bfloat16
def run(x, dy, n, hw, cin, cout, cp, pp, up=False):
    ndw = cout * 9 * cin
    buf = torch.zeros(ndw + cout + 8, dtype=torch.float32, device='cuda')
    wsf = ops.wgrad_workspace_floats(n, hw, hw, cin, cout, 3, dt)
    ws = torch.empty(max(wsf, 1), dtype=torch.float32, device='cuda')
    lib.sp_set_tuning(23, pp)
    L.call("sp_conv2d_wgrad_accum_pooled" if up else "sp_conv2d_wgrad_accum", ops.ptr(x), ops.ptr(dy), ops.ptr(buf), ctypes.c_void_p(buf.data_ptr() + 4 * (ndw + 4)),
           ops.ptr(ws) if wsf else None, wsf, n, hw, hw, cin, cout, cp, 3, L.SP_BF16, ops.stream())
    torch.cuda.synchronize()
    lib.sp_set_tuning(23, -1)
    return buf[:ndw].clone(), buf[ndw + 4:ndw + 4 + cout].clone()
ok_all = True
for n, cin, cout, hw, up in [(2, 64, 64, 32, False), (3, 128, 64, 64, False), (20, 256, 256, 64, False), (20, 128, 128, 128, False), (5, 72, 64, 64, False),
                             (20, 264, 256, 32, False), (4, 64, 128, 64, True), (20, 64, 64, 256, False), (1, 64, 64, 32, False), (7, 96, 136, 96, False)]:
    g = torch.Generator(device='cuda').manual_seed(n + cin)
    x = ops.nhwc_empty(n, cin, hw, hw, dt, 'cuda'); x.normal_(generator=g)
    cp = (cout + 7) // 8 * 8
    hd = hw // 2 if up else hw
    dy = ops.nhwc_empty(n, cp, hd, hd, dt, 'cuda'); dy.normal_(generator=g)
    d0, b0 = run(x, dy, n, hw, cin, cout, cp, 0, up)
    d1, b1 = run(x, dy, n, hw, cin, cout, cp, 1, up)
    d2, b2 = run(x, dy, n, hw, cin, cout, cp, 1, up)
    e = float((d1 - d0).abs().max() / d0.abs().max()); eb = float((b1 - b0).abs().max() / b0.abs().max())
    rep = torch.equal(d1, d2) and torch.equal(b1, b2)
    ref = ""
    if n * hw * hw <= 20000:
        dyf = dy.float()[:, :cout]
        if up: dyf = F.interpolate(dyf, scale_factor=2, mode='nearest') * 0.25
        w = torch.zeros(cout, cin, 3, 3, device='cuda', requires_grad=True)
        F.conv2d(x.float(), w, padding=1).backward(dyf)
        r = w.grad.permute(0, 2, 3, 1).reshape(-1)
        ref = " | vs torch fp32: pp %.2e old %.2e" % (float((d1 - r).abs().max() / r.abs().max()), float((d0 - r).abs().max() / r.abs().max()))
    good = e < 2e-3 and eb < 2e-3
    ok_all &= good
    print("n=%d %d->%d @%d up=%d: pp vs old dW %.2e dbias %.2e repeat-identical %s%s %s" % (n, cin, cout, hw, up, e, eb, rep, ref, "ok" if good else "FAIL"), flush=True)
print("ALL OK" if ok_all else "SOME FAILED")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
B = 20; tot = [0.0, 0.0, 0.0]
for cin, cout, hw in [(64, 64, 256), (128, 128, 128), (256, 256, 64), (512, 512, 32), (64, 128, 128), (128, 256, 64), (256, 512, 32), (256, 256, 32), (264, 256, 32), (72, 64, 128), (136, 128, 64)]:
    x = ops.nhwc_empty(B, cin, hw, hw, dt, 'cuda'); x.normal_()
    cp = (cout + 7) // 8 * 8
    dy = ops.nhwc_empty(B, cp, hw, hw, dt, 'cuda'); dy.normal_()
    ndw = cout * 9 * cin
    buf = torch.zeros(ndw + cout + 8, dtype=torch.float32, device='cuda')
    wsf = ops.wgrad_workspace_floats(B, hw, hw, cin, cout, 3, dt); ws = torch.empty(max(wsf, 1), dtype=torch.float32, device='cuda')
    call = (lambda: L.call("sp_conv2d_wgrad_accum", ops.ptr(x), ops.ptr(dy), ops.ptr(buf), ctypes.c_void_p(buf.data_ptr() + 4 * (ndw + 4)),
                            ops.ptr(ws) if wsf else None, wsf, B, hw, hw, cin, cout, cp, 3, L.SP_BF16, ops.stream()))
    flops = 2.0 * B * hw * hw * cin * cout * 9
    res = []
    for k, pp in enumerate((0, 2, 1)):
        lib.sp_set_tuning(23, pp)
        t = min(timeit(call) for _ in range(3)); res.append(t); tot[k] += t
    lib.sp_set_tuning(23, -1)
    print("%4d->%4d @%3d  row walker %7.1f us %6.0f TF | pp (alternating steps) %7.1f us %6.0f TF | pp3 (rows in registers) %7.1f us %6.0f TF" % (cin, cout, hw, res[0] * 1e3, flops / res[0] / 1e9, res[1] * 1e3, flops / res[1] / 1e9, res[2] * 1e3, flops / res[2] / 1e9), flush=True)
print("sum row walker %.3f ms pp %.3f ms pp3 %.3f ms" % tuple(tot))
