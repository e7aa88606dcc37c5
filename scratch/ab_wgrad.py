"""A/B micro-benchmark of the weight gradient (sp_conv2d_wgrad_accum incl. its reduce pass) on the step's shapes (bf16, B=20)."""
import sys, os, ctypes
sys.path.insert(0, '.')
import torch
from semantic_pyramid_for_image_generation_amd import ops, _lib as L
B = 20
dt = torch.bfloat16
SHAPES = [(64, 64, 256, 3), (128, 128, 128, 3), (256, 256, 64, 3), (512, 512, 32, 3), (64, 128, 128, 3), (128, 256, 64, 3), (256, 512, 32, 3),
          (256, 256, 32, 3), (8, 64, 256, 3), (264, 256, 32, 3), (72, 64, 128, 3), (512, 512, 16, 3), (256, 256, 16, 3), (512, 512, 8, 3),
          (768, 768, 4, 3), (128, 256, 32, 1), (64, 128, 64, 1), (256, 128, 32, 1), (128, 64, 64, 1), (8, 64, 128, 1), (256, 32, 32, 1), (256, 128, 16, 1),
          (64, 128, 128, 1), (512, 256, 16, 1), (64, 3, 256, 1), (256, 256, 16, 1), (256, 32, 16, 1), (256, 512, 4, 1), (512, 768, 2, 1)]
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
tag = os.path.basename(os.environ.get("SEMPYR_LIB", "current"))
tot = 0.0
for cin, cout, hw, k in SHAPES:
    if os.environ.get('ONLY') and int(os.environ['ONLY']) != k: continue
    x = ops.nhwc_empty(B, cin, hw, hw, dt, 'cuda'); x.normal_()
    cp = (cout + 7) // 8 * 8
    dy = ops.nhwc_empty(B, cp, hw, hw, dt, 'cuda'); dy.normal_()
    ndw = cout * k * k * cin
    buf = torch.zeros(ndw + cout + 8, dtype=torch.float32, device='cuda')
    wsf = ops.wgrad_workspace_floats(B, hw, hw, cin, cout, k, dt)
    ws = torch.empty(max(wsf, 1), dtype=torch.float32, device='cuda')
    flops = 2.0 * B * hw * hw * cin * cout * k * k
    call = (lambda: L.call("sp_conv2d_wgrad_accum", ops.ptr(x), ops.ptr(dy), ops.ptr(buf), ctypes.c_void_p(buf.data_ptr() + 4 * (ndw + 4)),
                              ops.ptr(ws) if wsf else None, wsf, B, hw, hw, cin, cout, cp, k, L.SP_BF16, ops.stream()))
    t = timeit(call)
    tot += t
    err = berr = float('nan')
    if k == 1:
        buf.zero_()
        call()
        ref = dy.permute(0, 2, 3, 1).reshape(-1, cp)[:, :cout].float().t() @ x.permute(0, 2, 3, 1).reshape(-1, cin).float()
        got = buf[:ndw].view(cout, cin)
        err = float((got - ref).abs().max() / ref.abs().max())
        bref = dy.permute(0, 2, 3, 1).reshape(-1, cp)[:, :cout].float().sum(0)
        berr = float((buf[ndw + 4:ndw + 4 + cout] - bref).abs().max() / bref.abs().max())
    print("%-20s %4d->%4d @%3d k%d  wgrad %7.1f us %7.1f TF  err %.1e bias %.1e" % (tag, cin, cout, hw, k, t * 1e3, flops / t / 1e9, err, berr))
print("%-20s sum %.3f ms" % (tag, tot))
