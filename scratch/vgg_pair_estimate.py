"""What would one 40-image VGG pass (fake with gradient + next real batch) cost against today's two 20-image passes?  Convolution /
pooling part only (the classifier is taken out: it would run per group either way)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semantic_pyramid_for_image_generation_amd as sp
from semantic_pyramid_for_image_generation_amd import ops, params
DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "f32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ops.set_compute_dtype(DT)
V = sp.VGG16()
V.load_state_dict(params.synth_state_dict(V.state_dict(), 2))
V.cuda().eval()
real_launch = ops.linear_launch
ops.linear_launch = lambda *a, **k: None
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
img20 = torch.rand(B, 3, 256, 256, device="cuda") * 2 - 1
img40 = torch.rand(2 * B, 3, 256, 256, device="cuda") * 2 - 1
imgb = torch.rand(B, 3, 256, 256, device="cuda") * 2 - 1
def nograd20():
    with torch.no_grad(): V(img20)
def grad20():
    V(img20.clone().requires_grad_(True))
def grad40():
    V(img40.clone().requires_grad_(True))
def pair():
    V.forward_pair(img20.clone().requires_grad_(True), imgb)
a, b, c, d = timeit(nograd20), timeit(grad20), timeit(grad40), timeit(pair)
print("%s B=%d: no-grad B: %.0f us   grad B: %.0f us   sum %.0f us   |   grad 2B: %.0f us   forward_pair: %.0f us   -> saves %.0f us per step" % (sys.argv[1:] , B, a, b, a + b, c, d, a + b - d))
