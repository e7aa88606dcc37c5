"""Where the step's torch-side fills / copies / adds come from: one eager step under torch.profiler with python stacks."""
import sys, collections
sys.path.insert(0, '.')
import torch
from torch.profiler import profile, ProfilerActivity
import semantic_pyramid_for_image_generation_amd as sp
from semantic_pyramid_for_image_generation_amd import ops, params, synthetic
B = 20
dev = torch.device("cuda", 0)
ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
G = sp.Generator(channels_factor=1).to(dev); D = sp.Discriminator(channel_factor=1).to(dev)
V = sp.VGG16(); V.load_state_dict(params.synth_state_dict(V.state_dict(), 2)); V.to(dev).eval()
og = torch.optim.Adam(G.parameters(), lr=1e-5); od = torch.optim.Adam(D.parameters(), lr=1e-5)
mw = sp.ModelWrapper(G, D, None, None, vgg16=V, generator_optimizer=og, discriminator_optimizer=od, save_data_path=None)
G.train(); D.train()
images, labels, masks = synthetic.synthetic_batch(B, 1234)
images, labels, masks = images.to(dev), labels.to(dev), [m.to(dev) for m in masks]
for _ in range(3): mw.train_step(images, labels, masks)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    mw.train_step(images, labels, masks)
    torch.cuda.synchronize()
WANT = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::zeros", "aten::clone", "aten::mul", "aten::sum", "aten::ones_like", "aten::cat")
agg = collections.Counter()
for e in prof.events():
    if e.name in WANT and e.cpu_parent is not None and e.cpu_parent.name not in WANT:
        st = [s for s in (e.stack or []) if "semantic_pyramid" in s or "torch/optim" in s or "autograd" in s][:3]
        par = e.cpu_parent.name if e.cpu_parent is not None else "-"
        agg[(e.name, str(e.input_shapes)[:60], par[:50], " <- ".join(s.split("/")[-1][:60] for s in st))] += 1
for k, n in sorted(agg.items(), key=lambda x: -x[1])[:70]:
    print("%3d  %-14s %-60s parent=%-50s %s" % ((n,) + k))
