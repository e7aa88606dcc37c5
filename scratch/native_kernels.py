"""Which Python call sites launch torch-native kernels (fills, adds, copies) in one eager training step?  Run on the GPU box:
   python scratch/native_kernels.py > gpurun_out/native_kernels.txt"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"]
import bench  # noqa: E402

dev = torch.device("cuda:0")
job = bench.Job(1, 20, "bf16", dev, 1, 0, use_graphs=False)
for _ in range(3):
    job.eager_step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    job.eager_step()
    torch.cuda.synchronize()

PKG = "semantic_pyramid_for_image_generation_amd"
sites = collections.Counter()
shapes = collections.defaultdict(set)
for ev in prof.events():
    name = ev.name
    if not name.startswith("aten::"):
        continue
    if ev.device_time_total <= 0 and not any(k in name for k in ("fill", "zero", "copy", "add", "mul", "clone", "contiguous", "cat", "sum", "to")):
        continue
    kids = [k for k in ev.cpu_children]
    # only leaf aten ops that own a kernel
    if not ev.kernels:
        continue
    frames = [f for f in (ev.stack or []) if PKG in f or "bench.py" in f]
    site = frames[0] if frames else ((ev.stack or ["?"])[0])
    key = (name, " <- ".join(x.split("/")[-1] for x in frames[:3]) or site)
    sites[key] += len(ev.kernels)
    shapes[key].add(str(ev.input_shapes)[:80])
print("torch-native kernel launches of one eager step, by call site")
tot = 0
for (name, site), c in sorted(sites.items(), key=lambda kv: -kv[1]):
    tot += c
    print("%4d  %-22s %s   %s" % (c, name, site, sorted(shapes[(name, site)])[:2]))
print("total", tot)
