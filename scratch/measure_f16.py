"""fp16 storage mode: golden-step errors for several loss scales, activation-gradient range probe, quick throughput."""
import sys, json, time
sys.path.insert(0, '.')
import torch
import bench
from semantic_pyramid_for_image_generation_amd import ops
dev = torch.device("cuda", 0)
for name in ("bf16", "fp16"):
    for scale in ((65536.0,) if name == "bf16" else (1.0, 256.0, 65536.0, 2.0 ** 20)):
        ops.set_loss_scale(scale)
        try:
            r = bench.bf16_parity_record(dev, dtype_name=name)
            print(name, "scale", scale, {k: r[k] for k in ("worst_loss_rel_err", "worst_pixel_abs_err", "pixel_rms_err")}, flush=True)
        except Exception as e:
            print(name, "scale", scale, "ERROR", type(e).__name__, e, flush=True)
ops.set_loss_scale(65536.0)
# range probe: max |dz| of every dgrad launch in one fp16 step at batch 20
orig = ops._conv_launch
stats = []
def spy(x, *a, **k):
    stats.append((float(x.float().abs().max()), float((x.float().abs() > 0).float().mean()), bool(torch.isfinite(x.float()).all())))
    return orig(x, *a, **k)
job = bench.Job(1, 20, "fp16", dev, 1, 0, use_graphs=False)
job.eager_step(); torch.cuda.synchronize()
ops._conv_launch = spy
out = job.eager_step(); torch.cuda.synchronize()
ops._conv_launch = orig
print("launches", len(stats), "max |x| over launches", max(s[0] for s in stats), "all finite", all(s[2] for s in stats))
print("losses", {k: float(v) for k, v in out.items() if k.startswith("loss")})
job.close()
for name in ("bf16", "fp16"):
    rec = bench.sub_record(1, 20, name, dev, 20, 5, True)
    print(name, rec["value"], rec["ms_per_step"], flush=True)
