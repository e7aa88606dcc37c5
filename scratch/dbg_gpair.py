"""Diagnostic (round 5): Generator.forward_pair vs two forwards vs the CPU oracle, per-tensor gradient errors (fp32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import golden_util as gu
import semantic_pyramid_for_image_generation_amd as sp
from semantic_pyramid_for_image_generation_amd import ops
from oracle import sempyr_oracle as O

cf, batch = (float(sys.argv[1]) if len(sys.argv) > 1 else 1), (int(sys.argv[2]) if len(sys.argv) > 2 else 6)
cf = int(cf) if cf == int(cf) else cf
ops.set_compute_dtype(torch.float32)
g = torch.Generator().manual_seed(17)
images, labels, masks = gu.golden_batches(4, 5)[0]
reps = (batch + 3) // 4
images = images.repeat(reps, 1, 1, 1)[:batch]
labels = labels.repeat(reps, 1)[:batch]
masks = [m.repeat(reps, *([1] * (m.dim() - 1)))[:batch] for m in masks]
z_d, z_g = torch.randn(batch, 128, generator=g), torch.randn(batch, 128, generator=g)
seed_img = torch.randn(batch, 3, 256, 256, generator=g)
Gsd, _, Vsd = gu.synth_states({"cf": cf, "seed": 3})
V = sp.VGG16(); V.load_state_dict(Vsd); V.cuda().eval()
with torch.no_grad():
    feats = V(images.cuda())
# oracle
oG, oV = O.make_state(Gsd), O.make_state(Vsd, frozen=True)
with torch.no_grad():
    of = O.vgg16_forward(oV, images)
    O.generator_forward(oG, z_d, of, masks, labels.float(), True)
fo = O.generator_forward(oG, z_g, of, masks, labels.float(), True)
fo.backward(seed_img)
names = [k for k, t in oG.items() if t.requires_grad]
ref = {k: oG[k].grad.clone() for k in names}
outs = {}
for mode in ("two", "pair"):
    G = sp.Generator(channels_factor=cf); G.load_state_dict(Gsd); G = G.cuda().train()
    G._bank.direct_grads, G._bank.expected_passes = True, 1
    cm, cl, cms = images.cuda(), labels.cuda(), [m.cuda() for m in masks]
    if mode == "two":
        with torch.no_grad():
            G(z_d.cuda(), feats, cms, cl)
        fake = G(z_g.cuda(), feats, cms, cl)
    else:
        fake, _ = G.forward_pair(z_g.cuda(), z_d.cuda(), feats, cms, cl)
    fake.backward(seed_img.cuda())
    G._bank.collect_extra()
    outs[mode] = ({n: p.grad.detach().float().cpu() for n, p in G.named_parameters()}, fake.detach().float().cpu())
print("image err two-vs-oracle %.2e pair-vs-oracle %.2e pair-vs-two %.2e" % (float((outs["two"][1] - fo.detach()).abs().max()),
      float((outs["pair"][1] - fo.detach()).abs().max()), float((outs["pair"][1] - outs["two"][1]).abs().max())))
rows = []
for n in names:
    r, a, b = ref[n], outs["two"][0][n], outs["pair"][0][n]
    den = float(r.abs().max()) + 1e-30
    rows.append((float((a - b).abs().max()) / den, float((a - r).abs().max()) / den, float((b - r).abs().max()) / den, den, n))
rows.sort(reverse=True)
print("%-50s %10s %10s %10s %10s" % ("tensor", "pair-two", "two-orac", "pair-orac", "max|g|"))
for e in rows[:14]:
    print("%-50s %10.2e %10.2e %10.2e %10.2e" % (e[4], e[0], e[1], e[2], e[3]))
def cos(x, y):
    a = torch.cat([x[n].double().flatten() for n in names]); b = torch.cat([y[n].double().flatten() for n in names])
    return float((a * b).sum() / (a.norm() * b.norm()))
print("cosine two-oracle %.9f pair-oracle %.9f pair-two %.9f" % (cos(outs["two"][0], ref), cos(outs["pair"][0], ref), cos(outs["pair"][0], outs["two"][0])))
