import os, sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import golden_util as gu
import semantic_pyramid_for_image_generation_amd as sp
from semantic_pyramid_for_image_generation_amd import ops
ops.set_compute_dtype(torch.float32)
batches = gu.golden_batches(4, 1)
noise = torch.randn(4, 4, 128, generator=torch.Generator().manual_seed(2)).cuda()
def build():
    Gsd, Dsd, Vsd = gu.synth_states({"cf": 4, "seed": 1})
    G = sp.Generator(channels_factor=4); D = sp.Discriminator(channel_factor=4); V = sp.VGG16()
    G.load_state_dict(Gsd); D.load_state_dict(Dsd); V.load_state_dict(Vsd)
    return G.cuda(), D.cuda(), V.cuda().eval()
def step(mw, it):
    images, labels, masks = batches[it]
    return mw.train_step(images.cuda(), labels.cuda(), [m.cuda() for m in masks], noise_d=noise[2*it], noise_g=noise[2*it+1])
res = {}
for kind in ("sp", "torch", "torch_nf"):
    G, D, V = build()
    if kind == "sp":
        og, od = sp.optim.Adam(G.parameters(), lr=1e-4), sp.optim.Adam(D.parameters(), lr=1e-4)
    else:
        kw = {"foreach": False} if kind == "torch_nf" else {}
        og, od = torch.optim.Adam(G.parameters(), lr=1e-4, **kw), torch.optim.Adam(D.parameters(), lr=1e-4, **kw)
    mw = sp.ModelWrapper(generator=G, discriminator=D, vgg16=V, training_dataset=None, validation_dataset=None, generator_optimizer=og, discriminator_optimizer=od, save_data_path=None)
    G.train(); D.train()
    grads = {}
    orig = og.step
    def spy(*a, **k):
        grads["g"] = {n: p.grad.detach().clone() for n, p in G.named_parameters() if p.grad is not None}
        return orig(*a, **k)
    step(mw, 0)
    og.step = spy
    step(mw, 1)
    res[kind] = ({k: v.detach().clone() for k, v in G.state_dict().items()}, grads["g"])
for other in ("torch", "torch_nf"):
    worst = []
    for k in res["sp"][0]:
        a, b = res["sp"][0][k].float(), res[other][0][k].float()
        worst.append((float((a-b).abs().max())/1e-4, k))
    worst.sort(reverse=True)
    print(other, "state diff / lr:", worst[:5])
    gw = []
    for k in res["sp"][1]:
        a, b = res["sp"][1][k], res[other][1][k]
        gw.append((float((a-b).abs().max()), k))
    gw.sort(reverse=True)
    print(other, "grad diff:", gw[:3])
