"""fp8 VGG chain: errors of the seven pyramid taps and of the input gradient vs the fp32 oracle (bf16 mode beside it), and the
cf=1 golden step in fp8 mode; step time with / without."""
import sys, os; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import golden_util as gu
from oracle import sempyr_oracle as O
import semantic_pyramid_for_image_generation_amd as sp
from semantic_pyramid_for_image_generation_amd import ops, params, synthetic
meta, arr = gu.load("step_cf1_b2_seed0")
Gsd, Dsd, Vsd = gu.synth_states(meta)
images, labels, masks = gu.golden_batches(2, 5)[0]
oV = O.make_state(Vsd, frozen=True)
img_ref = images.clone().requires_grad_(True)
fr = O.vgg16_forward(oV, img_ref)
gens = [torch.randn_like(f) for f in fr]
sum((f * g).sum() for f, g in zip(fr, gens)).backward()
for mode in ("bf16", "fp8"):
    ops.set_compute_dtype(torch.bfloat16); ops.set_vgg_fp8(1 if mode == "fp8" else 0)
    V = sp.VGG16(); V.load_state_dict(Vsd); V.cuda().eval()
    for rep in range(3):                      # call 1 calibrates, 2-3 run on delayed scales
        x = images.cuda().requires_grad_(True)
        feats = V(x)
        sum((f.float() * g.cuda()).sum() for f, g in zip(feats, gens)).backward()
    errs = [float((f.float().cpu() - r).norm() / r.norm()) for f, r in zip(feats, fr)]
    gerr = float((x.grad.float().cpu() - img_ref.grad).norm() / img_ref.grad.norm())
    print(mode, "tap rel-L2:", " ".join("%.4f" % e for e in errs), "| input-gradient rel-L2 %.4f" % gerr)
# golden step
sys.path.insert(0, 'tests')
for mode in ("bf16", "fp8"):
    ops.set_compute_dtype(torch.bfloat16); ops.set_vgg_fp8(1 if mode == "fp8" else 0)
    G = sp.Generator(channels_factor=1); D = sp.Discriminator(channel_factor=1); V = sp.VGG16()
    G.load_state_dict(Gsd); D.load_state_dict(Dsd); V.load_state_dict(Vsd)
    G.cuda().train(); D.cuda().train(); V.cuda().eval()
    mw = sp.ModelWrapper(G, D, None, None, vgg16=V, generator_optimizer=torch.optim.Adam(G.parameters(), lr=meta["lr"]),
                         discriminator_optimizer=torch.optim.Adam(D.parameters(), lr=meta["lr"]), save_data_path=None)
    noise = torch.from_numpy(arr["noise"]).cuda()
    if mode == "fp8":                          # calibration pass on the first batch (not a training step)
        with torch.no_grad(): V(gu.golden_batches(2, meta["seed"])[0][0].cuda())
    names = ("loss_discriminator_real", "loss_discriminator_fake", "loss_generator", "loss_generator_semantic_reconstruction", "loss_generator_diversity")
    pix_idx = gu.fixed_indices(2 * 3 * 256 * 256, gu.N_PIX, 0)
    for it, (im, lb, mk) in enumerate(gu.golden_batches(2, meta["seed"])):
        out = mw.train_step(im.cuda(), lb.cuda(), [m.cuda() for m in mk], noise_d=noise[2 * it], noise_g=noise[2 * it + 1])
        rel = {n: abs(float(out[n]) - meta[n][it]) / max(abs(meta[n][it]), 2e-2) for n in names}
        fake = out["images_fake"].float().cpu().contiguous().flatten()[pix_idx].numpy()
        ref = arr["fake_samples"][2 * it + 1]
        print(mode, "it", it, " ".join("%s %.4f" % (k.replace("loss_", "")[:14], v) for k, v in rel.items()), "| pix max %.4f rms %.5f" % (np.abs(fake - ref).max(), np.sqrt(np.mean((fake - ref) ** 2))))
ops.set_vgg_fp8(False)
