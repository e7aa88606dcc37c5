import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, golden_util as gu
from oracle import sempyr_oracle as O
import semantic_pyramid_for_image_generation_amd as sp
from semantic_pyramid_for_image_generation_amd import ops
meta, _ = gu.load("step_cf1_b2_seed0")
_, _, Vsd = gu.synth_states(meta)
oV = O.make_state(Vsd, frozen=True)
for bseed in (5, 6):
    images, _, masks = gu.golden_batches(2, bseed)[0]
    with torch.no_grad():
        real_ref = O.vgg16_forward(oV, images.flip(0))
    img_ref = images.clone().requires_grad_(True)
    fr = O.vgg16_forward(oV, img_ref)
    lref = O.semantic_reconstruction_loss(real_ref, fr, masks); lref.backward()
    for mode in ("bf16", "fp8-1", "fp8-2"):
        ops.set_compute_dtype(torch.float32 if mode == "f32" else torch.bfloat16); ops.set_vgg_fp8({"bf16": 0, "fp8-1": 1, "fp8-2": 2}[mode])
        V = sp.VGG16(); V.load_state_dict(Vsd); V.cuda().eval()
        loss_fn = sp.SemanticReconstructionLoss()
        for _ in range(3):
            with torch.no_grad():
                real = V(images.flip(0).cuda())
            x = images.cuda().requires_grad_(True)
            feats = V(x)
            loss = loss_fn(real, feats, [m.cuda() for m in masks]); loss.backward()
        g, r = x.grad.float().cpu(), img_ref.grad
        cos = float((g * r).sum() / (g.norm() * r.norm()))
        print("batch seed %d mask sums %s | %s: loss %.5f vs %.5f | grad rel-L2 %.4f cosine %.4f norm ratio %.3f" % (bseed, [int(m.sum()) for m in masks], mode, float(loss), float(lref), float((g - r).norm() / r.norm()), cos, float(g.norm() / r.norm())))
