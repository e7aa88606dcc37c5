"""BASELINE.json config 5 - the fp8 MFMA convolution path (include/sempyr.h: SP_F8), built as one vertical slice: the frozen VGG-16
pyramid's 3x3 layers with Cout > 64 on maps >= 32 wide (8 of its 13 convolutions, 79 % of its FLOPs; /root/reference/models.py:183-216)
run BOTH forward passes of a step on v_mfma_f32_16x16x32_fp8_fp8 with OCP e4m3 operands - filters quantised once with one scale
per output channel, activations re-quantised by the producing epilogue with a per-tensor scale that follows the previous pass's
maximum (delayed scaling) - while 16-bit (bf16) copies are still written wherever the pyramid taps or the backward pass read them;
the backward pass itself stays bf16.  Tolerances are RESTATED for this mode and are 2x the errors MEASURED on MI355X
(scratch/measure_f8.py; e4m3 has 3 mantissa bits, the reference is fp32 end to end):
    pyramid taps vs the fp32 oracle      rel-L2 0.003 / 0.044 / 0.067 / 0.083 / 0.059 / 0.043 / 0.046   (bf16 mode: 0.003 .. 0.007)
    cf=1 golden step (reference losses)  worst loss 1.4e-2 relative; generator pixels 0.171 worst sample, 0.031 rms
    reconstruction-loss gradient         see GRAD_MEASURED below (the weak spot: cosine 0.45 / 0.18 with the fp32 gradient)
ops.set_vgg_fp8(1) keeps the pass WITH gradient in bf16 (recommended), (2) runs both passes in e4m3.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import golden_util as gu  # noqa: E402
from oracle import sempyr_oracle as O  # noqa: E402
import semantic_pyramid_for_image_generation_amd as sp  # noqa: E402
from semantic_pyramid_for_image_generation_amd import _lib as L, ops  # noqa: E402

LOSS_NAMES = ("loss_discriminator_real", "loss_discriminator_fake", "loss_generator",
              "loss_generator_semantic_reconstruction", "loss_generator_diversity")
TAP_MEASURED = (0.0031, 0.0435, 0.0672, 0.0832, 0.0590, 0.0426, 0.0457)


@pytest.fixture(autouse=True)
def _reset():
    yield
    ops.set_vgg_fp8(False)
    ops.set_compute_dtype(torch.float32)


def test_fp8_quantiser_and_weight_packer_are_ocp_e4m3():
    """The bytes are OCP e4m3fn (gfx950), round-to-nearest-even, saturating at +-448: equal to torch.float8_e4m3fn."""
    g = torch.Generator(device="cuda").manual_seed(0)
    x = ops.nhwc_empty(3, 64, 32, 32, torch.bfloat16, "cuda")
    x.normal_(generator=g)
    x.mul_(40.0)                                                      # some values saturate
    inv = torch.full((1,), 3.7, device="cuda")
    amax = torch.zeros(1, device="cuda")
    q = ops.quantize_fp8(x, inv, amax)
    want = (x.float() * inv).clamp(-448, 448).to(torch.float8_e4m3fn)
    assert torch.equal(q.permute(0, 2, 3, 1).contiguous(), want.permute(0, 2, 3, 1).contiguous().view(torch.uint8))
    assert float(amax) == float(x.float().abs().max())
    w = torch.randn(96, 80, 3, 3, device="cuda", generator=g) * 0.05
    w8, sw, cin_p = ops.pack_weight_fp8(w)
    swr = w.abs().amax(dim=(1, 2, 3)) / 448.0
    assert cin_p == 80 and float((sw - swr).abs().max()) <= 1e-9
    wq = (w / swr[:, None, None, None]).clamp(-448, 448).to(torch.float8_e4m3fn).permute(0, 2, 3, 1).reshape(96, 9, 80).view(torch.uint8)
    assert float((w8.view(96, 9, 80) == wq).float().mean()) >= 0.9999   # (a division vs a reciprocal-multiply differ by an ulp on a handful)


@pytest.mark.parametrize("case", [(2, 64, 128, 64, 0), (3, 128, 128, 32, 2), (2, 256, 512, 32, 0), (20, 256, 256, 64, 2), (1, 80, 192, 32, 0)])
def test_fp8_convolution_vs_dequantised_fp32_reference(case):
    """sp_conv2d_igemm with dtype SP_F8 against conv2d on the DEQUANTISED operands in fp32: fp8 x fp8 products are exact in fp32,
    so what remains is summation order and the bf16 rounding of the output (4e-3); the e4m3 output codes may differ on
    rounding ties only (< 0.2 % of the elements, by one code)."""
    n, cin, cout, hw, pool2 = case
    g = torch.Generator(device="cuda").manual_seed(1)
    x = ops.nhwc_empty(n, cin, hw, hw, torch.bfloat16, "cuda")
    x.normal_(generator=g)
    x.abs_()
    sx = (x.float().abs().max() / 448.0).reshape(1)
    x8 = ops.quantize_fp8(x, 1.0 / sx)
    w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * 0.05
    w8, sw, cin_p = ops.pack_weight_fp8(w)
    bias = torch.randn(cout, device="cuda", generator=g)
    xd = (x.float() / sx).clamp(-448, 448).to(torch.float8_e4m3fn).float() * sx
    wd = (w / sw[:, None, None, None]).clamp(-448, 448).to(torch.float8_e4m3fn).float() * sw[:, None, None, None]
    ref = F.relu(F.conv2d(xd, wd, bias, padding=1))
    if pool2:
        ref = F.max_pool2d(ref, 2)
    ho = hw // 2 if pool2 else hw
    y = ops.nhwc_empty(n, cout, ho, ho, torch.bfloat16, "cuda")
    y8 = torch.empty((n, ho, ho, cout), dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2)
    sy = (ref.abs().max() / 448.0).reshape(1)
    amax = torch.zeros(1, device="cuda")
    ops.conv_launch_f8(x8, w8, sw, sx, bias, y, y8, 1.0 / sy, amax, n, hw, hw, cin_p, cout, ops.ACT_RELU, pool2)
    torch.cuda.synchronize()
    assert float((y.float() - ref).abs().max() / ref.abs().max()) <= 6e-3
    assert float(amax) == pytest.approx(float(ref.abs().max()), rel=1e-4)
    got8 = y8.permute(0, 2, 3, 1).contiguous().view(torch.float8_e4m3fn).float().permute(0, 3, 1, 2)
    want8 = (ref / sy).clamp(-448, 448).to(torch.float8_e4m3fn).float()
    assert float(((got8 - want8).abs() > 0).float().mean()) <= 2e-3
    with pytest.raises(L.SempyrError):                                    # outside the slice: rejected loudly, no silent fallback
        ops.conv_launch_f8(x8, w8, sw, sx, bias, y, None, None, None, n, hw, hw, cin_p, 64, ops.ACT_RELU, 0)


@pytest.mark.parametrize("case", [(1, 80, 192, 32, 0), (2, 64, 320, 32, 2)])
def test_fp8_partial_co_tile_repeated_launches(case):
    """Cout that is not a multiple of the 128-channel tile, launched repeatedly into a dirty output: the channel groups of the
    last tile that lie past Cout must not be stored (round 3: they landed on channels 0..63 of the next pixel and raced with the
    block that owns them - right on a first launch, wrong on most later ones; found as one flaky failure of the test above)."""
    n, cin, cout, hw, pool2 = case
    g = torch.Generator(device="cuda").manual_seed(3)
    x = ops.nhwc_empty(n, cin, hw, hw, torch.bfloat16, "cuda")
    x.normal_(generator=g)
    sx = (x.float().abs().max() / 448.0).reshape(1)
    x8 = ops.quantize_fp8(x, 1.0 / sx)
    w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * 0.05
    w8, sw, cin_p = ops.pack_weight_fp8(w)
    bias = torch.randn(cout, device="cuda", generator=g)
    xd = (x.float() / sx).clamp(-448, 448).to(torch.float8_e4m3fn).float() * sx
    wd = (w / sw[:, None, None, None]).clamp(-448, 448).to(torch.float8_e4m3fn).float() * sw[:, None, None, None]
    ref = F.relu(F.conv2d(xd, wd, bias, padding=1))
    if pool2:
        ref = F.max_pool2d(ref, 2)
    ho = hw // 2 if pool2 else hw
    for rep in range(25):
        y = ops.nhwc_empty(n, cout, ho, ho, torch.bfloat16, "cuda")
        y.fill_(-7.0)
        ops.conv_launch_f8(x8, w8, sw, sx, bias, y, None, None, None, n, hw, hw, cin_p, cout, ops.ACT_RELU, pool2)
        torch.cuda.synchronize()
        assert float((y.float() - ref).abs().max() / ref.abs().max()) <= 6e-3, rep


# measured on MI355X (two golden batches): reconstruction loss value and its gradient w.r.t. the image, against the fp32 oracle
#   storage / slice        loss rel. error   gradient cosine with fp32   gradient rel-L2
#   bf16                   4e-4              0.84 - 0.86                  0.53 - 0.55     (sign-like L1 loss + ReLU / max-pool routing:
#   bf16 + fp8 slice       6 - 12 %          0.45 - 0.48                  1.02 - 1.07      already bf16 storage moves this gradient)
#   fp16 / fp16 + slice    measured by the test below and written to gpurun_out/fp8_gradient_rule.json
# (round 3's "mode 2" - e4m3 in the pass WITH gradient: cosine 0.18 - was removed from the product)
GRAD_MEASURED = (0.12, 0.45, 1.07)


def _rec_gradient(dtype, fp8, images, masks, Vsd):
    ops.set_compute_dtype(dtype)
    ops.set_vgg_fp8(1 if fp8 else 0)
    V = sp.VGG16()
    V.load_state_dict(Vsd)
    V.cuda().eval()
    loss_fn = sp.SemanticReconstructionLoss()
    for _ in range(3):                                  # call 1 calibrates the activation scales in 16 bit, calls 2-3 run on delayed scales
        with torch.no_grad():
            real = V(images.flip(0).cuda())
        x = images.cuda().requires_grad_(True)
        feats = V(x)
        loss = loss_fn(real, feats, [m.cuda() for m in masks])
        loss.backward()
    if fp8:
        assert V._packs["f8"]["calibrated"]
    return real, float(loss), x.grad.float().cpu()


def test_fp8_vgg_pyramid_vs_oracle_restated_tolerance():
    """The seven pyramid taps of the fp8 pass against the fp32 oracle (bound: 2x the measured rel-L2 of every tap) and the
    reconstruction loss of lossfunction.py:31-68 with its gradient w.r.t. the image (loss 2x measured, cosine >= 2/3 of the
    measured one, rel-L2 <= 1.25x)."""
    meta, _ = gu.load("step_cf1_b2_seed0")
    _, _, Vsd = gu.synth_states(meta)
    images, _, masks = gu.golden_batches(2, 5)[0]
    oV = O.make_state(Vsd, frozen=True)
    with torch.no_grad():
        real_ref = O.vgg16_forward(oV, images.flip(0))
    img_ref = images.clone().requires_grad_(True)
    loss_ref = O.semantic_reconstruction_loss(real_ref, O.vgg16_forward(oV, img_ref), masks)
    loss_ref.backward()
    real, loss, g = _rec_gradient(torch.bfloat16, True, images, masks, Vsd)
    errs = [float((f.detach().float().cpu() - r.detach()).norm() / r.detach().norm()) for f, r in zip(real, real_ref)]
    print("fp8 taps rel-L2 %s" % " ".join("%.4f" % e for e in errs))
    for i, err in enumerate(errs):
        assert err <= 2 * TAP_MEASURED[i], (i, err)
    assert max(errs[1:]) > 0.02, "the e4m3 path did not run (errors look like bf16)"
    r = img_ref.grad
    lerr = abs(loss - float(loss_ref)) / float(loss_ref)
    cos = float((g * r).sum() / (g.norm() * r.norm()))
    rel = float((g - r).norm() / r.norm())
    print("fp8 slice: rec loss rel err %.4f, gradient cosine %.4f rel-L2 %.4f" % (lerr, cos, rel))
    m_loss, m_cos, m_rel = GRAD_MEASURED
    assert lerr <= 2 * m_loss and cos >= m_cos * 2 / 3 and rel <= 1.25 * m_rel, (lerr, cos, rel)


def test_fp8_slice_is_held_to_the_gradient_rule():
    """A bound that can fail (round-3 VERDICT: the old bounds accepted anything that was not anti-correlated).  RULE: a layer set may
    run in e4m3 by default only if the reconstruction-loss gradient w.r.t. the image keeps a cosine with the fp32 gradient of at
    least 0.9 x the cosine of the plain 16-bit storage it replaces; otherwise those layers stay 16-bit and the slice is an opt-in
    measurement mode.  The test measures all four (bf16, fp16, each with the slice), records them, and asserts the product's
    default against the rule - today the slice does NOT meet it (e4m3's 4 significant bits on random-sign dot products: 4-8 %
    noise per tap), so config.CFG.vgg_fp8 must default to 0 and fp16 storage alone is what BASELINE config 5 runs as."""
    import json
    import os
    from semantic_pyramid_for_image_generation_amd.config import Config
    meta, _ = gu.load("step_cf1_b2_seed0")
    _, _, Vsd = gu.synth_states(meta)
    images, _, masks = gu.golden_batches(2, 5)[0]
    oV = O.make_state(Vsd, frozen=True)
    with torch.no_grad():
        real_ref = O.vgg16_forward(oV, images.flip(0))
    img_ref = images.clone().requires_grad_(True)
    O.semantic_reconstruction_loss(real_ref, O.vgg16_forward(oV, img_ref), masks).backward()
    r = img_ref.grad
    rec = {}
    for name, dtype, fp8 in (("bf16", torch.bfloat16, False), ("bf16+fp8", torch.bfloat16, True), ("fp16", torch.float16, False),
                             ("fp16+fp8", torch.float16, True)):
        _, _, g = _rec_gradient(dtype, fp8, images, masks, Vsd)
        rec[name] = float((g * r).sum() / (g.norm() * r.norm()))
    print("reconstruction-loss gradient cosine with fp32: %s" % json.dumps(rec))
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        json.dump(rec, open(os.path.join("gpurun_out", "fp8_gradient_rule.json"), "w"))
    except OSError:
        pass
    assert rec["fp16"] >= 0.97 and rec["fp16"] >= rec["bf16"], rec              # 16-bit storage itself: fp16 is the better of the two
    for base in ("bf16", "fp16"):
        meets = rec[base + "+fp8"] >= 0.9 * rec[base]
        assert meets or Config().vgg_fp8 == 0, ("the fp8 slice is on by default but fails the gradient rule", rec)


def test_fp8_mode_train_step_vs_reference_golden():
    """The cf=1 golden step (reference losses and pixels, tests/golden/step_cf1_b2_seed0) with the fp8 VGG chain: 2x measured."""
    meta, arr = gu.load("step_cf1_b2_seed0")
    Gsd, Dsd, Vsd = gu.synth_states(meta)
    ops.set_compute_dtype(torch.bfloat16)
    ops.set_vgg_fp8(1)                                                     # e4m3 in the no-gradient pass
    G, D, V = sp.Generator(channels_factor=1), sp.Discriminator(channel_factor=1), sp.VGG16()
    G.load_state_dict(Gsd); D.load_state_dict(Dsd); V.load_state_dict(Vsd)
    G.cuda().train(); D.cuda().train(); V.cuda().eval()
    mw = sp.ModelWrapper(G, D, None, None, vgg16=V, generator_optimizer=torch.optim.Adam(G.parameters(), lr=meta["lr"]),
                         discriminator_optimizer=torch.optim.Adam(D.parameters(), lr=meta["lr"]), save_data_path=None)
    batches = gu.golden_batches(meta["batch_size"], meta["seed"])
    with torch.no_grad():
        V(batches[0][0].cuda())                                            # calibration pass (bf16, records the activation maxima)
    noise = torch.from_numpy(arr["noise"]).cuda()
    pix_idx = gu.fixed_indices(meta["batch_size"] * 3 * 256 * 256, gu.N_PIX, 0)
    for it, (im, lb, mk) in enumerate(batches):
        out = mw.train_step(im.cuda(), lb.cuda(), [m.cuda() for m in mk], noise_d=noise[2 * it], noise_g=noise[2 * it + 1])
        worst = max(abs(float(out[n]) - meta[n][it]) / max(abs(meta[n][it]), 2e-2) for n in LOSS_NAMES)
        fake = out["images_fake"].float().cpu().contiguous().flatten()[pix_idx].numpy()
        ref = arr["fake_samples"][2 * it + 1]
        print("fp8 step it %d: worst loss rel %.4f, pixel max %.4f rms %.5f" % (it, worst, np.abs(fake - ref).max(), np.sqrt(np.mean((fake - ref) ** 2))))
        assert worst <= 2 * 1.4e-2, (it, worst)
        assert np.abs(fake - ref).max() <= 2 * 0.171 and np.sqrt(np.mean((fake - ref) ** 2)) <= 2 * 0.031, it
