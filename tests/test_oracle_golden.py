"""Pins the CPU oracle (oracle/sempyr_oracle.py) to vectors recorded from the UNMODIFIED reference
driving its own ModelWrapper.train() loop (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import sempyr_oracle as O

TAGS = ["step_cf1_b2_seed0", "step_cf4_b4_seed1"]


@pytest.mark.parametrize("tag", TAGS)
def test_layouts_match_reference_keys(tag):
    meta, _ = gu.load(tag)
    assert list(O.generator_layout(meta["cf"]).keys()) == meta["keys_G"]
    assert list(O.discriminator_layout(meta["cf"]).keys()) == meta["keys_D"]
    assert list(O.vgg16_layout().keys()) == meta["keys_V"]


@pytest.mark.parametrize("tag", TAGS)
def test_train_step_matches_reference_loop(tag):
    meta, arr = gu.load(tag)
    torch.set_num_threads(meta["threads"])
    Gsd, Dsd, Vsd = gu.synth_states(meta)
    gu.check_checksums(Gsd, meta["synth_checksums_G"], what="G synth")
    gu.check_checksums(Dsd, meta["synth_checksums_D"], what="D synth")
    gu.check_checksums(Vsd, meta["synth_checksums_V"], what="V synth")
    G, D, V = O.make_state(Gsd), O.make_state(Dsd), O.make_state(Vsd, frozen=True)
    assert [k for k, t in G.items() if t.requires_grad] == meta["param_names_G"]
    assert [k for k, t in D.items() if t.requires_grad] == meta["param_names_D"]
    opt_g = torch.optim.Adam(O.trainable(G), lr=meta["lr"])
    opt_d = torch.optim.Adam(O.trainable(D), lr=meta["lr"])
    batches = gu.golden_batches(meta["batch_size"], meta["seed"])
    noise = torch.from_numpy(arr["noise"])
    pix_idx = gu.fixed_indices(batches[0][0].numel(), gu.N_PIX, 0)
    names = ("loss_discriminator_real", "loss_discriminator_fake", "loss_generator",
             "loss_generator_semantic_reconstruction", "loss_generator_diversity")
    for it, (images, labels, masks) in enumerate(batches):
        # dead-work skipping must not change any observable: exercise it on iteration 1
        out = O.train_step(G, D, V, opt_g, opt_d, images, labels, masks, noise[2 * it], noise[2 * it + 1],
                           skip_dead_d_wgrad=(it == 1))
        got = (out["loss_d_real"], out["loss_d_fake"], out["loss_g"], out["loss_rec"], out["loss_div"])
        for n, g in zip(names, got):
            assert float(g) == pytest.approx(meta[n][it], rel=2e-5, abs=1e-7), (n, it)
        for j, key in ((2 * it, "images_fake_d"), (2 * it + 1, "images_fake_g")):
            s = out[key].flatten()[pix_idx].numpy()
            ref = arr["fake_samples"][j]
            assert np.abs(s - ref).max() <= 1e-4 * np.abs(ref).max(), (key, it)
        for key, grads in (("grads_d", out["grads_d"]), ("grads_g", out["grads_g"])):
            norms = np.array([float(g.double().norm()) for g in grads])
            ref = arr[key + "_norms"][it]
            # biases in front of a BatchNorm and the key bias have mathematically zero gradients (1e-8 noise)
            assert np.all(np.abs(norms - ref) <= 2e-3 * ref + 1e-6 * ref.max()), (key, it)
            s, rs = gu.grad_samples(grads), arr[key + "_samples"][it]
            assert np.abs(s - rs).max() <= 5e-3 * np.abs(rs).max(), (key, it)
    gu.check_checksums({k: v.detach() for k, v in G.items()}, meta["final_checksums_G"], rtol=2e-4, what="G final",
                       noise_keys=gu.zero_gradient_keys(meta, arr, "grads_g"), noise_atol=2 * meta["lr"])
    gu.check_checksums({k: v.detach() for k, v in D.items()}, meta["final_checksums_D"], rtol=2e-4, what="D final",
                       noise_keys=gu.zero_gradient_keys(meta, arr, "grads_d"), noise_atol=2 * meta["lr"])


def test_discriminator_output_shape_quirk():
    # models.py:151-155: (B,128) * (B,1,128) broadcasts to (B,B,128)
    spec = O.discriminator_layout(8)
    from semantic_pyramid_for_image_generation_amd import params
    D = O.make_state(params.synth_state_dict(O.layout_template(spec), 3))
    x = torch.randn(3, 3, 256, 256)
    labels = torch.nn.functional.one_hot(torch.tensor([1, 5, 9]), 365)
    assert O.discriminator_forward(D, x, labels).shape == (3, 3, 128)


def test_mask_generator_statistics_match_the_reference_generator():
    """Row f1: the device mask generator's decisions (restated bit for bit in oracle.training_mask_decisions; the GPU test
    holds the kernel to the restatement) against the statistics of the UNMODIFIED reference generator
    (tests/golden/mask_stats.json, recorded by tests/golden/make_mask_stats.py from /root/reference/misc.py:13-68 over 45 000
    calls): the joint frequencies of (open stage, spatial mask used) must agree within 4.5 sigma of the two-sample binomial
    spread in every one of the 12 cells the reference produces - and no other cell may occur."""
    import json
    import math
    import os
    from oracle import sempyr_oracle as O
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mask_stats.json")))
    n_ref = ref["n"]
    ref_counts = {(s, bool(sp)): c for s, sp, c in ref["counts"]}
    n = 45000
    counts = {}
    for s, sp in O.training_mask_decisions(n, 20260, 0.3):
        counts[(s, sp)] = counts.get((s, sp), 0) + 1
    assert set(counts) == set(ref_counts), (sorted(counts), sorted(ref_counts))
    for cell, c_ref in ref_counts.items():
        p_ref, p_got = c_ref / n_ref, counts[cell] / n
        pooled = (c_ref + counts[cell]) / (n_ref + n)
        sigma = math.sqrt(pooled * (1 - pooled) * (1 / n_ref + 1 / n))
        assert abs(p_got - p_ref) <= 4.5 * sigma, (cell, p_got, p_ref, sigma)
    # the closed-form probabilities of misc.py:28-34: stage ~ choice([0..6, 0, 1]), spatial with p = 0.3 for 0 < stage < 6
    spatial = sum(c for (s, sp), c in counts.items() if sp) / n
    assert abs(spatial - 0.3 * 6 / 9) < 0.01


def test_storage_noise_model_is_neither_vacuous_nor_exact():
    """oracle.set_storage (the 16-bit storage-noise model the bf16 / fp16 GPU tests take their bounds from, tests/test_gpu_step.py):
    off, the oracle reproduces the goldens as before; on, it loses what 8 (bf16) / 11 (fp16 with the loss scale) significant bits
    lose - pinned to a window, so that a model that stopped rounding (a bound of zero) or rounds wildly (a vacuous bound) fails here."""
    tag = "step_cf4_b4_seed1"
    exact = gu.storage_noise_model(tag, None)
    assert max(exact["pixel_max"]) <= 1e-4 and max(exact["loss_rel"]) <= 1e-5, exact
    bf = gu.storage_noise_model(tag, torch.bfloat16)
    assert 2e-3 <= max(bf["pixel_rms"]) <= 2e-2 and 1e-4 <= max(bf["loss_rel"]) <= 5e-3 and max(bf["pixel_max"]) <= 0.12, bf
    h = gu.storage_noise_model(tag, torch.float16, 65536.0)
    assert 2e-4 <= max(h["pixel_rms"]) <= 4e-3 and max(h["pixel_max"]) <= 3e-2, h
    assert max(h["pixel_rms"]) < 0.4 * max(bf["pixel_rms"])
