"""Per-block parity against tensors recorded from the UNMODIFIED reference blocks (tests/golden/make_block_golden.py;
SURVEY.md section 8c: rows a4-a9, a12-a14 at tiny shapes, full tensors).

CPU part (no marker): pins the oracle's block functions to the reference.  GPU part (``-m gpu``): the HIP modules, through the
C ABI, in the fp32 parity mode - outputs, input gradients, every parameter gradient and the buffers the forward mutates.
"""
import json
import os

import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import sempyr_oracle as O

ARR = dict(np.load(os.path.join(gu.GOLDEN, "blocks.npz")))
META = json.load(open(os.path.join(gu.GOLDEN, "blocks.json")))
BLOCKS = ["linear_block", "generator_residual_block", "conditional_batch_norm", "self_attention",
          "discriminator_input_residual_block", "discriminator_residual_block"]


def T(key):
    return torch.from_numpy(ARR[key])


def block_inputs(name):
    ins = {}
    for k in META[name]["inputs"]:
        t = T("%s/in/%s" % (name, k)).clone()
        ins[k] = t.requires_grad_(True) if k in META[name]["diff_inputs"] else t
    return ins


def block_state(name, prefix=""):
    return {prefix + k: T("%s/param/%s" % (name, k)).clone() for k in META[name]["state_keys"]}


def assert_close(got, ref, tol, what, floor=1e-6):
    """max |err| <= tol * max(max|ref|, floor).  ``floor``: magnitude below which a tensor is rounding noise - the gradient of a
    bias that feeds a BatchNorm is exactly zero in exact arithmetic and ~1e-6 of the other gradients in fp32."""
    got, ref = got.detach().float().cpu(), ref.float()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(float(ref.abs().max()), floor)
    err = float((got - ref).abs().max())
    assert err <= tol * scale, "%s: max err %.3e vs scale %.3e (tol %.1e)" % (what, err, scale, tol)


ORACLE_CALL = {
    "linear_block": lambda S, i: O.linear_block(S, "b", i["input"], i["masked_features"], True),
    "generator_residual_block": lambda S, i: O.generator_residual_block(S, "b", i["input"], i["masked_features"], i["class_id"], True),
    "conditional_batch_norm": lambda S, i: O.conditional_batch_norm(S, "b", i["input"], i["class_id"], True),
    "self_attention": lambda S, i: O.self_attention(S, "b", i["input"], True),
    "discriminator_input_residual_block": lambda S, i: O.discriminator_input_block(S, "b", i["input"], True),
    "discriminator_residual_block": lambda S, i: O.discriminator_residual_block(S, "b", i["input"], True),
}


@pytest.mark.parametrize("name", BLOCKS)
def test_oracle_block_matches_reference(name):
    torch.set_num_threads(1)
    S = O.make_state(block_state(name, "b."))
    ins = block_inputs(name)
    out = ORACLE_CALL[name](S, ins)
    out.backward(T(name + "/gout"))
    assert_close(out, T(name + "/out"), 1e-5, "out")
    for k in META[name]["diff_inputs"]:
        assert_close(ins[k].grad, T("%s/gin/%s" % (name, k)), 2e-5, "d" + k)
    gscale = max(float(T("%s/grad/%s" % (name, k)).abs().max()) for k in META[name]["param_names"])
    for k in META[name]["param_names"]:
        g = S["b." + k].grad
        assert_close(g if g is not None else torch.zeros_like(S["b." + k]), T("%s/grad/%s" % (name, k)), 5e-5, "grad " + k,
                     floor=1e-2 * gscale)
    for key in ARR:
        if key.startswith(name + "/buf/"):
            k = key[len(name) + 5:]
            assert_close(S["b." + k].float(), T(key).float(), 1e-5, "buffer " + k)


def test_oracle_losses_match_reference():
    pr, pf, pg = (T("lsgan/in/" + k).clone().requires_grad_(True) for k in ("pred_real", "pred_fake", "pred_gen"))
    lr, lf = O.lsgan_discriminator_loss(pr, pf)
    (lr + lf).backward()
    lg = O.lsgan_generator_loss(pg)
    lg.backward()
    assert np.allclose([float(lr), float(lf), float(lg)], ARR["lsgan/out"], rtol=1e-6)
    for k, v in (("pred_real", pr), ("pred_fake", pf), ("pred_gen", pg)):
        assert_close(v.grad, T("lsgan/gin/" + k), 1e-6, k)
    n = META["losses"]["rec_levels"]
    real = [T("rec/in/real%d" % i) for i in range(n)]
    fake = [T("rec/in/fake%d" % i).clone().requires_grad_(True) for i in range(n)]
    masks = [T("rec/in/mask%d" % i) for i in range(n)]
    loss = O.semantic_reconstruction_loss(real, fake, masks)
    assert tuple(loss.shape) == (1,)
    loss.backward(torch.ones_like(loss))
    assert np.allclose(loss.detach().numpy(), ARR["rec/out"], rtol=1e-6)
    for i in range(n):
        assert_close(fake[i].grad, T("rec/gin/fake%d" % i), 1e-6, "rec level %d" % i)
    img = T("div/in/images").clone().requires_grad_(True)
    ld = O.diversity_loss(img, T("div/in/latents"))
    ld.backward()
    assert np.allclose(float(ld), ARR["div/out"], rtol=1e-6)
    assert_close(img.grad, T("div/gin/images"), 1e-6, "div")


# ------------------------------------------------------------------------------------------------------------------
# GPU: the HIP modules against the same reference tensors (fp32 parity mode)
# ------------------------------------------------------------------------------------------------------------------
def _gpu_modules():
    from semantic_pyramid_for_image_generation_amd import models, ops
    c = META["classes"]
    ctor = {
        "linear_block": lambda: models.LinearBlock(in_features=16, out_features=24, feature_size=40),
        "generator_residual_block": lambda: models.GeneratorResidualBlock(16, 8, 9, number_of_classes=c),
        "conditional_batch_norm": lambda: models.ConditionalBatchNorm(8, number_of_classes=c),
        "self_attention": lambda: models.SelfAttention(channels=32),
        "discriminator_input_residual_block": lambda: models.DiscriminatorInputResidualBlock(3, 8),
        "discriminator_residual_block": lambda: models.DiscriminatorResidualBlock(8, 16),
    }
    f32 = torch.float32

    def nhwc(t):
        return ops.as_nhwc(t.detach().cuda(), f32)

    def call(name, m, ins):
        """-> (output, {input name: tensor whose .grad is compared})"""
        if name == "linear_block":
            x = ins["input"].detach().cuda().requires_grad_(True)
            return m(x, ins["masked_features"].cuda()), {"input": x}
        if name == "generator_residual_block":
            x = nhwc(ins["input"]).requires_grad_(True)
            cat = ins["masked_features"].cuda()
            feat = ops.mask_concat(cat[:, :-1].contiguous(), cat[:, -1:].contiguous())     # (feat*mask)*mask == feat*mask: masks are 0/1
            return m(x, feat, ins["class_id"].cuda()), {"input": x}
        if name == "conditional_batch_norm":
            x = nhwc(ins["input"]).requires_grad_(True)
            return m(x, ins["class_id"].cuda()), {"input": x}
        if name == "discriminator_input_residual_block":
            x = ins["input"].detach().cuda().requires_grad_(True)
            return m(ops.ingest_image(x, f32)), {"input": x}
        x = nhwc(ins["input"]).requires_grad_(True)
        return m(x), {"input": x}
    return ctor, call


@pytest.mark.gpu
@pytest.mark.parametrize("name", BLOCKS)
def test_hip_block_matches_reference(name):
    from semantic_pyramid_for_image_generation_amd import ops
    ops.set_compute_dtype(torch.float32)
    ctor, call = _gpu_modules()
    m = ctor[name]()
    m.load_state_dict(block_state(name))
    m = m.cuda().train()
    ins = block_inputs(name)
    out, leaves = call(name, m, ins)
    gy = T(name + "/gout").cuda()
    out.backward(ops.as_nhwc(gy, torch.float32) if gy.dim() == 4 else gy)
    tol = 2e-4
    assert_close(out, T(name + "/out"), tol, "out")
    for k, leaf in leaves.items():
        assert_close(leaf.grad, T("%s/gin/%s" % (name, k)), tol, "d" + k)
    params = dict(m.named_parameters())
    gscale = max(float(T("%s/grad/%s" % (name, k)).abs().max()) for k in META[name]["param_names"])
    for k in META[name]["param_names"]:
        g = params[k].grad
        assert_close(g if g is not None else torch.zeros_like(params[k]), T("%s/grad/%s" % (name, k)), 2 * tol, "grad " + k,
                     floor=1e-2 * gscale)
    sd = m.state_dict()
    for key in ARR:
        if key.startswith(name + "/buf/"):
            k = key[len(name) + 5:]
            assert_close(sd[k].float(), T(key).float(), 1e-5, "buffer " + k)


@pytest.mark.gpu
def test_hip_losses_match_reference():
    from semantic_pyramid_for_image_generation_amd import lossfunction, ops
    ops.set_compute_dtype(torch.float32)
    pr, pf, pg = (T("lsgan/in/" + k).cuda().requires_grad_(True) for k in ("pred_real", "pred_fake", "pred_gen"))
    lr, lf = lossfunction.LSGANDiscriminatorLoss()(pr, pf)
    (lr + lf).backward()
    lg = lossfunction.LSGANGeneratorLoss()(pg)
    lg.backward()
    assert np.allclose([float(lr), float(lf), float(lg)], ARR["lsgan/out"], rtol=1e-5)
    for k, v in (("pred_real", pr), ("pred_fake", pf), ("pred_gen", pg)):
        assert_close(v.grad, T("lsgan/gin/" + k), 1e-5, k)
    n = META["losses"]["rec_levels"]
    to_dev = lambda t: ops.as_nhwc(t.cuda(), torch.float32) if t.dim() == 4 else t.cuda()     # noqa: E731
    real = [to_dev(T("rec/in/real%d" % i)) for i in range(n)]
    fake = [to_dev(T("rec/in/fake%d" % i)).requires_grad_(True) for i in range(n)]
    masks = [T("rec/in/mask%d" % i).cuda() for i in range(n)]
    loss = lossfunction.SemanticReconstructionLoss()(real, fake, masks)
    assert tuple(loss.shape) == (1,)                      # lossfunction.py:42
    loss.backward(torch.ones_like(loss))
    assert np.allclose(loss.detach().cpu().numpy(), ARR["rec/out"], rtol=1e-5)
    for i in range(n):
        assert_close(fake[i].grad, T("rec/gin/fake%d" % i), 1e-5, "rec level %d" % i)
    img = ops.as_nhwc(T("div/in/images").cuda(), torch.float32).requires_grad_(True)
    ld = lossfunction.DiversityLoss()(img, T("div/in/latents").cuda())
    ld.backward()
    assert np.allclose(float(ld), ARR["div/out"], rtol=1e-5)
    assert_close(img.grad, T("div/gin/images"), 1e-5, "div")
