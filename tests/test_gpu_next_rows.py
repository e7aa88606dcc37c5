"""SURVEY.md section 8(f) rows on the device: f1 on-device mask generation feeding a training step, f3 checkpoint round trip
through a running GPU job (reference layout, model_wrapper.py:215-223 / main.py:61,68-73) incl. the
``VGG16(path_to_pre_trained_model)`` constructor branch (models.py:163-181)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from semantic_pyramid_for_image_generation_amd.config import CFG  # noqa: E402

import golden_util as gu  # noqa: E402
import semantic_pyramid_for_image_generation_amd as sp  # noqa: E402
from semantic_pyramid_for_image_generation_amd import models, ops, params, synthetic  # noqa: E402

LOSS_NAMES = ("loss_discriminator_real", "loss_discriminator_fake", "loss_generator",
              "loss_generator_semantic_reconstruction", "loss_generator_diversity")


@pytest.fixture(autouse=True)
def _dtype_reset():
    yield
    ops.set_compute_dtype(torch.float32)


def build(cf, seed):
    meta = {"cf": cf, "seed": seed}
    Gsd, Dsd, Vsd = gu.synth_states(meta)
    G = sp.Generator(channels_factor=cf); D = sp.Discriminator(channel_factor=cf); V = sp.VGG16()
    G.load_state_dict(Gsd); D.load_state_dict(Dsd); V.load_state_dict(Vsd)
    return G.cuda(), D.cuda(), V.cuda().eval()


def wrapper(G, D, V, lr=1e-4):
    og, od = sp.optim.Adam(G.parameters(), lr=lr), sp.optim.Adam(D.parameters(), lr=lr)
    mw = sp.ModelWrapper(generator=G, discriminator=D, vgg16=V, training_dataset=None, validation_dataset=None,
                         generator_optimizer=og, discriminator_optimizer=od, save_data_path=None)
    G.train(); D.train()
    return mw, og, od


def test_device_mask_generation_contract_and_step():
    """Row f1: synthetic.training_masks_device() on the GPU keeps the a15 contract (exact 0/1 values, one open stage per
    sample counted from the deep end, finer levels = nearest-neighbour copies of one rectangle map, deeper levels zero) and
    its output drives a training step directly - no host round trip."""
    g = torch.Generator().manual_seed(3)
    masks = synthetic.training_masks_device(256, "cuda", g)
    # the sequence belongs to the generator: re-seeding restarts it, and two consecutive batches differ (round-3 ADVICE)
    again = synthetic.training_masks_device(256, "cuda", torch.Generator().manual_seed(3))
    assert all(torch.equal(a, b) for a, b in zip(masks, again))
    nxt = synthetic.training_masks_device(256, "cuda", g)
    assert any(not torch.equal(a, b) for a, b in zip(masks, nxt))
    shapes = [(1, 128, 128), (1, 64, 64), (1, 32, 32), (1, 16, 16), (1, 8, 8), (4096,), (365,)]
    assert all(m.is_cuda and m.dtype == torch.float32 for m in masks)
    assert [tuple(t.shape[1:]) for t in masks] == shapes
    host = [m.cpu() for m in masks]
    n_spatial = 0
    for b in range(256):
        m = [t[b] for t in host]
        for t in m:
            assert set(t.unique().tolist()) <= {0.0, 1.0}
        full = [i for i, t in enumerate(m) if bool((t == 1).all()) and all(float(m[j].max()) == 0.0 for j in range(i + 1, 7))]
        assert len(full) >= 1
        i = max(full)
        finer = [j for j in range(i) if float(m[j].max()) > 0]
        if finer:
            n_spatial += 1
            assert finer == list(range(i)) and i - 1 <= 4
            for j in range(i - 1):
                assert torch.equal(m[j], F.interpolate(m[i - 1][None], size=m[j].shape[1:], mode="nearest")[0])
    assert 0.05 < n_spatial / 256 < 0.35
    ops.set_compute_dtype(torch.bfloat16)
    G, D, V = build(4, 5)
    mw, _, _ = wrapper(G, D, V)
    images, labels, _ = synthetic.synthetic_batch(8, 2)
    out = mw.train_step(images.cuda(), labels.cuda(), synthetic.training_masks_device(8, "cuda", g))
    for n in LOSS_NAMES:
        assert np.isfinite(float(out[n])), n


@pytest.mark.parametrize("seed,p", [(0, 0.3), (12345678901234567, 0.3), (7, 1.0), (8, 0.0)])
def test_device_mask_generator_is_bit_exact_with_the_oracle(seed, p):
    """Row f1: sp_training_masks (one launch per batch, csrc/eltwise.hip) against oracle.training_masks - the generator is
    integer arithmetic on a counter-based hash, so all seven tensors must be EQUAL, for every sample of a 384-sample batch
    (every stage, spatial and non-spatial cases, 1-4 shapes of the four kinds; p = 1 / p = 0 force both branches)."""
    from oracle import sempyr_oracle as O
    got = synthetic.training_masks_device(384, "cuda", seed=seed, p_random_mask=p)
    want = O.training_masks(384, seed, p)
    torch.cuda.synchronize()
    assert [tuple(t.shape) for t in got] == [tuple(t.shape) for t in want]
    for i, (g, w) in enumerate(zip(got, want)):
        assert torch.equal(g.cpu(), w), (seed, p, i, int((g.cpu() != w).sum()))
    dec = O.training_mask_decisions(384, seed, p)
    assert {s for s, _ in dec} == set(range(7))
    assert any(sp for _, sp in dec) == (p > 0)


@pytest.mark.parametrize("per_channel", [True, False])
def test_minmax_ingest_is_bit_identical_to_the_torch_expression(per_channel):
    """data.py:53 on the device (round-3 VERDICT, f1 leftovers): kornia.normalize_min_max(x, -1, 1) =
    (max_val - min_val) * (x - x_min) / (x_max - x_min + eps) + min_val with the extrema per (image, channel) plane (kornia) or per
    image; fp32, same operation order - equal bit for bit, incl. a constant plane (0 / eps) and 8-bit-quantised inputs."""
    from semantic_pyramid_for_image_generation_amd import data
    g = torch.Generator().manual_seed(4)
    x = torch.rand(6, 3, 256, 256, generator=g)
    x[1] = (x[1] * 255).round() / 255            # what TVF.to_tensor yields for a JPEG
    x[2, 1] = 0.25                               # a constant plane
    x[3] *= 0.3
    b, c = x.shape[:2]
    if per_channel:
        lo = x.view(b, c, -1).min(-1)[0].view(b, c, 1, 1)
        hi = x.view(b, c, -1).max(-1)[0].view(b, c, 1, 1)
    else:
        lo = x.view(b, -1).min(-1)[0].view(b, 1, 1, 1)
        hi = x.view(b, -1).max(-1)[0].view(b, 1, 1, 1)
    want = (1.0 - (-1.0)) * (x - lo) / (hi - lo + 1e-6) + (-1.0)
    got = data.normalize_min_max_device(x.cuda(), per_channel=per_channel).cpu()
    assert torch.equal(got, want)
    assert float(got.max()) <= 1.0 and float(got.min()) == -1.0


def test_checkpoint_round_trip_on_the_device(tmp_path):
    """Row f3: iteration 1, checkpoint in the reference's layout, fresh objects restored from the file, iteration 2 - against the
    uninterrupted run.  fp32 mode reduces in a fixed order, so the two runs must agree bit for bit."""
    ops.set_compute_dtype(torch.float32)
    batches = gu.golden_batches(4, 1)
    noise = torch.randn(4, 4, 128, generator=torch.Generator().manual_seed(2)).cuda()

    def step(mw, it):
        images, labels, masks = batches[it]
        return mw.train_step(images.cuda(), labels.cuda(), [m.cuda() for m in masks], noise_d=noise[2 * it], noise_g=noise[2 * it + 1])

    G, D, V = build(4, 1)
    mw, og, od = wrapper(G, D, V)
    step(mw, 0)
    path = os.path.join(tmp_path, "checkpoint_000.pt")
    torch.save({"generator": G.state_dict(), "discriminator": D.state_dict(),               # model_wrapper.py:215-223
                "generator_optimizer": og.state_dict(), "discriminator_optimizer": od.state_dict()}, path)
    ref = step(mw, 1)
    ref_state = {k: v.detach().clone() for k, v in G.state_dict().items()}

    ck = torch.load(path, map_location="cpu")                                                # main.py:68-73
    assert all(any(k.endswith(s) for k in ck["generator"]) for s in ("weight_orig", "weight_u", "weight_v"))
    G2 = sp.Generator(channels_factor=4).cuda(); D2 = sp.Discriminator(channel_factor=4).cuda()
    og2 = sp.optim.Adam(G2.parameters(), lr=1e-4); od2 = sp.optim.Adam(D2.parameters(), lr=1e-4)
    G2.load_state_dict(ck["generator"]); D2.load_state_dict(ck["discriminator"])
    og2.load_state_dict(ck["generator_optimizer"]); od2.load_state_dict(ck["discriminator_optimizer"])
    # the same file restores plain torch optimizers (what main.py:64-65 constructs): same keys, steps and moment shapes
    ot = torch.optim.Adam(sp.Generator(channels_factor=4).cuda().parameters(), lr=1e-4)
    ot.load_state_dict(ck["generator_optimizer"])
    p0 = ot.param_groups[0]["params"][0]
    assert float(ot.state[p0]["step"]) == 1.0 and ot.state[p0]["exp_avg"].shape == p0.shape and ot.state[p0]["exp_avg"].is_cuda
    mw2 = sp.ModelWrapper(generator=G2, discriminator=D2, vgg16=V, training_dataset=None, validation_dataset=None,
                          generator_optimizer=og2, discriminator_optimizer=od2, save_data_path=None)
    G2.train(); D2.train()
    got = step(mw2, 1)
    for n in LOSS_NAMES:
        assert float(got[n]) == float(ref[n]), n
    assert torch.equal(got["images_fake"], ref["images_fake"])
    for k, v in G2.state_dict().items():                      # parameters, spectral-norm vectors, BatchNorm statistics, counters
        assert torch.equal(v, ref_state[k]), k


def test_vgg16_constructor_loads_a_pretrained_file(tmp_path):
    """models.py:163-181: ``VGG16(path_to_pre_trained_model=...)`` unpickles a whole torchvision-style model and adopts its
    parameters (main.py:61 instead loads a state_dict into ``VGG16()``: both routes must give the same pyramid)."""
    topo = models._VGG16Topology(num_classes=365)
    sd = params.synth_state_dict({"vgg16." + k: v for k, v in topo.state_dict().items()}, 9)
    topo.load_state_dict({k[len("vgg16."):]: v for k, v in sd.items()})
    path = os.path.join(tmp_path, "vgg_places_365_fine_tuned.pt")
    torch.save(topo, path)
    a = sp.VGG16(path_to_pre_trained_model=path).cuda().eval()
    b = sp.VGG16()
    b.load_state_dict(sd)
    b = b.cuda().eval()
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())
    ops.set_compute_dtype(torch.float32)
    images, _, _ = synthetic.synthetic_batch(2, 4)
    with torch.no_grad():
        fa, fb = a(images.cuda()), b(images.cuda())
    assert len(fa) == 7
    for x, y in zip(fa, fb):
        assert torch.equal(x, y)


def test_every_gradient_lives_in_the_flat_buffer():
    """Row e (round-2 ADVICE, high): the data-parallel reducer all-reduces ranges of the networks' flat gradient buffers, so a
    gradient that autograd leaves OUTSIDE the buffer would silently stay un-averaged on N > 1 ranks.  The discriminator head's
    classification bias was one (its gradient is returned through autograd, not written by a kernel): after the backward of the
    D phase / the G phase every parameter's .grad must be a view of its bank's buffer."""
    ops.set_compute_dtype(torch.float32)
    G, D, V = build(4, 1)
    og, od = sp.optim.Adam(G.parameters(), lr=1e-4), sp.optim.Adam(D.parameters(), lr=1e-4)
    mw = sp.ModelWrapper(generator=G, discriminator=D, vgg16=V, training_dataset=None, validation_dataset=None,
                         generator_optimizer=og, discriminator_optimizer=od, save_data_path=None)
    G.train(); D.train()
    images, labels, masks = gu.golden_batches(4, 1)[0]
    images, labels, masks = images.cuda(), labels.cuda(), [m.cuda() for m in masks]

    def inside(net, key):
        flat = mw._banks[key].flat
        lo, hi = flat.data_ptr(), flat.data_ptr() + 4 * flat.numel()
        missing = [n for n, p in net.named_parameters() if p.grad is None]
        loose = [n for n, p in net.named_parameters() if p.grad is not None and not (lo <= p.grad.data_ptr() < hi)]
        return missing, loose

    feats, _, _ = mw._d_phase(images, labels, labels.float(), masks, None)
    assert inside(D, "d") == ([], []), inside(D, "d")
    assert float(D.classification.bias.grad.abs().sum()) > 0.0
    fake, z = mw._g_forward(images, labels.float(), masks, feats, None)
    mw._g_rest(fake, z, labels, masks, feats, 0.1, 0.1)
    assert inside(G, "g") == ([], []), inside(G, "g")


class _HalvingReducer:
    """Stands in for an all-reduce whose result DIFFERS from the rank's own gradients (what a second rank with other data causes):
    every range handed over is halved in place on the current stream, exactly once."""
    bucket_bytes = 1 << 20

    def __init__(self):
        self.ranges = []

    def active(self):
        return True

    def reduce_range(self, flat, a, b):
        self.ranges.append((a, b))
        flat[a:b].mul_(0.5)

    def reduce_flat(self, flat, ranges):
        for a, b in ranges:
            self.reduce_range(flat, a, b)

    def reduce(self, params):
        raise AssertionError("a gradient outside the flat buffer reached the per-parameter path")

    def join(self, tag=""):
        pass


def test_eager_group_hooks_reduce_every_gradient_exactly_once():
    """Round-3 ADVICE (medium): on the eager data-parallel path the bank's layer groups hand their flat range to the reducer from
    INSIDE the backward pass.  A gradient that reaches the flat buffer only after .backward() returned (the discriminator head's
    classification bias used to) is then skipped as 'already reduced' - un-averaged on N > 1 ranks - and its late copy races with
    the collective.  With a reducer whose result differs from the local gradients (it halves every range it is given), every
    gradient of D and of G must come out at exactly half of the reducer-less run."""
    ops.set_compute_dtype(torch.float32)
    images, labels, masks = gu.golden_batches(4, 1)[0]
    images, labels, masks = images.cuda(), labels.cuda(), [m.cuda() for m in masks]
    noise = torch.randn(2, 4, 128, generator=torch.Generator().manual_seed(5)).cuda()

    def run(reducer):
        G, D, V = build(4, 1)
        og, od = sp.optim.Adam(G.parameters(), lr=1e-4), sp.optim.Adam(D.parameters(), lr=1e-4)
        mw = sp.ModelWrapper(generator=G, discriminator=D, vgg16=V, training_dataset=None, validation_dataset=None,
                             generator_optimizer=og, discriminator_optimizer=od, save_data_path=None, gradient_reducer=reducer)
        G.train(); D.train()
        feats, _, _ = mw._d_phase(images, labels, labels.float(), masks, noise[0], None, noise[1])     # (both latents: the generator pair pass)
        mw._start_reduce("d", mw._d_params, eager=True)
        dg = {n: p.grad.detach().clone() for n, p in D.named_parameters()}
        fake, z = mw._g_forward(images, labels.float(), masks, feats, noise[1])
        mw._g_rest(fake, z, labels, masks, feats, 0.1, 0.1)
        mw._start_reduce("g", mw._g_params, eager=True)
        gg = {n: p.grad.detach().clone() for n, p in G.named_parameters()}
        torch.cuda.synchronize()
        return dg, gg

    ref_d, ref_g = run(None)
    red = _HalvingReducer()
    got_d, got_g = run(red)
    assert len(red.ranges) > 8
    assert float(ref_d["classification.bias"].abs().sum()) > 0.0
    for ref, got, tag in ((ref_d, got_d, "D"), (ref_g, got_g, "G")):
        for n in ref:
            assert torch.equal(got[n], ref[n] * 0.5), (tag, n)


def test_multi_gpu_code_path_in_a_one_rank_rccl_group():
    """Row e on one GPU: a real RCCL process group of ONE rank with the reducer kept live (single_rank_passthrough=False), so the
    side stream, the events, the in-place bucketed all-reduce of the flat gradient buffers, the group hooks of the eager
    backward and the three-graph replay all run - the collective itself is the identity, so every result must be bit-identical
    to the run without a reducer (fp32 mode: deterministic)."""
    import torch.distributed as dist
    from semantic_pyramid_for_image_generation_amd.distributed import GradientReducer
    ops.set_compute_dtype(torch.float32)
    batches = gu.golden_batches(4, 1)
    noise = torch.randn(4, 4, 128, generator=torch.Generator().manual_seed(2)).cuda()

    def run(reducer, graphed):
        G, D, V = build(4, 1)
        og, od = sp.optim.Adam(G.parameters(), lr=1e-4), sp.optim.Adam(D.parameters(), lr=1e-4)
        mw = sp.ModelWrapper(generator=G, discriminator=D, vgg16=V, training_dataset=None, validation_dataset=None,
                             generator_optimizer=og, discriminator_optimizer=od, save_data_path=None, gradient_reducer=reducer)
        G.train(); D.train()
        outs = []
        images, labels, masks = batches[0]
        images, labels, masks = images.cuda(), labels.cuda(), [m.cuda() for m in masks]
        outs.append(mw.train_step(images, labels, masks, noise_d=noise[0], noise_g=noise[1]))
        if graphed:
            mw.capture_graphs(images, labels, masks)
            out = mw.train_step_graphed(noise_d=noise[2], noise_g=noise[3])
        else:
            out = mw.train_step(images, labels, masks, noise_d=noise[2], noise_g=noise[3])
        outs.append({k: v.clone() for k, v in out.items()})
        torch.cuda.synchronize()
        return outs, {k: v.detach().clone() for k, v in G.state_dict().items()}, {k: v.detach().clone() for k, v in D.state_dict().items()}

    ref = run(None, False)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for graphed in (False, True):
            red = GradientReducer(bucket_bytes=1 << 20, single_rank_passthrough=False)
            log = []
            orig = red.reduce_range

            def spy(flat, a, b, orig=orig, log=log):
                log.append((flat.data_ptr(), a, b))
                return orig(flat, a, b)
            red.reduce_range = spy
            loose_calls = []
            red.reduce = lambda params, loose_calls=loose_calls: loose_calls.append(len(list(params)))
            got = run(red, graphed)
            assert loose_calls == [], "gradients outside the flat buffers: %r" % loose_calls
            assert len(log) > 8, "no reduction was issued"
            for it in range(2):
                for n in LOSS_NAMES:
                    assert float(got[0][it][n]) == float(ref[0][it][n]), (graphed, it, n)
                assert torch.equal(got[0][it]["images_fake"], ref[0][it]["images_fake"]), (graphed, it)
            for k in ref[1]:
                assert torch.equal(got[1][k], ref[1][k]), (graphed, "G", k)
            for k in ref[2]:
                assert torch.equal(got[2][k], ref[2][k]), (graphed, "D", k)
    finally:
        dist.destroy_process_group()
