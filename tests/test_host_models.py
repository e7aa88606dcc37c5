"""Host-side mirror of the reference module API (CPU only; no kernels run): constructor signatures, state_dict keys
and order, parameter order, seeded-initialisation parity with the reference, and the loud failure off-GPU."""
import inspect

import pytest
import torch

import golden_util as gu
import semantic_pyramid_for_image_generation_amd as sp
from semantic_pyramid_for_image_generation_amd import _lib, ops


@pytest.mark.parametrize("tag", ["step_cf1_b2_seed0", "step_cf4_b4_seed1"])
def test_state_dict_layout_and_seeded_init_match_reference(tag):
    meta, _ = gu.load(tag)
    torch.manual_seed(meta["seed"])
    G = sp.Generator(channels_factor=meta["cf"])
    D = sp.Discriminator(channel_factor=meta["cf"])
    V = sp.VGG16()
    assert list(G.state_dict().keys()) == meta["keys_G"]
    assert list(D.state_dict().keys()) == meta["keys_D"]
    assert list(V.state_dict().keys()) == meta["keys_V"]
    assert [n for n, _ in G.named_parameters()] == meta["param_names_G"]      # optimizer-state compatibility
    assert [n for n, _ in D.named_parameters()] == meta["param_names_D"]
    # torch.manual_seed(s) + construction consumes the RNG like the reference: identical initial parameters
    gu.check_checksums(G.state_dict(), meta["init_checksums_G"], what="G init")
    gu.check_checksums(D.state_dict(), meta["init_checksums_D"], what="D init")


def test_constructor_signatures_match_reference():
    g = inspect.signature(sp.Generator.__init__).parameters
    assert list(g)[1:] == ["out_channels", "latent_dimensions", "channels_factor", "number_of_classes"]
    assert (g["out_channels"].default, g["latent_dimensions"].default, g["channels_factor"].default, g["number_of_classes"].default) == (3, 128, 1, 365)
    d = inspect.signature(sp.Discriminator.__init__).parameters
    assert list(d)[1:] == ["in_channels", "channel_factor", "number_of_classes"]
    v = inspect.signature(sp.VGG16.__init__).parameters
    assert list(v)[1:] == ["path_to_pre_trained_model", "return_output"]
    t = inspect.signature(sp.ModelWrapper.train).parameters
    assert list(t)[1:] == ["epochs", "validate_after_n_iterations", "device", "save_model_after_n_epochs", "w_rec", "w_div"]
    assert (t["epochs"].default, t["validate_after_n_iterations"].default, t["device"].default, t["w_rec"].default, t["w_div"].default) == (20, 100000, "cuda", 0.1, 0.1)
    mw = inspect.signature(sp.ModelWrapper.__init__).parameters
    for name in ("generator", "discriminator", "training_dataset", "validation_dataset", "vgg16", "generator_optimizer",
                 "discriminator_optimizer", "generator_loss", "discriminator_loss", "semantic_reconstruction_loss",
                 "diversity_loss", "save_data_path"):
        assert name in mw
    assert sp.Generator().latent_dimensions == 128


def test_loss_reprs_match_reference():
    assert repr(sp.LSGANGeneratorLoss()) == "LSGANGeneratorLoss"
    assert repr(sp.LSGANDiscriminatorLoss()) == "LSGANDiscriminatorLoss"
    assert repr(sp.DiversityLoss()) == "DiversityLoss"
    assert repr(sp.SemanticReconstructionLoss()) == "SemanticReconstructionLoss, maxpool kernel size2"


def test_product_path_refuses_cpu_tensors():
    """There is no CPU fallback: running the product modules off the GPU must fail loudly."""
    G = sp.Generator(channels_factor=8)
    with pytest.raises(_lib.SempyrError):
        G(input=torch.randn(2, 128), features=[torch.zeros(2, 1)] * 7, masks=[torch.zeros(2, 1)] * 7,
          class_id=torch.zeros(2, 365))
    with pytest.raises(_lib.SempyrError):
        ops.upsample2(torch.zeros(1, 4, 2, 2))


def test_mask_generators_follow_the_contract():
    from semantic_pyramid_for_image_generation_amd import misc, synthetic
    shapes = [(1, 128, 128), (1, 64, 64), (1, 32, 32), (1, 16, 16), (1, 8, 8), (4096,), (365,)]
    for stage in range(7):
        m = misc.get_masks_for_inference(stage)
        assert [tuple(t.shape) for t in m] == shapes
        assert [float(t.max()) for t in m] == [1.0 if i == 6 - stage else 0.0 for i in range(7)]   # stage counts from the deep end
    for _ in range(20):
        m = misc.get_masks_for_training()
        assert [tuple(t.shape) for t in m] == shapes
        for t in m:
            assert set(t.unique().tolist()) <= {0.0, 1.0}
    images, labels, masks = synthetic.synthetic_batch(3, 0)
    assert images.shape == (3, 3, 256, 256) and float(images.min()) >= -1 and float(images.max()) <= 1
    assert labels.dtype == torch.long and labels.shape == (3, 365) and bool((labels.sum(1) == 1).all())
    assert [tuple(t.shape[1:]) for t in masks] == shapes


def test_device_mask_generator_follows_the_contract():
    """Row f1: the device mask generator's contract, checked on the oracle's bit-exact restatement of the kernel
    (oracle.training_masks; the GPU test holds sp_training_masks to it) - the product function itself has no CPU path."""
    import pytest
    import torch.nn.functional as F
    from oracle import sempyr_oracle as O
    from semantic_pyramid_for_image_generation_amd import _lib, synthetic
    shapes = [(1, 128, 128), (1, 64, 64), (1, 32, 32), (1, 16, 16), (1, 8, 8), (4096,), (365,)]
    with pytest.raises(_lib.SempyrError):
        synthetic.training_masks_device(4, "cpu")
    masks = O.training_masks(512, 0)
    assert [tuple(t.shape[1:]) for t in masks] == shapes and all(t.dtype == torch.float32 for t in masks)
    stage_hist = torch.zeros(7)
    n_spatial = 0
    for b in range(512):
        m = [t[b] for t in masks]
        for t in m:
            assert set(t.unique().tolist()) <= {0.0, 1.0}
        ones = [i for i, t in enumerate(m) if bool((t == 1).all())]
        full = [i for i in ones if all(float(m[j].max()) == 0.0 for j in range(i + 1, 7))]
        assert len(full) >= 1
        i = max(full)                                   # list index of the stage (deeper levels are zero)
        stage_hist[i] += 1
        finer = [j for j in range(i) if float(m[j].max()) > 0]
        if finer:
            n_spatial += 1
            assert finer == list(range(i)) and i - 1 <= 4          # every finer level carries the map
            src = m[i - 1]
            for j in range(i - 1):                                  # ... as the nearest-neighbour upsampling of the coarsest one
                assert torch.equal(m[j], F.interpolate(src[None], size=m[j].shape[1:], mode="nearest")[0])
            assert float(src.min()) == 0.0                          # at least one shape
    freq = stage_hist / 512
    assert abs(float(freq[6]) - 2 / 9) < 0.07 and abs(float(freq[5]) - 2 / 9) < 0.07 and abs(float(freq[0]) - 1 / 9) < 0.06
    assert 0.08 < n_spatial / 512 < 0.32


def test_device_mask_shapes_are_the_four_kinds():
    """Row f1 leftovers (round-3 VERDICT): the shape map holds rectangles, circles, triangles and ellipses - the four kinds
    skimage.draw.random_shapes draws (misc.py:37) - each kind about a quarter of the shapes; a circle / triangle / ellipse covers
    less of its bounding box than the rectangle (pi/4, ~1/2, pi/4), every shape at least one pixel.  Checked on the oracle's
    bit-exact restatement of the kernel."""
    from oracle import sempyr_oracle as O
    kinds = [0, 0, 0, 0]
    fill = [[], [], [], []]
    seed, n_samples = 11, 2048
    masks = O.training_masks(n_samples, seed, 1.0)
    for b, (stage, spatial) in enumerate(O.training_mask_decisions(n_samples, seed, 1.0)):
        if not spatial:
            continue
        base = [1, 1, 8, 16, 32, 64, 128][min(stage + 1, 6)]
        lo = min(8, base // 2)
        if 1 + O._draw(seed, b, 2) % 4 != 1:
            continue                                        # single-shape samples: the covered area is that shape's area
        h = lo + O._draw(seed, b, 3) % (base - lo + 1)
        w = lo + O._draw(seed, b, 4) % (base - lo + 1)
        kind = O._draw(seed, b, 19) % 4
        kinds[kind] += 1
        covered = float((masks[6 - (stage + 1)][b] == 0).sum())          # the level just finer than the stage holds the shape map itself
        assert covered >= 1
        fill[kind].append(covered / (h * w))
    n = sum(kinds)
    assert n > 250 and all(abs(k / n - 0.25) < 0.08 for k in kinds), kinds
    mean = [sum(f) / len(f) for f in fill]
    # rectangle: its box; inscribed ellipse: pi / 4; triangle: 1 / 2; circle of diameter min(h, w): pi / 4 of the SMALLER square
    assert mean[0] == 1.0 and 0.4 < mean[1] < 0.8 and 0.4 < mean[2] < 0.6 and 0.72 < mean[3] < 0.88, mean


def test_train_loop_fetches_one_batch_ahead():
    """ModelWrapper._device_batches (the loop of model_wrapper.py:131-135 with the next iteration's real images announced to the
    current one - config.CFG.vgg_pair): every batch exactly once and in order, the announced tensor IS the next iteration's images
    tensor, a last batch of another shape is not announced, and with the switch off nothing is fetched ahead."""
    import types
    from semantic_pyramid_for_image_generation_amd import model_wrapper as mwm

    def batch(i, n):
        return torch.full((n, 3, 8, 8), float(i)), torch.zeros(n, 5), [torch.ones(n, 1, 4, 4)]
    loader = [batch(0, 4), batch(1, 4), batch(2, 4), batch(3, 2)]
    stub = types.SimpleNamespace(training_dataset=loader)
    old = mwm.CFG.vgg_pair
    try:
        mwm.CFG.vgg_pair = True
        got = list(mwm.ModelWrapper._device_batches(stub, "cpu"))
        assert [float(g[0].flatten()[0]) for g in got] == [0.0, 1.0, 2.0, 3.0]
        assert got[0][3] is got[1][0] and got[1][3] is got[2][0]
        assert got[2][3] is None and got[3][3] is None          # other shape / no successor
        mwm.CFG.vgg_pair = False
        got = list(mwm.ModelWrapper._device_batches(stub, "cpu"))
        assert [float(g[0].flatten()[0]) for g in got] == [0.0, 1.0, 2.0, 3.0] and all(g[3] is None for g in got)
    finally:
        mwm.CFG.vgg_pair = old


def test_device_mask_seeds_leave_the_global_generator_alone():
    """Round-4 ADVICE: without a caller's generator the device-mask seeds come from a module-private CPU generator (seeded once from
    torch.initial_seed() and the rank), not from torch's global one - whose stream DataLoader seeds and user code also consume."""
    from semantic_pyramid_for_image_generation_amd import synthetic
    torch.manual_seed(1234)
    synthetic._PRIVATE_GEN[0] = None
    before = torch.get_rng_state()
    g = synthetic._private_generator()
    draws = [int(torch.randint(0, (1 << 63) - 1, (1,), dtype=torch.int64, generator=g)) for _ in range(3)]
    assert torch.equal(before, torch.get_rng_state())
    assert synthetic._private_generator() is g and len(set(draws)) == 3
    synthetic._PRIVATE_GEN[0] = None
    g2 = synthetic._private_generator()                         # same initial seed -> same mask sequence
    assert int(torch.randint(0, (1 << 63) - 1, (1,), dtype=torch.int64, generator=g2)) == draws[0]
    synthetic._PRIVATE_GEN[0] = None


def test_image_grid_png_matches_torchvisions_layout(tmp_path):
    """misc.save_image_grid = torchvision.utils.save_image(x, path, nrow) of /root/reference/model_wrapper.py:290-292 without
    torchvision: make_grid's geometry (padding 2, zeros) and save_image's rounding, as a PNG any decoder reads (PIL here)."""
    import numpy as np
    from semantic_pyramid_for_image_generation_amd import misc
    g = torch.Generator().manual_seed(5)
    x = torch.rand(10, 3, 12, 9, generator=g)
    path = str(tmp_path / "grid.png")
    misc.save_image_grid(x, path, nrow=7)
    got = misc.load_png_rgb8(path)
    assert got.shape == (2 * 14 + 2, 7 * 11 + 2, 3)
    want = np.zeros(got.shape, dtype=np.uint8)
    for k in range(10):
        r, c = divmod(k, 7)
        tile = (x[k] * 255 + 0.5).clamp(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
        want[r * 14 + 2:r * 14 + 14, c * 11 + 2:c * 11 + 11] = tile
    assert (got == want).all()
    try:
        from PIL import Image
    except ImportError:
        return
    assert (np.asarray(Image.open(path).convert("RGB")) == want).all()
    n01 = misc.normalize_0_1_batch(torch.randn(3, 3, 4, 4, generator=g))
    assert float(n01.min()) == 0.0 and float(n01.max()) == 1.0
