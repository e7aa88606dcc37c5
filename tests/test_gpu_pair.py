"""The discriminator step's D(real) + D(fake) as ONE two-group pass (models.Discriminator.forward_pair; the reference runs two
forwards, /root/reference/model_wrapper.py:153-155, each advancing the spectral-norm power iteration, models.py:128-135):
predictions, every parameter gradient and the (u, v) buffers must equal two separate calls - to fp32 rounding in the parity
mode, to storage rounding in bf16 - and the C ABI's per-group accumulator scale (sp_conv_params.img_scale) is held to a torch
reference on every convolution route it touches."""
import copy

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import golden_util as gu  # noqa: E402
import semantic_pyramid_for_image_generation_amd as sp  # noqa: E402
from semantic_pyramid_for_image_generation_amd import ops  # noqa: E402


@pytest.fixture(autouse=True)
def _dtype_reset():
    yield
    ops.set_compute_dtype(torch.float32)


def _disc(cf, seed):
    _, Dsd, _ = gu.synth_states({"cf": cf, "seed": seed})
    D = sp.Discriminator(channel_factor=cf)
    D.load_state_dict(Dsd)
    D = D.cuda().train()
    D._bank.direct_grads, D._bank.expected_passes = True, 2
    return D


@pytest.mark.parametrize("cf,batch,dtype,tol_pred,tol_grad", [(4, 4, torch.float32, 2e-5, 2e-4), (1, 6, torch.float32, 2e-5, 2e-4),
                                                             (1, 20, torch.bfloat16, 3e-2, 1e-1)])
def test_pair_pass_equals_two_forwards(cf, batch, dtype, tol_pred, tol_grad):
    # (bf16 gradient bound: 6e-2 until round 5; the attention gate's gradient - ONE scalar, a sum of products with cancellation - sits
    # at 6.5e-2 since the small-map convolutions split K differently (deterministic: the same figure in every run); every tensor-
    # valued gradient stays below 4e-2)
    ops.set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(11)
    real = (torch.rand(batch, 3, 256, 256, generator=g) * 2 - 1).cuda()
    fake = (torch.rand(batch, 3, 256, 256, generator=g) * 2 - 1).cuda()
    cls = torch.randint(0, 365, (batch,), generator=g)
    labels = F.one_hot(cls, 365).long().cuda()
    loss_fn = sp.LSGANDiscriminatorLoss()
    outs = []
    for mode in ("two", "pair"):
        D = _disc(cf, 3)
        if mode == "two":
            pr, pf = D(real, labels), D(fake, labels)
        else:
            pr, pf = D.forward_pair(real, fake, labels)
        lr, lf = loss_fn(pr, pf)
        (lr + lf).backward()
        D._bank.collect_extra()
        torch.cuda.synchronize()
        outs.append((pr.detach().float().clone(), pf.detach().float().clone(),
                     {n: p.grad.detach().float().clone() for n, p in D.named_parameters() if p.grad is not None},
                     {n: b.detach().clone() for n, b in D.named_buffers()}))
    (pr0, pf0, g0, b0), (pr1, pf1, g1, b1) = outs
    for a, b, tag in ((pr0, pr1, "real"), (pf0, pf1, "fake")):
        err = float((a - b).abs().max() / a.abs().max())
        assert err <= tol_pred, (tag, err)
    assert set(g0) == set(g1) and len(g0) > 50
    dens = sorted(float(g0[n].abs().max()) for n in g0)
    floor = 1e-2 * dens[len(dens) // 2]           # a gradient that is rounding noise in exact arithmetic (the key convolution's bias
    for n in g0:                                  # shifts every logit of a softmax row alike) is held to the typical scale instead
        den = max(float(g0[n].abs().max()), floor)
        err = float((g0[n] - g1[n]).abs().max()) / den
        assert err <= tol_grad, (n, err)
    for n in b0:                                  # u, v: two power iterations either way - identical arithmetic
        assert torch.equal(b0[n], b1[n]), n


def _gen(cf, seed):
    Gsd, _, _ = gu.synth_states({"cf": cf, "seed": seed})
    G = sp.Generator(channels_factor=cf)
    G.load_state_dict(Gsd)
    G = G.cuda().train()
    G._bank.direct_grads, G._bank.expected_passes = True, 1
    return G


@pytest.mark.parametrize("cf,batch,dtype,tol_img,tol_grad,min_cos", [(4, 4, torch.float32, 2e-5, 5e-2, 0.99999), (1, 6, torch.float32, 2e-5, 1e-1, 0.99999),
                                                                    (1, 20, torch.bfloat16, 8e-2, 0.35, 0.99)])
def test_generator_pair_pass_equals_two_forwards(cf, batch, dtype, tol_img, tol_grad, min_cos):
    """Generator.forward_pair (round 5; the two generator forwards of an iteration, /root/reference/model_wrapper.py:144-151 without
    gradient and :165-172 with, as one pass over 2B images below 256 x 256) against the two calls the reference makes, in its order:
    both image batches, every parameter gradient of the second forward, and ALL buffers - spectral-norm u / v after two power
    iterations, the BatchNorm running statistics after forward d's update then forward g's, num_batches_tracked + 2 - must agree
    (buffers: u / v and the counters bit for bit; the running statistics to the rounding of the activations they average)."""
    ops.set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(17)
    images, labels, masks = gu.golden_batches(4, 5)[0]
    reps = (batch + 3) // 4
    images = images.repeat(reps, 1, 1, 1)[:batch].cuda()
    labels = labels.repeat(reps, 1)[:batch].cuda()
    masks = [m.repeat(reps, *([1] * (m.dim() - 1)))[:batch].cuda() for m in masks]
    z_d, z_g = torch.randn(batch, 128, generator=g).cuda(), torch.randn(batch, 128, generator=g).cuda()
    seed_img = torch.randn(batch, 3, 256, 256, generator=g).cuda()
    V = sp.VGG16()
    _, _, Vsd = gu.synth_states({"cf": cf, "seed": 3})
    V.load_state_dict(Vsd)
    V.cuda().eval()
    with torch.no_grad():
        feats = V(images)
    outs = []
    for mode in ("two", "pair"):
        G = _gen(cf, 3)
        if mode == "two":
            with torch.no_grad():
                fake_d = G(z_d, feats, masks, labels)
            fake_g = G(z_g, feats, masks, labels)
        else:
            fake_g, fake_d = G.forward_pair(z_g, z_d, feats, masks, labels)
            assert not fake_d.requires_grad and fake_g.requires_grad
        fake_g.backward(seed_img.to(fake_g.dtype))
        G._bank.collect_extra()
        torch.cuda.synchronize()
        outs.append((fake_d.detach().float().clone(), fake_g.detach().float().clone(),
                     {n: p.grad.detach().float().clone() for n, p in G.named_parameters() if p.grad is not None},
                     {n: b.detach().clone() for n, b in G.named_buffers()}))
    (d0, g0, gr0, b0), (d1, g1, gr1, b1) = outs
    for a, b, tag in ((d0, d1, "forward d"), (g0, g1, "forward g")):
        err = float((a - b).abs().max())
        assert err <= tol_img, (tag, err)
    assert set(gr0) == set(gr1) and len(gr0) == len(list(G.parameters()))
    # The gradients.  The two runs launch other kernels (tile shapes and K splits follow the batch), so the activations differ in
    # the last bits (measured in fp32: every convolution input and output of the network within 1e-6 .. 8e-6 of the other run's,
    # scratch/dbg_gpair2.py) - and a LeakyReLU whose input sits within that distance of zero then takes the other slope in the
    # backward pass: isolated elements of an activation gradient move by 80 %, and a weight gradient that sums over few pixels feels
    # it (cf = 1, batch 6, the 512 -> 512 layer on 8 x 8 maps: 7.9e-2 of its largest element against the two-forward run, while that
    # run sits 3e-3 from the CPU oracle and the pair pass 7.8e-2 - one flipped element; every other weight tensor 2e-3 .. 8e-3; the
    # biases in front of a BatchNorm are rounding noise in all three, 100 % apart).  So: per WEIGHT tensor in relative L2, and - the
    # sharper statement - the whole gradient vector's cosine and norm; parity of the step itself is held by the reference goldens
    # (tests/test_gpu_step.py runs with the pair pass on).
    worst = 0.0
    for n in gr0:
        if not n.endswith("weight_orig"):
            continue
        err = float((gr0[n] - gr1[n]).double().norm() / gr0[n].double().norm().clamp_min(1e-30))
        worst = max(worst, err)
        assert err <= tol_grad, (n, err)
    a = torch.cat([gr0[n].double().flatten() for n in sorted(gr0)])
    b = torch.cat([gr1[n].double().flatten() for n in sorted(gr0)])
    cos = float((a * b).sum() / (a.norm() * b.norm()))
    print("generator pair pass vs two forwards (%s, cf=%s): worst weight-gradient rel-L2 %.2e, gradient cosine %.8f, norm ratio %.6f"
          % (dtype, cf, worst, cos, float(b.norm() / a.norm())))
    assert cos >= min_cos and abs(float(b.norm() / a.norm()) - 1.0) <= (1e-3 if dtype == torch.float32 else 2e-2), (cos, float(b.norm() / a.norm()))
    for n in b0:
        if n.endswith("weight_u") or n.endswith("weight_v") or n.endswith("num_batches_tracked"):
            assert torch.equal(b0[n], b1[n]), n
        else:
            assert torch.allclose(b0[n].float(), b1[n].float(), rtol=1e-5 if dtype == torch.float32 else 2e-2,
                                  atol=1e-7 if dtype == torch.float32 else 2e-3), n
    assert int(b1["final_block.1.num_batches_tracked"]) == 2


def test_fused_tail_under_autograd_matches_the_two_layers():
    """config.CFG.fuse_tail_grad (round 5): the generator's last two layers (conv3x3 -> LeakyReLU -> conv1x1 -> tanh,
    /root/reference/models.py:55-61) as ONE launch in the forward WITH autograd - it stores the 64-channel tensor too and the layers'
    autograd nodes are built around the results - against the two launches: images to the summation order of a 64-term dot product,
    every parameter gradient likewise."""
    from semantic_pyramid_for_image_generation_amd.config import CFG
    ops.set_compute_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(23)
    images, labels, masks = gu.golden_batches(2, 5)[0]
    images, labels, masks = images.cuda(), labels.cuda(), [m.cuda() for m in masks]
    z = torch.randn(2, 128, generator=g).cuda()
    seed_img = torch.randn(2, 3, 256, 256, generator=g).cuda()
    V = sp.VGG16()
    _, _, Vsd = gu.synth_states({"cf": 1, "seed": 3})
    V.load_state_dict(Vsd)
    V.cuda().eval()
    with torch.no_grad():
        feats = V(images)
    outs = []
    saved = CFG.fuse_tail_grad
    try:
        for on in (False, True):
            CFG.fuse_tail_grad = on
            G = _gen(1, 3)
            fake = G(z, feats, masks, labels)
            fake.backward(seed_img.to(fake.dtype))
            G._bank.collect_extra()
            outs.append((fake.detach().float().clone(), {n: p.grad.detach().float().clone() for n, p in G.named_parameters() if p.grad is not None}))
    finally:
        CFG.fuse_tail_grad = saved
    (f0, g0), (f1, g1) = outs
    assert float((f0 - f1).abs().max()) <= 2 ** -7                    # one bf16 step of a tanh output
    assert set(g0) == set(g1)
    # (weight tensors in relative L2 + the whole gradient's cosine: the biases in front of a BatchNorm are rounding noise in bf16 - two
    # runs of the same layers in another summation order differ by ~100 % there, see test_generator_pair_pass_equals_two_forwards)
    for n in g0:
        if n.endswith("weight_orig"):
            err = float((g0[n] - g1[n]).double().norm() / g0[n].double().norm().clamp_min(1e-30))
            assert err <= 0.25, (n, err)
    a = torch.cat([g0[n].double().flatten() for n in sorted(g0)])
    b = torch.cat([g1[n].double().flatten() for n in sorted(g0)])
    cos = float((a * b).sum() / (a.norm() * b.norm()))
    assert cos >= 0.995 and abs(float(b.norm() / a.norm()) - 1.0) <= 2e-2, (cos, float(b.norm() / a.norm()))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [
    # (cin, cout, ksize, h, w, pool2, up, act): one per convolution route the discriminator trunk takes
    (8, 64, 3, 64, 64, 0, False, 1),        # 8-channel input (conv3x3_cin8 in bf16)
    (64, 64, 3, 32, 64, 1, False, 0),       # Cout <= 64, pooled epilogue
    (64, 128, 3, 32, 32, 0, False, 1),      # Cout > 64 (ping-pong FAST / tall)
    (128, 128, 3, 32, 32, 1, False, 0),     # pooled epilogue, Cout > 64
    (128, 64, 3, 32, 32, 0, True, 0),       # input gradient of a pooled layer (in_up2)
    (256, 256, 3, 16, 16, 0, False, 1),     # 16-wide maps
    (256, 512, 3, 8, 8, 0, False, 1),       # small-spatial split-K igemm + finalize
    (512, 768, 3, 4, 4, 0, False, 0),
    (64, 128, 1, 32, 32, 0, False, 0),      # 1x1 direct
    (512, 768, 1, 2, 2, 0, False, 0),       # 1x1 split-K
    (72, 40, 3, 24, 24, 0, False, 1),       # odd shapes: generic igemm
])
def test_img_scale_every_route(dt, case):
    cin, cout, k, h, w, pool2, up, act = case
    if dt == torch.float32:
        cin = (cin + 3) // 4 * 4
    torch.manual_seed(cin * 7 + cout)
    n, split = 5, 2
    x = ops.nhwc_empty(n, cin, h // 2 if up else h, w // 2 if up else w, dt, "cuda").normal_()
    wt = (torch.randn(cout, k, k, cin, device="cuda") * 0.05).to(dt)
    b = torch.randn(cout, device="cuda")
    scales = torch.tensor([0.75, 1.5], device="cuda")
    ho, wo = (h // 2, w // 2) if pool2 else (h, w)
    res = ops.nhwc_empty(n, cout, ho, wo, dt, "cuda").normal_()
    xin = 0.25 * F.interpolate(x.float(), scale_factor=2, mode="nearest") if up else x.float()
    ref = F.conv2d(xin, wt.float().permute(0, 3, 1, 2), None, padding=k // 2)
    if pool2:
        ref = F.avg_pool2d(ref, 2)
    per = torch.cat([scales[:1].expand(split), scales[1:].expand(n - split)])[:, None, None, None]
    ref = ref * per + b[None, :, None, None] + res.float()
    if act:
        ref = F.leaky_relu(ref, 0.2)
    for _ in range(2):
        y = ops.nhwc_empty(n, cout, ho, wo, dt, "cuda").fill_(-7.0)
        ops._conv_launch(x, wt.data_ptr(), b, y, res, None, None, 0.2, n, h, w, cin, cout, cout, k, act, dt, pool2, up,
                         scales.data_ptr(), split)
        err = float((y.float() - ref).abs().max() / ref.abs().max())
        assert err <= (3e-4 if dt == torch.float32 else 1e-2), (case, err)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_fused_1x1_tail_matches_the_two_launches(dt):
    """sp_conv_params.tail_w: conv3x3 (64 -> 64) -> LeakyReLU -> conv1x1 (64 -> 3) -> tanh in one launch (the generator's last two
    layers in a pass without autograd) against the two separate launches on the same packed weights - equal up to the summation
    order of the 64-term dot product - and with the 64-channel output also requested."""
    import ctypes
    from semantic_pyramid_for_image_generation_amd import _lib as L
    torch.manual_seed(5)
    n, h, w = 3, 32, 64
    x = ops.nhwc_empty(n, 64, h, w, dt, "cuda").normal_()
    w3 = (torch.randn(64, 3, 3, 64, device="cuda") * 0.05).to(dt)
    w1 = (torch.randn(3, 1, 1, 64, device="cuda") * 0.2).to(dt)
    b3, b1 = torch.randn(64, device="cuda") * 0.1, torch.randn(3, device="cuda") * 0.1
    y64 = ops.nhwc_empty(n, 64, h, w, dt, "cuda")
    ops._conv_launch(x, w3.data_ptr(), b3, y64, None, None, None, 0.0, n, h, w, 64, 64, 64, 3, ops.ACT_LRELU, dt)
    ref = ops.nhwc_empty(n, 3, h, w, dt, "cuda")
    ops._conv_launch(y64, w1.data_ptr(), b1, ref, None, None, None, 0.0, n, h, w, 64, 3, 3, 1, ops.ACT_TANH, dt)
    for keep in (False, True):
        out = ops.nhwc_empty(n, 3, h, w, dt, "cuda").fill_(-7.0)
        mid = ops.nhwc_empty(n, 64, h, w, dt, "cuda").fill_(-7.0) if keep else None
        p = L.SpConvParams()
        p.x, p.w, p.bias, p.y = x.data_ptr(), w3.data_ptr(), b3.data_ptr(), (mid.data_ptr() if keep else None)
        p.n, p.h, p.w_, p.cin_p, p.cout, p.ldy, p.ksize, p.act, p.dtype = n, h, w, 64, 64, 64, 3, ops.ACT_LRELU, ops.sp_dtype(dt)
        p.tail_w, p.tail_bias, p.tail_y, p.tail_cout, p.tail_act, p.tail_ld = w1.data_ptr(), b1.data_ptr(), out.data_ptr(), 3, ops.ACT_TANH, 3
        for _ in range(2):
            L.call("sp_conv2d_igemm", ctypes.byref(p), ops.stream())
        torch.cuda.synchronize()
        err = float((out.float() - ref.float()).abs().max())
        assert err <= (1.6e-2 if dt == torch.bfloat16 else 2e-3), (keep, err)
        if keep:
            assert torch.equal(mid, y64)
    # outside its shapes the tail is refused loudly (no silent launch of a kernel that ignores it)
    p.cout = 128
    with pytest.raises(L.SempyrError):
        L.call("sp_conv2d_igemm", ctypes.byref(p), ops.stream())


@pytest.mark.parametrize("case", [(8, 4, 64, 128, 32, 64, False), (40, 20, 128, 128, 64, 64, False), (8, 4, 64, 64, 64, 64, True),
                                  (6, 2, 72, 40, 16, 32, False), (6, 3, 256, 256, 8, 8, False), (4, 2, 64, 128, 32, 32, False),
                                  (40, 20, 64, 128, 64, 64, False, 1), (8, 4, 128, 256, 32, 32, False, 1), (6, 1, 64, 64, 16, 16, False, 1),
                                  (40, 20, 8, 64, 128, 128, False, 3), (6, 2, 8, 64, 128, 128, False, 3), (5, 2, 8, 32, 24, 24, False, 3)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_pair_weight_gradient_equals_two_launches(case, dt):
    """sp_conv2d_wgrad_accum_pair: one launch of the row walker over both groups + a reduce pass per group (where the group boundary
    falls between two blocks), or the two groups one after the other - either way each group's (dW, dbias) must equal its own
    sp_conv2d_wgrad_accum call up to fp32 summation order.  Cases: the single-launch form (first three, the third with a pooled
    gradient), a boundary that does not align, a small map (per-tap kernel), fp32 storage; 1x1 layers and the padded-RGB first layer
    (streaming kernels: one launch over both groups when the boundary falls between two of their pixel splits)."""
    import ctypes
    from semantic_pyramid_for_image_generation_amd import _lib as L
    n, split, cin, cout, h, w, pooled = case[:7]
    ks = case[7] if len(case) > 7 else 3
    if dt == torch.float32 and pooled:
        pytest.skip("pooled gradients: 16-bit row walker only")
    torch.manual_seed(n * 100 + cin)
    pad = 4 if dt == torch.float32 else 8
    cp = (cout + pad - 1) // pad * pad
    x = ops.nhwc_empty(n, cin, h, w, dt, "cuda").normal_()
    hd, wd = (h // 2, w // 2) if pooled else (h, w)
    dy = ops.nhwc_empty(n, cp, hd, wd, dt, "cuda").normal_()
    ndw = cout * ks * ks * cin
    spd = ops.sp_dtype(dt)

    def single(lo, hi):
        buf = torch.zeros(ndw + cout + 8, dtype=torch.float32, device="cuda")
        wsf = ops.wgrad_workspace_floats(hi - lo, h, w, cin, cout, ks, dt)
        ws = torch.empty(max(wsf, 1), dtype=torch.float32, device="cuda")
        L.call("sp_conv2d_wgrad_accum_pooled" if pooled else "sp_conv2d_wgrad_accum", ops.ptr(x.narrow(0, lo, hi - lo)),
               ops.ptr(dy.narrow(0, lo, hi - lo)), ops.ptr(buf), ctypes.c_void_p(buf.data_ptr() + 4 * (ndw + 4)), ops.ptr(ws) if wsf else None,
               wsf, hi - lo, h, w, cin, cout, cp, ks, spd, ops.stream())
        return buf
    ra, rb = single(0, split), single(split, n)
    for _ in range(2):
        ba = torch.zeros(ndw + cout + 8, dtype=torch.float32, device="cuda")
        bb = torch.zeros(ndw + cout + 8, dtype=torch.float32, device="cuda")
        wsf = ops.wgrad_workspace_floats(n, h, w, cin, cout, ks, dt)
        ws = torch.empty(max(wsf, 1), dtype=torch.float32, device="cuda")
        L.call("sp_conv2d_wgrad_accum_pair", ops.ptr(x), ops.ptr(dy), ops.ptr(ba), ctypes.c_void_p(ba.data_ptr() + 4 * (ndw + 4)), ops.ptr(bb),
               ctypes.c_void_p(bb.data_ptr() + 4 * (ndw + 4)), ops.ptr(ws) if wsf else None, wsf, n, split, h, w, cin, cout, cp, ks,
               1 if pooled else 0, spd, ops.stream())
        torch.cuda.synchronize()
        for got, ref in ((ba, ra), (bb, rb)):
            err = float((got - ref).abs().max() / ref.abs().max())
            assert err <= 2e-5, (case, err)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_second_generator_forward_reuses_the_feature_mappings(dtype, tol):
    """Generator.map_mode (ModelWrapper: "stash" for the discriminator step's forward, "reuse" for the generator step's): the seven
    masked-feature mappings of the second forward are derived from the first one's (same pyramid, masks and weight_orig; sigma has
    moved on) by sp_rescale_bias, and their weight / bias gradients come from the stashed inputs.  Output, every gradient and the
    (u, v) buffers must equal the plain computation (fp32: rounding of one multiply-add per element)."""
    import semantic_pyramid_for_image_generation_amd as spm
    from semantic_pyramid_for_image_generation_amd import models
    ops.set_compute_dtype(dtype)
    meta = {"cf": 4, "seed": 7}
    Gsd, _, Vsd = gu.synth_states(meta)
    images, labels, masks = gu.golden_batches(4, 3)[0]
    images, labels, masks = images.cuda(), labels.cuda(), [m.cuda() for m in masks]
    z = torch.randn(2, 4, 128, generator=torch.Generator().manual_seed(1)).cuda()
    outs = []
    for reuse in (False, True):
        G, V = spm.Generator(channels_factor=4), spm.VGG16()
        G.load_state_dict(Gsd); V.load_state_dict(Vsd)
        G.cuda().train(); V.cuda().eval()
        G._bank.direct_grads = True
        with torch.no_grad():
            feats = V(images)
            if reuse:
                G.map_mode = "stash"
            G(input=z[0], features=feats, masks=masks, class_id=labels.float())
        if reuse:
            G.map_mode = "reuse"
        img = G(input=z[1], features=feats, masks=masks, class_id=labels.float())
        if reuse:
            assert G._map_stash is None and G.map_mode is None
        (img.float() ** 2).mean().backward()
        G._bank.collect_extra()
        torch.cuda.synchronize()
        outs.append((img.detach().float().clone(), {n: p.grad.detach().float().clone() for n, p in G.named_parameters() if p.grad is not None},
                     {n: b.detach().clone() for n, b in G.named_buffers()}))
    (i0, g0, b0), (i1, g1, b1) = outs
    # bf16: the two computations round differently (the reused mapping is rounded twice), and a 16-bit generator amplifies that like any
    # other storage rounding - the worst of 786 K pixels moves as far as the bf16 mode moves against fp32 (tests/test_gpu_step.py)
    assert float((i0 - i1).abs().max()) <= (tol if dtype == torch.float32 else 0.1)
    assert float(((i0 - i1) ** 2).mean().sqrt()) <= (tol if dtype == torch.float32 else 1e-2)
    assert set(g0) == set(g1)
    if dtype == torch.float32:
        dens = sorted(float(g0[n].abs().max()) for n in g0)
        floor = 1e-2 * dens[len(dens) // 2]
        for n in g0:
            err = float((g0[n] - g1[n]).abs().max()) / max(float(g0[n].abs().max()), floor)
            assert err <= 1e-3, (n, err)                 # 1e-7 differences of the mappings, amplified through the backward chain
    else:
        # 16-bit storage: parameters whose true gradient is zero (a bias in front of a BatchNorm) carry pure rounding noise - the
        # gradient as a whole is what can be compared
        a = torch.cat([g0[n].double().flatten() for n in sorted(g0)])
        b = torch.cat([g1[n].double().flatten() for n in sorted(g0)])
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        assert cos >= 0.99 and abs(float(a.norm() / b.norm()) - 1.0) <= 0.05, (cos, float(a.norm() / b.norm()))
    for n in b0:
        if n.endswith("weight_u") or n.endswith("weight_v"):
            assert torch.equal(b0[n], b1[n]), n           # the power iterations see the weights only
        elif b0[n].dtype.is_floating_point:                # BatchNorm running statistics follow the activations
            assert torch.allclose(b0[n], b1[n], rtol=1e-5 if dtype == torch.float32 else 2e-2, atol=1e-7 if dtype == torch.float32 else 2e-3), n


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
def test_vgg_pyramid_of_two_batches_in_one_pass(dtype, tol):
    """VGG16.forward_pair (ModelWrapper: the next iteration's real images ride in the generator step's pass over the fake images,
    model_wrapper.py:141,179): the features of both batches and the image gradient of the first equal two separate forwards (the
    network is frozen, eval mode; only kernel routes / tile counts differ - fp32: summation order)."""
    import semantic_pyramid_for_image_generation_amd as spm
    ops.set_compute_dtype(dtype)
    try:
        V = spm.VGG16()
        _, _, Vsd = gu.synth_states({"cf": 4, "seed": 7})
        V.load_state_dict(Vsd)
        V.cuda().eval()
        g = torch.Generator().manual_seed(3)
        a = (torch.rand(3, 3, 256, 256, generator=g) * 2 - 1).cuda()
        b = (torch.rand(3, 3, 256, 256, generator=g) * 2 - 1).cuda()
        seeds = None
        a1 = a.clone().requires_grad_(True)
        fa = V(a1)
        with torch.no_grad():
            fb = V(b)
        seeds = [torch.randn(f.shape, generator=torch.Generator().manual_seed(10 + i)).cuda().to(f.dtype) for i, f in enumerate(fa)]
        torch.autograd.backward(fa, seeds)
        a2 = a.clone().requires_grad_(True)
        pa, pb = V.forward_pair(a2, b)
        assert all(not f.requires_grad for f in pb)
        torch.autograd.backward(pa, seeds)
        for i, (x, y) in enumerate(zip(pa + pb, fa + fb)):
            assert x.shape == y.shape
            err = float((x.float() - y.float()).norm() / y.float().norm().clamp_min(1e-12))
            assert err <= tol, ("feature", i, err)
        err = float((a2.grad - a1.grad).norm() / a1.grad.norm())
        assert err <= 2 * tol, ("image gradient", err)
    finally:
        ops.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("pair", [False, True])
def test_vgg_pass_with_gradient_without_the_unpooled_tensors(pair, dtype):
    """config.CFG.vgg_pool_idx (round 5): in the pass with gradient a stage's last convolution stores the pooled output + the 2-bit
    window positions of the maxima (sp_conv_params.pool_idx) and the pooling's backward reads those (sp_maxpool2_bwd_idx) - the seven
    features and the image gradient are BIT-IDENTICAL to the pass that writes the unpooled tensors and pools them separately, alone
    and as the first group of a two-group pass (/root/reference/models.py:183-216, model_wrapper.py:179)."""
    import semantic_pyramid_for_image_generation_amd as spm
    from semantic_pyramid_for_image_generation_amd import models
    ops.set_compute_dtype(dtype)
    V = spm.VGG16()
    _, _, Vsd = gu.synth_states({"cf": 4, "seed": 7})
    V.load_state_dict(Vsd)
    V.cuda().eval()
    g = torch.Generator().manual_seed(5)
    a = (torch.rand(2, 3, 256, 256, generator=g) * 2 - 1).cuda()
    b = (torch.rand(2, 3, 256, 256, generator=g) * 2 - 1).cuda()
    results = []
    saved = models._VGG_POOL_IDX

    def run(on):
        models._VGG_POOL_IDX = on
        x = a.clone().requires_grad_(True)
        feats = V.forward_pair(x, b)[0] if pair else V(x)
        seeds = [torch.randn(f.shape, generator=torch.Generator().manual_seed(10 + i)).cuda().to(f.dtype) for i, f in enumerate(feats)]
        torch.autograd.backward(feats, seeds)
        return [f.detach().clone() for f in feats], x.grad.clone()
    try:
        # bit-identity is a statement about the two epilogue forms on the SAME schedule of fp32 sums: the K-split of a launch's last
        # partial round (csrc/conv_pp.hip, round 6) depends on the item count, which the two forms do not share (one launch over both
        # groups vs one per group) - it is switched off for this comparison and checked for closeness below
        ops.set_tuning(28, 0)
        try:
            for on in (False, True):
                results.append(run(on))
        finally:
            ops.set_tuning(28, -1)
        split_on = run(True)
    finally:
        models._VGG_POOL_IDX = saved
    (f0, g0), (f1, g1) = results
    for i, (u, v) in enumerate(zip(f0, f1)):
        assert torch.equal(u, v), ("feature", i)
    assert torch.equal(g0, g1) and float(g0.abs().max()) > 0
    for i, (u, v) in enumerate(zip(f1, split_on[0])):
        d = (u.float() - v.float()).abs()
        assert float(d.max()) <= 2.0 ** -6 * max(float(u.float().abs().max()), 1.0), ("feature with the K-split", i, float(d.max()))
    # (the image gradient runs through 13 ReLU / 5 max-pool routing decisions: a feature that rounds the other way re-routes isolated
    # elements, so the bound is on the whole tensor, not on its worst element)
    rel = float((g1.float() - split_on[1].float()).norm() / g1.float().norm())
    assert rel <= 5e-2, rel


def test_train_step_with_the_next_batch_announced_matches_plain_steps():
    """ModelWrapper.train_step(next_images_real=...) (config.CFG.vgg_pair): three iterations over two alternating batches, each
    announcing the next one's real images, against the same iterations without the announcement - fp32, same RNG stream: losses and
    pixels agree to summation order; and the pyramid computed ahead is really the one used (the VGG runs twice per step, not three times)."""
    import semantic_pyramid_for_image_generation_amd as spm
    from semantic_pyramid_for_image_generation_amd import models
    ops.set_compute_dtype(torch.float32)
    meta = {"cf": 4, "seed": 7}
    Gsd, Dsd, Vsd = gu.synth_states(meta)
    batches = [(im.cuda(), lb.cuda(), [m.cuda() for m in ms]) for im, lb, ms in gu.golden_batches(2, 5)[:2]]
    runs = []
    for announce in (False, True):
        G, D, V = spm.Generator(channels_factor=4), spm.Discriminator(channel_factor=4), spm.VGG16()
        G.load_state_dict(Gsd); D.load_state_dict(Dsd); V.load_state_dict(Vsd)
        G.cuda().train(); D.cuda().train(); V.cuda().eval()
        mw = spm.ModelWrapper(generator=G, discriminator=D, vgg16=V, training_dataset=None, validation_dataset=None,
                              generator_optimizer=spm.optim.Adam(G.parameters(), lr=1e-4), discriminator_optimizer=spm.optim.Adam(D.parameters(), lr=1e-4),
                              save_data_path=None)
        calls = [0]
        plain, pair = models.VGG16.forward, models.VGG16.forward_pair
        def count(self, *a, _f=None, **k):
            calls[0] += 1
            return _f(self, *a, **k)
        models.VGG16.forward = lambda self, *a, **k: count(self, *a, _f=plain, **k)
        models.VGG16.forward_pair = lambda self, *a, **k: count(self, *a, _f=pair, **k)
        try:
            torch.manual_seed(21)
            outs = []
            for t in range(3):
                im, lb, ms = batches[t % 2]
                nxt = batches[(t + 1) % 2][0] if announce else None
                out = mw.train_step(im, lb, ms, next_images_real=nxt)
                outs.append({k: v.detach().float().clone() for k, v in out.items()})
        finally:
            models.VGG16.forward, models.VGG16.forward_pair = plain, pair
        runs.append((outs, calls[0]))
    (plain_outs, plain_calls), (pair_outs, pair_calls) = runs
    assert plain_calls == 6 and pair_calls == 4, (plain_calls, pair_calls)       # V(real) + V(fake) per step | V(real_0), then one pass per step
    for a, b in zip(plain_outs, pair_outs):
        for k in a:
            if k.startswith("loss"):
                assert float(b[k]) == pytest.approx(float(a[k]), rel=2e-4, abs=1e-6), k
        assert float((a["images_fake"] - b["images_fake"]).abs().max()) <= 2e-4
