"""GPU parity of every HIP operator against the CPU oracle (oracle/sempyr_oracle.py, plain torch fp32).

All calls go through the C ABI of libsempyr.so (via the ctypes binding); the oracle is only the checker.
Tolerances: fp32 mode (exact-fp32 MFMA) 2e-4 of the tensor's max magnitude unless stated; bf16 mode
(bf16 storage + bf16 MFMA, fp32 accumulate) 3e-2.  Index / mask semantics (max-pool routing, feature
masking, concat) are checked bit-exactly on fp32.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import sempyr_oracle as O  # noqa: E402
from semantic_pyramid_for_image_generation_amd import _lib as L, models, ops, params  # noqa: E402

DTYPES = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-4, torch.bfloat16: 3e-2}


@pytest.fixture(autouse=True)
def _dtype_reset():
    yield
    ops.set_compute_dtype(torch.float32)


def dev(x, dtype):
    """CPU NCHW / rows fp32 -> device tensor in the kernels' layout."""
    x = x.detach().cuda()
    return ops.as_nhwc(x, dtype) if x.dim() == 4 else x.to(dtype).contiguous()


def host(t):
    return t.detach().float().cpu().contiguous()


def close(got, ref, tol, what="", robust=None):
    """Strict mode (fp32): max |err| <= tol * max|ref|.  Robust mode (default for bf16 tolerances, or on request):
    relative L2 error <= tol and at most 0.5% of the elements off by more than 8*tol*max|ref| - a ReLU/LeakyReLU/max-pool
    decision taken on a value that rounds across the kink legitimately flips isolated elements."""
    got, ref = host(got), ref.detach().float()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(float(ref.abs().max()), 1e-6)
    diff = (got - ref).abs()
    err = float(diff.max())
    if robust is None:
        robust = tol >= 1e-2
    if robust:
        l2 = float((got - ref).norm() / max(float(ref.norm()), 1e-12))
        frac = float((diff > 8 * tol * scale).float().mean())
        assert l2 <= tol and frac <= 5e-3, "%s: rel-L2 %.3e, max err %.3e (scale %.3e), outliers %.2e (tol %.1e)" % (what, l2, err, scale, frac, tol)
        return
    assert err <= tol * scale, "%s: max err %.3e vs scale %.3e (tol %.1e)" % (what, err, scale, tol)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def synth(module, seed, prefix=""):
    sd = params.synth_state_dict(module.state_dict(), seed)
    module.load_state_dict(sd)
    return {prefix + k: v.clone() for k, v in sd.items()}


def q(x, dtype):
    """Round-trip through the storage dtype so oracle and kernel see the same inputs."""
    return x.to(dtype).float()


# ----------------------------------------------------------------------------------------------
# spectral-normalised convolution: SN batch kernel + packing + igemm fwd + dgrad + wgrad + SN backward
# ----------------------------------------------------------------------------------------------
CONV_CASES = [  # (cin, cout, k, n, h, w)
    (64, 64, 3, 2, 16, 16), (128, 256, 3, 2, 8, 8), (256, 32, 1, 2, 16, 16), (512, 768, 3, 3, 4, 4),
    (128, 64, 1, 1, 8, 24), (64, 3, 1, 2, 16, 16), (8, 64, 3, 2, 32, 32), (520, 128, 3, 1, 8, 8),
    (64, 128, 3, 1, 40, 24), (64, 64, 3, 1, 64, 64), (96, 160, 3, 2, 32, 64), (128, 64, 3, 1, 16, 128),
    # 1x1 layers wide enough for the row-walker weight-gradient kernel (W % 32 == 0)
    (128, 64, 1, 2, 32, 32), (64, 136, 1, 1, 16, 64), (8, 64, 1, 2, 32, 32),
    # thin output on a big map (RGB head): routed to the halo-reuse kernels with idle MFMA rows
    (64, 3, 3, 4, 256, 256), (32, 3, 3, 5, 256, 256),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_sn_conv_forward_backward(case, dtype):
    cin, cout, k, n, h, w = case
    ops.set_compute_dtype(dtype)
    torch.manual_seed(1)
    m = models.SNConv2d(cin, cout, k).cuda()
    sd = synth(m, 7, "c.")
    S = O.make_state(sd)
    x = q(rnd(n, cin, h, w, seed=2), dtype).requires_grad_(True)
    gy = q(rnd(n, cout, h, w, seed=3), dtype)
    ref = O.sn_conv(S, "c", x, True, k // 2)
    ref.backward(gy)
    xd = dev(x, dtype).requires_grad_(True)
    y = m(xd)
    y.backward(dev(gy, dtype))
    tol = TOL[dtype]
    close(y, ref, tol, "y")
    close(m.weight_u, S["c.weight_u"], 1e-5, "u after power iteration")
    close(m.weight_v, S["c.weight_v"], 1e-5, "v after power iteration")
    close(xd.grad, x.grad, tol, "dx")
    close(m.weight_orig.grad, S["c.weight_orig"].grad, tol * 2, "dW_orig")
    close(m.bias.grad, S["c.bias"].grad, tol * 2, "dbias")


# The launches that dominate the benchmark step (bench.py: channel_factor 1, batch 20 per GPU), at their real sizes
REAL_SHAPE_CASES = [(256, 256, 3, 20, 64, 64), (512, 512, 3, 20, 32, 32), (64, 64, 3, 20, 256, 256), (512, 512, 3, 20, 16, 16),
                    (128, 128, 3, 20, 128, 128), (512, 512, 3, 20, 8, 8), (768, 768, 3, 20, 4, 4),
                    # the two-group passes' batch of 40 on 8 x 8 maps: 320 tiles of 64 x 64 -> the 128 co x 64 px tile + three K splits (round 5)
                    (512, 512, 3, 40, 8, 8), (768, 768, 3, 40, 4, 4)]


@pytest.mark.parametrize("case", REAL_SHAPE_CASES)
def test_sn_conv_real_benchmark_shapes_bf16(case):
    """Forward, input gradient and weight gradient of the fat 3x3 layers at the benchmark's own shapes (B = 20, bf16: the tall
    LDS-DMA kernels with hundreds of work items per launch, the row-walker weight gradient with 512 partial slabs, the
    split-K small-spatial kernels) against the oracle."""
    import os
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    test_sn_conv_forward_backward(case, torch.bfloat16)


TALL_CASES = [(64, 128, 3, 1, 32, 32), (96, 160, 3, 2, 32, 64), (40, 192, 3, 1, 16, 32), (520, 128, 3, 1, 16, 32),
              (128, 136, 3, 3, 48, 96),
              # 64-channel-output variant (a wave = 64 co x 2 rows), incl. the 3->8 padded first layer
              (64, 64, 3, 1, 32, 32), (8, 64, 3, 2, 32, 64), (72, 64, 3, 1, 16, 32), (64, 40, 3, 1, 16, 64),
              # more work items than blocks: the persistent loop carries the DMA pipeline across items (320 items / 256 blocks)
              (64, 64, 3, 10, 128, 128), (64, 128, 3, 10, 128, 128), (8, 64, 3, 5, 256, 128)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", TALL_CASES)
def test_conv3x3_tall_kernel_forced(case, dtype):
    """conv3x3_tall_kernel (LDS-DMA halo, 128 co x 16x32 px tiles) forced onto small layers: image borders, partial
    channel chunks (cin 40, 520), partial output-channel tiles (cout 160, 192, 136), forward and (via dgrad) backward."""
    ops.set_tuning(ops.TUNE_CONV_TALL, 2)
    try:
        test_sn_conv_forward_backward(case, dtype)
    finally:
        ops.set_tuning(ops.TUNE_CONV_TALL, -1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [c for c in TALL_CASES if c[1] > 64] + [(128, 128, 3, 2, 8, 32), (64, 256, 3, 1, 24, 64)])
def test_conv3x3_short_tall_kernel_forced(case, dtype):
    """conv3x3_tall_kernel<2, 8> (the halo kernel's 128 co x 8x32 tile on the LDS-DMA pipeline), tuning value 3."""
    ops.set_tuning(ops.TUNE_CONV_TALL, 3)
    try:
        test_sn_conv_forward_backward(case, dtype)
    finally:
        ops.set_tuning(ops.TUNE_CONV_TALL, -1)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_epilogue_act_and_residuals(dtype):
    ops.set_compute_dtype(dtype)
    m = models.SNConv2d(64, 128, 3).cuda()
    sd = synth(m, 11, "c.")
    S = O.make_state(sd)
    x = q(rnd(2, 64, 8, 8, seed=1), dtype).requires_grad_(True)
    r1 = q(rnd(2, 128, 8, 8, seed=2), dtype).requires_grad_(True)
    r2 = q(rnd(2, 128, 8, 8, seed=3), dtype).requires_grad_(True)
    gy = q(rnd(2, 128, 8, 8, seed=4), dtype)
    # a LeakyReLU mask taken from a bf16-rounded value flips isolated elements; the activation path is
    # dtype-independent code, so it is exercised in fp32 and the bf16 run checks the residual epilogue alone
    use_act = dtype == torch.float32
    pre = (O.sn_conv(S, "c", x, True, 1) + r1) + r2
    ref = O.lrelu(pre) if use_act else pre
    ref.backward(gy)
    xd, r1d, r2d = (dev(t, dtype).requires_grad_(True) for t in (x, r1, r2))
    y = m(xd, ops.ACT_LRELU if use_act else ops.ACT_NONE, r1d, r2d)
    y.backward(dev(gy, dtype))
    tol = TOL[dtype]
    close(y, ref, tol, "y")
    close(xd.grad, x.grad, tol, "dx")
    close(r1d.grad, r1.grad, tol, "dres1")
    close(r2d.grad, r2.grad, tol, "dres2")
    close(m.weight_orig.grad, S["c.weight_orig"].grad, 2 * tol, "dW")


PP_EPILOGUE_CASES = [  # (n, cin, cout, hw, act, residuals, mask, pool2, input-upsampled, bias)
    (2, 128, 128, 64, 1, 0, False, 0, False, True), (3, 64, 128, 64, 0, 2, False, 0, False, True), (3, 136, 128, 64, 1, 0, False, 0, False, True),
    (2, 128, 128, 64, 0, 0, True, 0, False, False), (2, 128, 256, 64, 0, 2, False, 1, False, True), (2, 64, 128, 64, 2, 0, False, 2, False, True),
    (2, 256, 128, 64, 0, 0, True, 0, True, False), (5, 32, 192, 32, 3, 0, False, 0, False, True), (2, 64, 72, 32, 1, 0, False, 0, False, True),
    (20, 256, 512, 32, 1, 1, False, 0, False, True),
    # pooling with a partial last channel tile (Cout % 16 == 0 is all the API asks): the groups past Cout must not be stored
    (2, 64, 80, 64, 2, 0, False, 2, False, True), (3, 72, 192, 32, 0, 1, False, 1, False, True), (2, 64, 48, 64, 0, 0, False, 1, False, True),
    # Cout <= 64: the eight-row-pair-wave form
    (2, 64, 64, 64, 1, 0, False, 0, False, True), (3, 128, 64, 128, 0, 2, False, 0, False, True), (2, 72, 64, 32, 0, 0, True, 0, False, False),
    (2, 64, 64, 64, 2, 0, False, 2, False, True), (2, 64, 64, 64, 0, 1, False, 1, False, True), (3, 32, 40, 32, 1, 0, False, 0, False, True),
]


@pytest.mark.parametrize("case", PP_EPILOGUE_CASES)
def test_conv3x3_pingpong_epilogue_variants_bf16(case):
    """The ping-pong 3x3 kernel (conv_pp.hip: both co-tile forms, the one-pass-per-operand epilogue and the general one) over its
    epilogue variants - bias, LeakyReLU / ReLU / tanh, one and two residuals, the activation-gradient mask, 2x2 average / max
    pooling, the pooled-gradient input - at shapes that reach it (W % 32 == 0): (a) against fp32 arithmetic on the same bf16
    operands, (b) BIT-IDENTICAL to the round-2 kernels (SP_TUNE_CONV_PP = 0), three launches each into a dirty output (the
    block is persistent: the second item of a block must not depend on what the first left behind)."""
    n, cin, cout, hw, act, res, mask, pool2, up, bias = case
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(5)
    hin = hw // 2 if up else hw
    ho = hw // 2 if pool2 else hw
    x = ops.nhwc_empty(n, cin, hin, hin, dt, "cuda").normal_(generator=g)
    w = (torch.randn(cout, 3, 3, cin, device="cuda", generator=g) * 0.05).to(dt)          # packed forward layout [co][tap][ci]
    b = torch.randn(cout, device="cuda", generator=g) if bias else None
    mk = lambda: ops.nhwc_empty(n, cout, ho, ho, dt, "cuda").normal_(generator=g)
    r1 = mk() if res >= 1 else None
    r2 = mk() if res >= 2 else None
    ms = mk() if mask else None

    def launch(y):
        ops._conv_launch(x, w.data_ptr(), b, y, r1, r2, ms, 0.2, n, hw, hw, cin, cout, cout, 3, act, dt, pool2, up)

    # (a) fp32 reference on the bf16 operands
    xin = x.float()
    if up:                                                     # the input stands for the gradient of a 2x2 average pooling
        xin = 0.25 * F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, w.float().permute(0, 3, 1, 2), b, padding=1)
    if pool2 == 1:
        ref = F.avg_pool2d(ref, 2)
    elif pool2 == 2:
        ref = F.max_pool2d(ref, 2)
    if ms is not None:
        ref = ref * torch.where(ms.float() > 0, 1.0, 0.2)
    if r1 is not None:
        ref = ref + r1.float()
    if r2 is not None:
        ref = ref + r2.float()
    ref = {0: lambda t: t, 1: lambda t: F.leaky_relu(t, 0.2), 2: F.relu, 3: torch.tanh}[act](ref)
    y = ops.nhwc_empty(n, cout, ho, ho, dt, "cuda").fill_(-7.0)
    launch(y)
    close(y, ref.cpu(), 8e-3, "ping-pong kernel vs fp32")
    # (b) bit-identical to the round-2 kernels (without the K-split of the last round, which changes the order of the fp32 sums)
    ops.set_tuning(21, 0)
    try:
        y0 = ops.nhwc_empty(n, cout, ho, ho, dt, "cuda").fill_(3.0)
        launch(y0)
    finally:
        ops.set_tuning(21, -1)
    ops.set_tuning(TUNE_CONV_PP_SPLIT, 0)
    try:
        for rep in range(3):
            y1 = ops.nhwc_empty(n, cout, ho, ho, dt, "cuda").fill_(-7.0)
            launch(y1)
            assert torch.equal(y0, y1), (case, rep, int((y0 != y1).sum()))
    finally:
        ops.set_tuning(TUNE_CONV_PP_SPLIT, -1)


TUNE_CONV_PP_SPLIT = 28
PP_SPLIT_CASES = [  # (n, cin, cout, h, w, act, residuals, mask, pool2, input-upsampled, bias, two groups) -> items, tail items R, pieces P
    (20, 512, 512, 32, 32, 1, 1, False, 0, False, True, False),     # 320 items: R = 64, P = 4 (the metric's 32 x 32 layers)
    (20, 256, 256, 64, 64, 0, 0, True, 0, False, False, False),     # 640: R = 128, P = 2
    (40, 512, 512, 32, 32, 0, 2, False, 0, False, True, True),      # 640, two groups
    (20, 128, 64, 128, 128, 1, 0, False, 0, False, True, False),    # 64 co x 16 x 32 px items: 640
    (20, 128, 256, 64, 64, 0, 2, False, 1, False, True, False),     # average pooling in the (general) epilogue
    (20, 64, 128, 64, 64, 2, 0, False, 2, False, True, False),      # maximum pooling; K of two chunks: pieces of one chunk are refused -> unsplit
    (20, 256, 128, 64, 64, 0, 0, True, 0, True, False, False),      # pooled-gradient input: 320
    (33, 136, 128, 64, 64, 1, 0, False, 0, False, True, False),     # 528: R = 16, five chunks (partial last chunk) -> P = 2
    (10, 520, 320, 64, 64, 0, 1, False, 0, False, True, False),     # partial channel tile (320 = 2.5 x 128), 17 chunks: 480 items, R = 224 -> unsplit
    (9, 200, 384, 32, 64, 3, 0, False, 0, False, True, False),      # tanh (general epilogue), 9 x 4 x 2 x 3 = 216 items < 256 -> unsplit
    (11, 264, 384, 32, 64, 1, 0, False, 0, False, True, False),     # 264 items: R = 8, P = 4 of nine chunks (2, 2, 2, 3)
    # less than one round: every item is a tail item, the grid is items x pieces
    (4, 256, 256, 64, 64, 1, 1, False, 0, False, True, False),      # 128 items x 2
    (20, 512, 512, 16, 16, 1, 0, True, 0, False, True, False),      # 128 co x 16 x 16 px items: 80 x 3 (the metric's 16 x 16 layers)
    (20, 520, 512, 16, 16, 0, 2, False, 0, False, True, True),      # ... partial last chunk, two groups
    (3, 96, 128, 32, 64, 1, 0, False, 0, False, True, False),       # 24 items x 3 of three chunks?  no: pieces of one chunk are refused -> unsplit
    (3, 160, 128, 32, 64, 1, 0, False, 0, False, True, False),      # 24 items, five chunks: x 2
]


@pytest.mark.parametrize("case", PP_SPLIT_CASES)
def test_conv3x3_pingpong_tail_split_bf16(case):
    _tail_split_case(case)


def test_conv3x3_tail_split_on_two_streams_bf16():
    """The K-split keeps its counters in the CALLER's zero-at-rest area, one per stream (sp_conv_params.split_sync; the library has no
    device-side state): two streams launching split convolutions at the same time get what one stream gets."""
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(31)
    n, cin, cout, hw = 20, 256, 256, 64                                   # 640 items: R = 128, two pieces
    xs = [ops.nhwc_empty(n, cin, hw, hw, dt, "cuda").normal_(generator=g) for _ in range(2)]
    ws = [(torch.randn(cout, 3, 3, cin, device="cuda", generator=g) * 0.05).to(dt) for _ in range(2)]

    def launch(i, y):
        ops._conv_launch(xs[i], ws[i].data_ptr(), None, y, None, None, None, 0.2, n, hw, hw, cin, cout, cout, 3, 1, dt)
    ref = []
    for i in range(2):
        y = ops.nhwc_empty(n, cout, hw, hw, dt, "cuda")
        launch(i, y)
        ref.append(y)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]
    for rep in range(6):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                y = ops.nhwc_empty(n, cout, hw, hw, dt, "cuda").fill_(-3.0)
                launch(i, y)
                outs[i].append(y)
    torch.cuda.synchronize()
    assert len({k for k in ops._SPLIT_SYNC if k[1] in (streams[0].cuda_stream, streams[1].cuda_stream)}) == 2
    for i in range(2):
        for y in outs[i]:
            assert torch.equal(y, ref[i]), i
    for t in ops._SPLIT_SYNC.values():
        assert int(t.abs().sum()) == 0                                    # every launch left its counters clean


PPW_SPLIT_CASES = [  # the same for conv_ppw.hip's 16-row items (forced: SP_TUNE_CONV_PPW = 2): (n, cin, cout, h, w, act, res, mask, pool2, up, bias, groups)
    (20, 256, 256, 64, 64, 1, 1, False, 0, False, True, False),     # 320 items: R = 64
    (40, 512, 512, 32, 32, 0, 2, False, 0, False, True, True),      # 320, two groups, 16 chunks
    (40, 256, 128, 64, 64, 0, 0, True, 0, False, False, False),     # 320, mask operand
    (20, 512, 256, 32, 32, 1, 0, False, 0, False, True, False),     # 80 items: less than a round, x 3
    (20, 128, 256, 64, 64, 0, 2, False, 1, False, True, False),     # average pooling (the POOL instantiation), 320 items of four chunks
    (11, 264, 384, 32, 64, 1, 0, False, 0, False, True, False),     # 132 items: no split - partial last chunk, unsplit control
    (33, 136, 128, 64, 64, 2, 0, False, 2, False, True, False),     # 264 items: R = 8, five chunks, maximum pooling
    (20, 256, 128, 64, 64, 0, 0, True, 0, True, False, False),      # pooled-gradient input, 160 items: unsplit (R > 128)
    (36, 192, 128, 64, 64, 1, 0, False, 0, False, True, False),     # 288 items: R = 32, six chunks
]


@pytest.mark.parametrize("case", PPW_SPLIT_CASES)
def test_conv3x3_ppw_tail_split_bf16(case):
    """conv_ppw.hip's K-split of the last partial round (256 KB pieces: 128 accumulator registers per lane), forced onto every launch."""
    ops.set_tuning(26, 2)
    try:
        _tail_split_case(case, want_route="conv3x3_ppw")
    finally:
        ops.set_tuning(26, -1)


def _tail_split_case(case, want_route=None):
    """conv_pp.hip's K-split of the last, partial round of work items (round 6): launches whose item count leaves a remainder over the
    256 persistent blocks - the metric's batch of 20 puts most mid-network layers there.  (a) against fp32 arithmetic on the same
    bf16 operands; (b) against the unsplit launch (SP_TUNE_CONV_PP_SPLIT = 0): the same values up to the order of the fp32 partial sums;
    (c) five launches into dirty outputs are bit-identical (pieces are added in a fixed order; the counters are clean after every
    launch), interleaved with an unrelated launch that reuses counters and scratch."""
    n, cin, cout, h, w_, act, res, mask, pool2, up, bias, groups = case
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(23)
    ho, wo = (h // 2, w_ // 2) if pool2 else (h, w_)
    hin, win = (h // 2, w_ // 2) if up else (h, w_)
    x = ops.nhwc_empty(n, cin, hin, win, dt, "cuda").normal_(generator=g)
    w = (torch.randn(cout, 3, 3, cin, device="cuda", generator=g) * 0.05).to(dt)
    b = torch.randn(cout, device="cuda", generator=g) if bias else None
    mk = lambda: ops.nhwc_empty(n, cout, ho, wo, dt, "cuda").normal_(generator=g)
    r1 = mk() if res >= 1 else None
    r2 = mk() if res >= 2 else None
    ms = mk() if mask else None
    split = n // 2
    scales = torch.tensor([1.0, 0.8125], device="cuda") if groups else None

    def launch(y):
        ops._conv_launch(x, w.data_ptr(), b, y, r1, r2, ms, 0.2, n, h, w_, cin, cout, cout, 3, act, dt, pool2, up,
                         img_scale=scales.data_ptr() if groups else 0, img_split=split if groups else 0)

    xin = x.float()
    if up:
        xin = 0.25 * F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, w.float().permute(0, 3, 1, 2), None, padding=1)
    if groups:
        ref[split:] *= 0.8125
    if b is not None:
        ref = ref + b.view(1, -1, 1, 1)
    if pool2 == 1:
        ref = F.avg_pool2d(ref, 2)
    elif pool2 == 2:
        ref = F.max_pool2d(ref, 2)
    if ms is not None:
        ref = ref * torch.where(ms.float() > 0, 1.0, 0.2)
    if r1 is not None:
        ref = ref + r1.float()
    if r2 is not None:
        ref = ref + r2.float()
    ref = {0: lambda t: t, 1: lambda t: F.leaky_relu(t, 0.2), 2: F.relu, 3: torch.tanh}[act](ref)
    ops.set_tuning(TUNE_CONV_PP_SPLIT, 0)
    try:
        y0 = ops.nhwc_empty(n, cout, ho, wo, dt, "cuda").fill_(3.0)
        launch(y0)
    finally:
        ops.set_tuning(TUNE_CONV_PP_SPLIT, -1)
    close(y0, ref.cpu(), 8e-3, "unsplit launch vs fp32")
    if want_route is not None:
        assert L.lib().sp_last_route().decode().startswith(want_route), L.lib().sp_last_route().decode()
    # an unrelated split launch between the repetitions: it shares the counters and (through the allocator) the scratch
    xo = ops.nhwc_empty(20, 64, 32, 32, dt, "cuda").normal_(generator=g)
    wo_ = (torch.randn(512, 3, 3, 64, device="cuda", generator=g) * 0.05).to(dt)
    yo = ops.nhwc_empty(20, 512, 32, 32, dt, "cuda")
    first = None
    for rep in range(5):
        y1 = ops.nhwc_empty(n, cout, ho, wo, dt, "cuda").fill_(-7.0)
        launch(y1)
        ops._conv_launch(xo, wo_.data_ptr(), None, yo, None, None, None, 0.2, 20, 32, 32, 64, 512, 512, 3, 0, dt, 0, False)
        if first is None:
            first = y1
            close(y1, ref.cpu(), 8e-3, "split launch vs fp32")
            d = (y1.float() - y0.float()).abs()
            # same products, other order of the fp32 sums: at most one bf16 rounding step apart, and rarely
            assert float((d / (y0.float().abs() + 1.0)).max()) <= 2.0 ** -7, float(d.max())
            assert float((d > 0).float().mean()) < 0.05, float((d > 0).float().mean())
        else:
            assert torch.equal(first, y1), (case, rep, int((first != y1).sum()))
    # the hand-over's re-read path (the closing piece stores and counts like the others; otherwise only a delayed block takes it):
    # the same bits as the usual path
    ops.set_tuning(TUNE_CONV_PP_SPLIT, 3)
    try:
        for rep in range(2):
            y3 = ops.nhwc_empty(n, cout, ho, wo, dt, "cuda").fill_(5.0)
            launch(y3)
            assert torch.equal(first, y3), (case, "re-read path", rep, int((first != y3).sum()))
    finally:
        ops.set_tuning(TUNE_CONV_PP_SPLIT, -1)


PPW_CASES = [  # (n, cin, cout, h, w, act, residuals, mask, input-upsampled, bias, two groups, pool2)
    (2, 128, 128, 64, 64, 1, 0, False, False, True, False, 0), (3, 64, 128, 32, 64, 0, 2, False, False, True, False, 0),
    (3, 136, 192, 32, 32, 2, 0, False, False, True, False, 0), (2, 128, 256, 64, 64, 0, 0, True, False, False, False, 0),
    (2, 256, 128, 64, 64, 0, 0, True, True, False, False, 0), (5, 32, 144, 16, 32, 1, 1, False, False, True, False, 0),
    (4, 72, 128, 32, 32, 0, 0, False, False, True, True, 0), (20, 512, 512, 32, 32, 1, 1, True, False, True, True, 0),
    (40, 64, 128, 32, 32, 0, 0, False, False, False, False, 0),
    # pooled epilogues: average (+ residual at the pooled resolution, two groups: the discriminator's pair pass), maximum, a partial channel tile
    (4, 128, 128, 64, 64, 0, 1, False, False, True, True, 1), (3, 64, 256, 32, 64, 2, 0, False, False, True, False, 2),
    (2, 96, 208, 32, 32, 0, 2, False, False, True, False, 1),
]


@pytest.mark.parametrize("case", PPW_CASES)
def test_conv3x3_pingpong_four_row_waves_bf16(case):
    """conv_ppw.hip (128 co x 16 x 32 px per block, 64 co x 4 rows per wave, stages cut into two column halves, three-slot weight
    ring) FORCED onto every launch its epilogue covers (SP_TUNE_CONV_PPW = 2) - bias, LeakyReLU / ReLU, residuals, the
    activation-gradient mask, the pooled-gradient input, 2x2 average / max pooling in the epilogue, the per-group accumulator scales of a
    two-group batch, partial channel tiles
    and partial K chunks, one and many items per block: (a) against fp32 arithmetic on the same bf16 operands, (b) BIT-IDENTICAL to
    the round-2 kernels (SP_TUNE_CONV_PP = 0), three launches each into a dirty output."""
    n, cin, cout, h, w_, act, res, mask, up, bias, groups, pool2 = case
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(11)
    ho, wo = (h // 2, w_ // 2) if pool2 else (h, w_)
    hin, win = (h // 2, w_ // 2) if up else (h, w_)
    x = ops.nhwc_empty(n, cin, hin, win, dt, "cuda").normal_(generator=g)
    w = (torch.randn(cout, 3, 3, cin, device="cuda", generator=g) * 0.05).to(dt)
    b = torch.randn(cout, device="cuda", generator=g) if bias else None
    mk = lambda: ops.nhwc_empty(n, cout, ho, wo, dt, "cuda").normal_(generator=g)
    r1 = mk() if res >= 1 else None
    r2 = mk() if res >= 2 else None
    ms = mk() if mask else None
    split = n // 2
    scales = torch.tensor([1.0, 0.8125], device="cuda") if groups else None

    def launch(y):
        ops._conv_launch(x, w.data_ptr(), b, y, r1, r2, ms, 0.2, n, h, w_, cin, cout, cout, 3, act, dt, pool2, up,
                         img_scale=scales.data_ptr() if groups else 0, img_split=split if groups else 0)

    xin = x.float()
    if up:
        xin = 0.25 * F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, w.float().permute(0, 3, 1, 2), None, padding=1)
    if groups:
        ref[split:] *= 0.8125
    if pool2:
        ref = F.avg_pool2d(ref, 2) if pool2 == 1 else F.max_pool2d(ref, 2)
    if b is not None:
        ref = ref + b.view(1, -1, 1, 1)
    if ms is not None:
        ref = ref * torch.where(ms.float() > 0, 1.0, 0.2)
    if r1 is not None:
        ref = ref + r1.float()
    if r2 is not None:
        ref = ref + r2.float()
    ref = {0: lambda t: t, 1: lambda t: F.leaky_relu(t, 0.2), 2: F.relu}[act](ref)
    ops.set_tuning(21, 0)
    try:
        y0 = ops.nhwc_empty(n, cout, ho, wo, dt, "cuda").fill_(3.0)
        launch(y0)
    finally:
        ops.set_tuning(21, -1)
    ops.set_tuning(L.TUNE_KEYS["SP_CONV_PPW"], 2)
    ops.set_tuning(TUNE_CONV_PP_SPLIT, 0)         # (bit-identity is about one order of the fp32 sums: the K-split of the last round has its own test)
    try:
        for rep in range(3):
            y1 = ops.nhwc_empty(n, cout, ho, wo, dt, "cuda").fill_(-7.0)
            launch(y1)
            assert "conv3x3_ppw" in L.lib().sp_last_route().decode(), L.lib().sp_last_route().decode()
            if rep == 0:
                close(y1, ref.cpu(), 8e-3, "four-row-wave ping-pong kernel vs fp32")
            assert torch.equal(y0, y1), (case, rep, int((y0 != y1).sum()))
    finally:
        ops.set_tuning(L.TUNE_KEYS["SP_CONV_PPW"], -1)
        ops.set_tuning(TUNE_CONV_PP_SPLIT, -1)


@pytest.mark.parametrize("case", [(20, 128, 512, 1, 0, False, True), (16, 520, 512, 0, 2, False, True), (32, 72, 256, 2, 0, True, False),
                                  (9, 64, 1024, 1, 1, False, True)])
def test_conv3x3_pingpong_16_wide_tiles_bf16(case):
    """16 x 16 maps with enough (image, 128-channel tile) items: the ping-pong kernel on 128 co x 16 x 16 px tiles (conv_pp.hip,
    FW = 1: a wave owns 4 rows x 16 columns, halo pitch 20) against fp32 arithmetic on the same bf16 operands, and three launches
    into dirty outputs that must agree bit for bit."""
    n, cin, cout, act, res, mask, bias = case
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(7)
    x = ops.nhwc_empty(n, cin, 16, 16, dt, "cuda").normal_(generator=g)
    w = (torch.randn(cout, 3, 3, cin, device="cuda", generator=g) * 0.05).to(dt)
    b = torch.randn(cout, device="cuda", generator=g) if bias else None
    mk = lambda: ops.nhwc_empty(n, cout, 16, 16, dt, "cuda").normal_(generator=g)
    r1 = mk() if res >= 1 else None
    r2 = mk() if res >= 2 else None
    ms = mk() if mask else None
    ref = F.conv2d(x.float(), w.float().permute(0, 3, 1, 2), b, padding=1)
    if ms is not None:
        ref = ref * torch.where(ms.float() > 0, 1.0, 0.2)
    if r1 is not None:
        ref = ref + r1.float()
    if r2 is not None:
        ref = ref + r2.float()
    ref = {0: lambda t: t, 1: lambda t: F.leaky_relu(t, 0.2), 2: F.relu}[act](ref)
    ys = []
    for rep in range(3):
        y = ops.nhwc_empty(n, cout, 16, 16, dt, "cuda").fill_(-7.0)
        ops._conv_launch(x, w.data_ptr(), b, y, r1, r2, ms, 0.2, n, 16, 16, cin, cout, cout, 3, act, dt, 0, False)
        ys.append(y)
    close(ys[0], ref.cpu(), 8e-3, "16-wide ping-pong tiles vs fp32")
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])


@pytest.mark.parametrize("case", [(128, 256, 3, 2, 8, 8), (520, 128, 3, 1, 8, 8), (64, 72, 3, 5, 8, 8), (512, 512, 3, 20, 8, 8), (72, 64, 3, 3, 16, 16)])
def test_wgrad_row_walker_narrow_maps_bf16(case):
    """Row-walking weight-gradient kernel on narrow maps (several images side by side in one 32-pixel strip): 8 x 8 maps are off
    by default (tuning value 3 enables them), 16 x 16 maps with an odd image count (a half-empty last strip) ride along."""
    ops.set_tuning(ops.TUNE_WGRAD_ROWS, 3)
    try:
        test_sn_conv_forward_backward(case, torch.bfloat16)
    finally:
        ops.set_tuning(ops.TUNE_WGRAD_ROWS, -1)


SPLITK_1X1_CASES = [(768, 512, 1, 20, 2, 2), (512, 512, 1, 20, 4, 4), (256, 256, 1, 20, 8, 8), (520, 136, 1, 1, 5, 7), (256, 30, 1, 1, 6, 6),
                    (128, 64, 1, 3, 3, 3), (512, 768, 1, 2, 2, 2),
                    # channel_factor 0.5: Cin / Cout beyond 1024 (the K-steps of a wave in two or more chunks; 1040: a ragged last chunk) -
                    # forward on 1536 / 1040 input channels, and the 1024 -> 1536 layer's input gradient (K = 1536) through dgrad
                    (1536, 1024, 1, 40, 2, 2), (1024, 1536, 1, 40, 2, 2), (1040, 72, 1, 3, 3, 5)]


@pytest.mark.parametrize("case", [(64, 64, 3, 2, 32, 32), (128, 96, 3, 2, 16, 16), (256, 64, 1, 2, 16, 16), (32, 48, 3, 3, 32, 32)])
def test_wgrad_c_abi_without_workspace_fp32(case):
    """Round-2 ADVICE (medium): sp_conv2d_wgrad has no workspace argument and sp_conv2d_wgrad_accum documents its scratch as
    optional, but the deterministic fp32 mode used to reject every split plan without scratch.  Straight through the C ABI:
    fp32, no workspace -> one ordered split per tile; result vs torch's conv2d weight gradient, twice bit-identical."""
    import ctypes
    from semantic_pyramid_for_image_generation_amd import _lib as L
    cin, cout, k, n, h, w = case
    x, dy = rnd(n, cin, h, w, seed=1), rnd(n, cout, h, w, seed=2)
    xr = x.clone().requires_grad_(False)
    wref = torch.zeros(cout, cin, k, k, requires_grad=True)
    F.conv2d(xr, wref, padding=k // 2).backward(dy)
    want = wref.grad.permute(0, 2, 3, 1).reshape(cout, k * k, cin)                  # [cout][tap][cin]
    xd, dyd = dev(x, torch.float32), dev(dy, torch.float32)
    outs = []
    for entry in ("sp_conv2d_wgrad", "sp_conv2d_wgrad_accum"):
        for rep in range(2):
            dw = torch.zeros(cout, k * k, cin, device="cuda")
            if entry == "sp_conv2d_wgrad":
                dw.fill_(7.0)                                                         # the call zeroes dw itself
                L.call(entry, ops.ptr(xd), ops.ptr(dyd), ops.ptr(dw), n, h, w, cin, cout, cout, k, L.SP_F32, ops.stream())
            else:
                L.call(entry, ops.ptr(xd), ops.ptr(dyd), ops.ptr(dw), None, None, 0, n, h, w, cin, cout, cout, k, L.SP_F32, ops.stream())
            torch.cuda.synchronize()
            close(dw, want, 2e-4, "%s %r" % (entry, case))
            outs.append(dw)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[2], outs[3])


@pytest.mark.parametrize("case", SPLITK_1X1_CASES)
def test_conv1x1_splitk_kernel_bf16(case):
    """conv1x1_splitk_kernel (small maps: the four waves of a block split K, partial tiles meet in LDS) at the step's own deep
    shapes and on ragged ones: pixel counts that are no multiple of 32, Cin 520 (17 K-steps: unequal shares, a partial last
    chunk), Cout 136 / 30 (partial channel tiles, scalar stores); forward here, input and weight gradients through dgrad."""
    test_sn_conv_forward_backward(case, torch.bfloat16)


def test_conv1x1_splitk_matches_direct_kernel_with_residuals():
    """Same layer through both 1x1 kernels (tuning key CONV1X1_SPLITK) with bias, two residuals and the activation fused:
    the results agree to bf16 rounding of one output."""
    ops.set_compute_dtype(torch.bfloat16)
    m = models.SNConv2d(512, 192, 1).cuda()
    synth(m, 13, "c.")
    x = dev(rnd(5, 512, 4, 4, seed=1), torch.bfloat16)
    r1 = dev(rnd(5, 192, 4, 4, seed=2), torch.bfloat16)
    r2 = dev(rnd(5, 192, 4, 4, seed=3), torch.bfloat16)
    with torch.no_grad():
        m.eval()
        y_split = m(x, ops.ACT_LRELU, r1, r2).float()
        ops.set_tuning(ops.TUNE_CONV1X1_SPLITK, 0)
        try:
            y_direct = m(x, ops.ACT_LRELU, r1, r2).float()
        finally:
            ops.set_tuning(ops.TUNE_CONV1X1_SPLITK, -1)
    assert float((y_split - y_direct).abs().max()) <= 2 ** -7 * float(y_direct.abs().max())


WGRAD_1X1_CASES = [(8, 64, 3, 2, 128, 128), (8, 48, 3, 3, 64, 128), (72, 128, 1, 3, 41, 24), (64, 136, 1, 2, 32, 48), (8, 64, 1, 20, 128, 128), (64, 3, 1, 20, 128, 128), (128, 256, 1, 20, 32, 32),
                   (256, 32, 1, 20, 16, 16)]


@pytest.mark.parametrize("case", WGRAD_1X1_CASES)
def test_conv1x1_streaming_weight_gradient_bf16(case):
    """wgrad1x1_stream_kernel / wgrad3x3_cin8_stream_kernel (conv_wgrad_1x1.hip; the first two cases: 3x3 on the 8-channel input with
    its im2col X tile): long pixel ranges per block, LDS-DMA ring, transposed LDS reads, slabs + ordered reduce.  Ragged pixel counts (a partial last 64-pixel stage, 9 splits on 8 XCDs), partial channel tiles (Cin 72 / 8, Cout 136 / 3)
    and the step's own shapes; the weight and bias gradients of two runs are bit-identical (no atomics)."""
    test_sn_conv_forward_backward(case, torch.bfloat16)
    cin, cout, k, n, h, w = case
    ops.set_compute_dtype(torch.bfloat16)
    m = models.SNConv2d(cin, cout, k).cuda()
    synth(m, 7, "c.")
    x = dev(rnd(n, cin, h, w, seed=2), torch.bfloat16)
    gy = dev(rnd(n, cout, h, w, seed=3), torch.bfloat16)
    grads = []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        synth(m, 7, "c.")                       # same u / v: the power iteration of the first run must not leak into the second
        m(x).backward(gy)
        grads.append((m.weight_orig.grad.clone(), m.bias.grad.clone()))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])


def test_deferred_slab_reductions_are_bit_identical_and_scoped():
    """sp_wgrad_reduce_defer / _flush (include/sempyr.h): three streaming weight-gradient launches (1x1, 1x1 with a partial channel tile,
    3x3 on the 8-channel input) queue their reductions and ONE flush launch adds them up - bit-identical to the immediate form; a
    second launch onto a dW that is already queued keeps its own reduce kernel; a dropped queue launches nothing; outside the scope
    the C ABI reduces at once."""
    dt = torch.bfloat16
    lib = L.lib()
    g = torch.Generator(device="cuda").manual_seed(3)
    cases = [(64, 128, 1, 6, 64, 64), (72, 136, 1, 5, 32, 64), (8, 64, 3, 4, 64, 128)]
    ops_in = []
    for cin, cout, k, n, h, w in cases:
        x = ops.nhwc_empty(n, cin, h, w, dt, "cuda").normal_(generator=g)
        dy = ops.nhwc_empty(n, cout, h, w, dt, "cuda").normal_(generator=g)
        floats = ops.wgrad_workspace_floats(n, h, w, cin, cout, k, dt)
        assert floats > 0, "the case must run with pixel splits"
        ops_in.append((x, dy, floats, cin, cout, k, n, h, w))

    def run(deferred, twice=False):
        outs, keep = [], []
        for x, dy, floats, cin, cout, k, n, h, w in ops_in:
            dw = torch.zeros(cout * k * k * cin, device="cuda")
            db = torch.zeros(cout, device="cuda")
            for _ in range(2 if twice else 1):
                ws = torch.empty(floats, device="cuda")
                keep.append(ws)
                if deferred:
                    L.call("sp_wgrad_reduce_defer", 1)
                L.call("sp_conv2d_wgrad_accum", ops.ptr(x), ops.ptr(dy), ops.ptr(dw), ops.ptr(db), ops.ptr(ws), floats, n, h, w, cin, cout, cout, k,
                       ops.sp_dtype(dt), ops.stream())
                L.call("sp_wgrad_reduce_defer", 0)
            outs.append((dw, db))
        return outs, keep

    ref, _ = run(False)
    torch.cuda.synchronize()
    assert int(lib.sp_wgrad_reduce_pending()) == 0
    got, keep = run(True)
    assert int(lib.sp_wgrad_reduce_pending()) == 3
    torch.cuda.synchronize()
    assert all(float(dw.abs().max()) == 0.0 for dw, _ in got), "nothing may reach dW before the flush"
    L.call("sp_wgrad_reduce_flush", 1, ops.stream())
    assert int(lib.sp_wgrad_reduce_pending()) == 0
    torch.cuda.synchronize()
    for (dw0, db0), (dw1, db1) in zip(ref, got):
        assert float(dw0.abs().max()) > 0
        assert torch.equal(dw0, dw1) and torch.equal(db0, db1)
    # the same dW twice in one scope: the second launch reduces at once (stream order keeps the two sums apart)
    ref2, _ = run(False, twice=True)
    got2, keep2 = run(True, twice=True)
    assert int(lib.sp_wgrad_reduce_pending()) == 3
    L.call("sp_wgrad_reduce_flush", 1, ops.stream())
    torch.cuda.synchronize()
    for (dw0, db0), (dw1, db1) in zip(ref2, got2):
        assert torch.allclose(dw0, dw1, rtol=1e-6, atol=1e-6 * float(dw0.abs().max())) and torch.allclose(db0, db1, rtol=1e-6, atol=1e-5)
    # a dropped queue
    got3, keep3 = run(True)
    assert int(lib.sp_wgrad_reduce_pending()) == 3
    L.call("sp_wgrad_reduce_flush", 0, ops.stream())
    assert int(lib.sp_wgrad_reduce_pending()) == 0
    torch.cuda.synchronize()
    assert all(float(dw.abs().max()) == 0.0 for dw, _ in got3)


POOL2_CASES = [(64, 64, 2, 16, 32, 0), (64, 64, 2, 8, 32, 0), (64, 128, 2, 16, 32, 0), (128, 128, 3, 8, 64, 0), (40, 256, 1, 16, 32, 0),
               (64, 128, 2, 16, 32, 2), (64, 64, 5, 64, 128, 0), (64, 128, 5, 64, 128, 2)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", POOL2_CASES)
def test_conv_pool2_epilogue(case, dtype):
    """2x2 average pooling fused into the 3x3 convolution's epilogue (sp_conv_params.pool2) with a residual at the pooled
    resolution, every kernel that implements it (halo<64>, tall<1>, halo<128>, tall<2>), forward and backward, against
    avg_pool2d(conv(x)) + res of the oracle."""
    cin, cout, n, h, w, tall = case
    ops.set_compute_dtype(dtype)
    m = models.SNConv2d(cin, cout, 3).cuda()
    sd = synth(m, 13, "c.")
    S = O.make_state(sd)
    x = q(rnd(n, cin, h, w, seed=1), dtype).requires_grad_(True)
    r1 = q(rnd(n, cout, h // 2, w // 2, seed=2), dtype).requires_grad_(True)
    gy = q(rnd(n, cout, h // 2, w // 2, seed=4), dtype)
    ref = torch.nn.functional.avg_pool2d(O.sn_conv(S, "c", x, True, 1), 2) + r1
    ref.backward(gy)
    xd, r1d = (dev(t, dtype).requires_grad_(True) for t in (x, r1))
    assert ops.conv_pool2_ok(h, w, cout, 3)
    if tall:
        ops.set_tuning(ops.TUNE_CONV_TALL, tall)
    try:
        y = m(xd, ops.ACT_NONE, r1d, None, pool2=True)
        y.backward(dev(gy, dtype))
    finally:
        ops.set_tuning(ops.TUNE_CONV_TALL, -1)
    assert tuple(y.shape) == (n, cout, h // 2, w // 2)
    tol = TOL[dtype]
    close(y, ref, tol, "y")
    close(xd.grad, x.grad, tol, "dx")
    close(r1d.grad, r1.grad, tol, "dres1")
    close(m.weight_orig.grad, S["c.weight_orig"].grad, 2 * tol, "dW")
    close(m.bias.grad, S["c.bias"].grad, 2 * tol, "db")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(64, 2, 16, 16), (12, 1, 6, 10), (256, 3, 8, 8)])
def test_act_avgpool2(case, dtype):
    """(lrelu(x), avgpool2(x)) in one pass and its single backward kernel, incl. a missing branch gradient."""
    c, n, h, w = case
    ops.set_compute_dtype(dtype)
    x = q(rnd(n, c, h, w, seed=1), dtype).requires_grad_(True)
    ga, gp = q(rnd(n, c, h, w, seed=2), dtype), q(rnd(n, c, h // 2, w // 2, seed=3), dtype)
    ra, rp = O.lrelu(x), F.avg_pool2d(x, 2)
    ((ra * ga).sum() + (rp * gp).sum()).backward()
    xd = dev(x, dtype).requires_grad_(True)
    ya, yp = ops.act_avgpool2(xd, ops.ACT_LRELU)
    torch.autograd.backward([ya, yp], [dev(ga, dtype), dev(gp, dtype)])
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    close(ya, ra, tol, "act", robust=False)
    close(yp, rp, tol, "pool", robust=False)
    close(xd.grad, x.grad, tol, "dx", robust=False)
    xd2 = dev(x, dtype).requires_grad_(True)
    _, yp2 = ops.act_avgpool2(xd2, ops.ACT_LRELU)
    yp2.backward(dev(gp, dtype))
    close(xd2.grad, F.interpolate(gp, scale_factor=2, mode="nearest") * 0.25, tol, "dx (pool branch only)", robust=False)


@pytest.mark.parametrize("dtype", DTYPES)
def test_batch_norm_fused_upsample(dtype):
    """CBN + LeakyReLU + bilinear x2 in one pass (sp_bn_apply_upsample2) against the two separate operators."""
    ops.set_compute_dtype(dtype)
    n, c, h, w = 3, 64, 8, 8
    x = q(rnd(n, c, h, w, seed=1), dtype)
    emb = torch.cat([1.0 + 0.1 * rnd(5, c, seed=2), 0.1 * rnd(5, c, seed=3)], dim=1).cuda()
    cls = torch.tensor([0, 3, 3]).cuda()
    gy = dev(q(rnd(n, c, 2 * h, 2 * w, seed=4), dtype), dtype)
    outs = []
    for fused in (False, True):
        xd = dev(x, dtype).requires_grad_(True)
        e = emb.clone().requires_grad_(True)
        rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
        y = ops.batch_norm(xd, None, None, e, cls, rm, rv, 0.001, 1e-5, True, ops.ACT_LRELU, fused)
        if not fused:
            y = ops.upsample2(y)
        y.backward(gy)
        outs.append((y.detach(), xd.grad.detach(), e.grad.detach()))
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for k, what in enumerate(("y", "dx", "demb")):
        close(outs[1][k], host(outs[0][k]), tol, what, robust=False)


def test_conv_pool2_rejects_unsupported_layers():
    ops.set_compute_dtype(torch.float32)
    m = models.SNConv2d(64, 64, 3).cuda()
    synth(m, 1)
    with pytest.raises(Exception):
        m(dev(rnd(1, 64, 4, 4, seed=1), torch.float32), pool2=True)


def test_conv_linearity_property_full_size():
    """Size-independent property at a BASELINE-sized layer (64->64 @256^2, B=2): conv(a*x1 + x2) = a*conv(x1) + conv(x2) - b."""
    ops.set_compute_dtype(torch.float32)
    m = models.SNConv2d(64, 64, 3).cuda().eval()       # eval: no power iteration, same weights for all three calls
    synth(m, 5)
    x1, x2 = rnd(2, 64, 256, 256, seed=1), rnd(2, 64, 256, 256, seed=2)
    with torch.no_grad():
        y1, y2 = m(dev(x1, torch.float32)), m(dev(x2, torch.float32))
        y3 = m(dev(0.5 * x1 + x2, torch.float32))
    b = m.bias.detach().float().cpu()[None, :, None, None]
    close(y3, 0.5 * host(y1) + host(y2) - 0.5 * b, 2e-5, "linearity")


# ----------------------------------------------------------------------------------------------
# linear
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(128, 128, 2), (365, 2048, 4), (4096, 365, 3), (768, 128, 20), (512, 130, 20), (1000, 128, 32), (1024, 256, 7),
                                  (136, 77, 5), (1536, 128, 20), (128, 16384, 20),
                                  # 33 .. 64 rows: the four-fragment form of the split-K MFMA kernel (round 5: both groups of the two-batch VGG pass)
                                  (4096, 2048, 40), (365, 130, 48), (128, 128, 64), (2048, 1000, 33)])
def test_sn_linear(case, dtype):
    """(16-bit storage: rows of up to 1024 values finish in one launch - the single K-split applies the epilogue itself -, longer
    rows go through slabs + the finalize pass: csrc/linear.hip)"""
    k, n, b = case
    ops.set_compute_dtype(dtype)
    m = models.SNLinear(k, n).cuda()
    sd = synth(m, 3, "l.")
    S = O.make_state(sd)
    x = q(rnd(b, k, seed=1), dtype).requires_grad_(True)
    r = q(rnd(b, n, seed=2), dtype).requires_grad_(True)
    gy = q(rnd(b, n, seed=3), dtype)
    use_act = dtype == torch.float32          # see test_conv_epilogue_act_and_residuals
    pre = O.sn_linear(S, "l", x, True) + r
    ref = O.lrelu(pre) if use_act else pre
    ref.backward(gy)
    xd, rd = dev(x, dtype).requires_grad_(True), dev(r, dtype).requires_grad_(True)
    y = m(xd, ops.ACT_LRELU if use_act else ops.ACT_NONE, rd)
    y.backward(dev(gy, dtype))
    tol = TOL[dtype]
    close(y, ref, tol, "y")
    close(xd.grad, x.grad, tol, "dx")
    close(rd.grad, r.grad, tol, "dres")
    close(m.weight_orig.grad, S["l.weight_orig"].grad, 2 * tol, "dW")
    close(m.bias.grad, S["l.bias"].grad, 2 * tol, "db")


# ----------------------------------------------------------------------------------------------
# conditional batch norm (+ fused LeakyReLU), plain affine BN
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 512, 4, 4), (4, 64, 16, 16), (3, 768, 2, 2)])
def test_conditional_batch_norm(case, dtype):
    n, c, h, w = case
    ops.set_compute_dtype(dtype)
    m = models.ConditionalBatchNorm(c).cuda()
    sd = synth(m, 5, "b.")
    S = O.make_state(sd)
    x = q(rnd(n, c, h, w, seed=1) * 1.5 + 0.3, dtype).requires_grad_(True)
    cls = torch.tensor([3, 100, 3, 364][:n])
    onehot = F.one_hot(cls, 365).float()
    gy = q(rnd(n, c, h, w, seed=2), dtype)
    ref = O.lrelu(O.conditional_batch_norm(S, "b", x, onehot, True))
    ref.backward(gy)
    xd = dev(x, dtype).requires_grad_(True)
    y = m(xd, onehot.cuda(), ops.ACT_LRELU)
    y.backward(dev(gy, dtype))
    tol = TOL[dtype]
    close(y, ref, tol, "y")
    close(xd.grad, x.grad, 2 * tol, "dx")
    close(m.embedding.weight.grad, S["b.embedding.weight"].grad, 2 * tol, "dembedding")
    close(m.batch_norm.running_mean, S["b.batch_norm.running_mean"], 1e-5 if dtype == torch.float32 else 1e-2, "running_mean")
    close(m.batch_norm.running_var, S["b.batch_norm.running_var"], 1e-5 if dtype == torch.float32 else 1e-2, "running_var")
    assert int(m.batch_norm.num_batches_tracked) == int(S["b.batch_norm.num_batches_tracked"]) == 1


@pytest.mark.parametrize("dtype", DTYPES)
def test_affine_batch_norm(dtype):
    ops.set_compute_dtype(dtype)
    bn = torch.nn.BatchNorm2d(64).cuda()
    S = {"b." + k: v.clone() for k, v in params.synth_state_dict(bn.state_dict(), 4).items()}
    bn.load_state_dict({k[2:]: v for k, v in S.items()})
    S = O.make_state(S)
    x = q(rnd(2, 64, 8, 8, seed=1), dtype).requires_grad_(True)
    gy = q(rnd(2, 64, 8, 8, seed=2), dtype)
    ref = O.lrelu(O.batch_norm(S, "b", x, True, 0.1, True))
    ref.backward(gy)
    xd = dev(x, dtype).requires_grad_(True)
    y = ops.batch_norm(xd, bn.weight, bn.bias, None, None, bn.running_mean, bn.running_var, 0.1, 1e-5, True, ops.ACT_LRELU)
    y.backward(dev(gy, dtype))
    tol = TOL[dtype]
    close(y, ref, tol, "y")
    close(xd.grad, x.grad, 2 * tol, "dx")
    close(bn.weight.grad, S["b.weight"].grad, 2 * tol, "dgamma")
    close(bn.bias.grad, S["b.bias"].grad, 2 * tol, "dbeta")
    close(bn.running_var, S["b.running_var"], 1e-5 if dtype == torch.float32 else 1e-2, "running_var")


# ----------------------------------------------------------------------------------------------
# resampling
# ----------------------------------------------------------------------------------------------
def _fwd_bwd(fn_dev, fn_ref, x, dtype, tol=None, gy_seed=9):
    x = q(x, dtype).requires_grad_(True)
    ref = fn_ref(x)
    gy = q(rnd(*ref.shape, seed=gy_seed), dtype)
    ref.backward(gy)
    xd = dev(x, dtype).requires_grad_(True)
    y = fn_dev(xd)
    y.backward(dev(gy, dtype))
    tol = TOL[dtype] if tol is None else tol
    close(y, ref, tol, "forward")
    close(xd.grad, x.grad, tol, "backward")


@pytest.mark.parametrize("dtype", DTYPES)
def test_upsample_bilinear_align_corners(dtype):
    _fwd_bwd(ops.upsample2, O.upsample2, rnd(2, 16, 5, 7, seed=1), dtype)
    _fwd_bwd(ops.upsample2, O.upsample2, rnd(1, 64, 16, 16, seed=2), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_avgpool_and_dual_output(dtype):
    _fwd_bwd(ops.avgpool2, lambda t: F.avg_pool2d(t, 2), rnd(2, 32, 8, 12, seed=1), dtype)
    x = q(rnd(2, 32, 8, 8, seed=3), dtype).requires_grad_(True)
    p = F.avg_pool2d(x, 2)
    (p * 2.0 + O.lrelu(p)).sum().backward()
    xd = dev(x, dtype).requires_grad_(True)
    y, ya = ops.avgpool2(xd, ops.ACT_LRELU)
    close(ya, O.lrelu(p), TOL[dtype], "lrelu(pooled)")
    (y.float() * 2.0 + ya.float()).sum().backward()
    close(xd.grad, x.grad, TOL[dtype], "dual backward")


def test_maxpool_routing_is_bit_exact_with_ties():
    ops.set_compute_dtype(torch.float32)
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, 3, (2, 8, 6, 6), generator=g).float()      # many ties inside windows
    x.requires_grad_(True)
    ref = F.max_pool2d(x, 2, 2)
    gy = rnd(*ref.shape, seed=1)
    ref.backward(gy)
    xd = dev(x, torch.float32).requires_grad_(True)
    y = ops.maxpool2(xd)
    y.backward(dev(gy, torch.float32))
    assert torch.equal(host(y), ref.detach())
    assert torch.equal(host(xd.grad), x.grad)


POOL_IDX_CASES = [  # (cin, cout, n, h, w, ping-pong kernel off): pp<1> | tall<1,16>, halo<64> (h % 16 != 0), pp<2> | tall<2,8>; fp32: tall / halo
    (64, 64, 2, 32, 32, 0), (64, 64, 2, 32, 32, 1), (64, 64, 2, 24, 32, 0), (128, 128, 2, 16, 32, 0), (128, 256, 3, 8, 64, 1), (64, 128, 2, 32, 64, 0),
    (16, 80, 1, 16, 32, 0), (16, 48, 1, 16, 32, 1)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ties", [False, True])
@pytest.mark.parametrize("case", POOL_IDX_CASES)
def test_conv_maxpool_epilogue_records_the_window_positions(case, ties, dtype):
    """conv3x3 -> ReLU -> MaxPool2d(2) with the pooling AND its routing in the convolution's epilogue (sp_conv_params.pool_idx, the
    frozen VGG-16's stages in the pass with gradient, /root/reference/models.py:183-216): pooled output and the gradient routed by
    sp_maxpool2_bwd_idx are BIT-IDENTICAL to the separate path (unpooled tensor -> sp_maxpool2_fwd -> sp_maxpool2_bwd with relu,
    itself pinned to torch's first-maximum routing by test_maxpool_routing_is_bit_exact_with_ties), in every kernel that pools.
    ties: small-integer operands, so that windows are full of exactly equal values and of all-negative windows."""
    from semantic_pyramid_for_image_generation_amd import _lib as L
    cin, cout, n, h, w, pp_off = case
    ops.set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(11)
    if ties:
        x = torch.randint(0, 2, (n, cin, h, w), generator=g).float()
        wt = torch.randint(-1, 2, (cout, cin, 3, 3), generator=g).float() * (torch.rand(cout, cin, 3, 3, generator=g) < 0.1).float()
        bias = torch.randint(-2, 2, (cout,), generator=g).float()
    else:
        x, wt, bias = rnd(n, cin, h, w, seed=1), rnd(cout, cin, 3, 3, seed=2) * 0.05, rnd(cout, seed=3)
    e = ops.chunk_elems(dtype)
    cin_p = ops.pad_to(cin, e)
    xd = dev(x, dtype)
    wd = wt.cuda().contiguous()
    fwd = torch.empty(cout * 9 * cin_p, dtype=dtype, device="cuda")
    dg = torch.empty(cin * 9 * ops.pad_to(cout, e), dtype=dtype, device="cuda")
    sd = ops.sp_dtype(dtype)
    L.call("sp_pack_weight", ops.ptr(wd), cout, cin * 9, cin, 9, cin_p, ops.pad_to(cout, e), 0, 0, ops.ptr(fwd), ops.ptr(dg), sd, ops.stream())
    bd = bias.cuda()
    if pp_off:
        ops.set_tuning(L.TUNE_KEYS["SP_CONV_PP"], 0)
    try:
        full = ops.nhwc_empty(n, cout, h, w, dtype, "cuda")
        ops.conv_launch(xd, fwd.data_ptr(), bd, full, None, None, None, 0.0, n, h, w, cin_p, cout, cout, 3, ops.ACT_RELU, dtype)
        pooled = ops.nhwc_empty(n, cout, h // 2, w // 2, dtype, "cuda")
        idx = torch.full((n * (h // 2) * (w // 2) * (cout // 16),), -1, dtype=torch.int32, device="cuda")
        ops.conv_launch(xd, fwd.data_ptr(), bd, pooled, None, None, None, 0.0, n, h, w, cin_p, cout, cout, 3, ops.ACT_RELU, dtype, pool2=2,
                        pool_idx=idx)
    finally:
        ops.set_tuning(L.TUNE_KEYS["SP_CONV_PP"], -1)
    ref_pooled = ops.nhwc_empty(n, cout, h // 2, w // 2, dtype, "cuda")
    L.call("sp_maxpool2_fwd", ops.ptr(full), ops.ptr(ref_pooled), n, h, w, cout, 0, sd, ops.stream())
    assert torch.equal(pooled, ref_pooled)
    gy = dev(rnd(n, cout, h // 2, w // 2, seed=5), dtype)
    ref_dx, dx = ops.nhwc_empty(n, cout, h, w, dtype, "cuda"), ops.nhwc_empty(n, cout, h, w, dtype, "cuda")
    L.call("sp_maxpool2_bwd", ops.ptr(gy), ops.ptr(full), ops.ptr(ref_dx), n, h, w, cout, 1, sd, ops.stream())
    L.call("sp_maxpool2_bwd_idx", ops.ptr(gy), ops.ptr(pooled), ops.ptr(idx), ops.ptr(dx), n, h, w, cout, sd, ops.stream())
    torch.cuda.synchronize()
    assert torch.equal(dx, ref_dx)
    if ties:
        assert float((ref_dx != 0).float().mean()) > 0.02           # (the case routes something)


@pytest.mark.parametrize("dtype", DTYPES)
def test_adaptive_avgpool(dtype):
    _fwd_bwd(lambda t: ops.adaptive_avgpool(t, 7, 7), lambda t: F.adaptive_avg_pool2d(t, (7, 7)), rnd(2, 16, 8, 8, seed=1), dtype)
    _fwd_bwd(lambda t: ops.adaptive_avgpool(t, 1, 1, ops.ACT_LRELU), lambda t: F.adaptive_avg_pool2d(O.lrelu(t), 1),
             rnd(3, 768, 2, 2, seed=2), dtype)


# ----------------------------------------------------------------------------------------------
# attention block (4 SN 1x1 convs + max-pool + fused softmax(QK^T)V + gamma residual)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("channels", [256, 64])
def test_self_attention_block(channels, dtype):
    ops.set_compute_dtype(dtype)
    m = models.SelfAttention(channels).cuda()
    bank = ops.SpectralNormBank(models._collect_sn(m))
    sd = synth(m, 9, "a.")
    S = O.make_state(sd)
    x = q(rnd(2, channels, 32, 32, seed=1, scale=0.5), dtype).requires_grad_(True)
    gy = q(rnd(2, channels, 32, 32, seed=2), dtype)
    ref = O.self_attention(S, "a", x, True)
    ref.backward(gy)
    xd = dev(x, dtype).requires_grad_(True)
    bank.begin(True, dtype, xd.device)
    y = m(xd)
    bank.end()
    y.backward(dev(gy, dtype))
    tol = TOL[dtype]
    close(y, ref, tol, "y")
    close(xd.grad, x.grad, 2 * tol, "dx")
    # dgamma is one scalar = sum of 1e5 signed products: compare against the magnitude of the terms, not of the sum
    gscale = float((gy.abs() * (ref.detach() - x.detach()).abs()).sum()) / max(abs(float(S["a.gamma"])), 1e-6)
    assert abs(float(m.gamma.grad) - float(S["a.gamma"].grad)) <= (1e-5 if dtype == torch.float32 else 2e-3) * gscale, "dgamma"
    for name in ("query_convolution", "key_convolution", "value_convolution", "attention_convolution"):
        close(getattr(m, name).weight_orig.grad, S["a.%s.weight_orig" % name].grad, 4 * tol, "dW " + name)


# ----------------------------------------------------------------------------------------------
# masks: bit-exact
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 32, 32, 16, 16, 32, 128), (2, 12, 16, 12, 8, 64, 64), (1, 16, 16, 8, 8, 32, 32),
                                  (2, 8, 8, 4, 4, 8, 32), (2, 32, 32, 16, 16, 64, 256)])
def test_attention_core_direct(case, dtype):
    """softmax(Q K^T) V and its three gradients on free-standing tensors: the MFMA kernels (bf16; d = 32 / 64, ragged
    query tail N = 192) and the VALU kernels (fp32, d = 8) against plain torch."""
    b, hq, wq, hk, wk, d, dv = case
    ops.set_compute_dtype(dtype)
    qq = q(rnd(b, d, hq, wq, seed=1, scale=0.7), dtype).requires_grad_(True)
    kk = q(rnd(b, d, hk, wk, seed=2, scale=0.7), dtype).requires_grad_(True)
    vv = q(rnd(b, dv, hk, wk, seed=3), dtype).requires_grad_(True)
    go = q(rnd(b, dv, hq, wq, seed=4), dtype)
    Q, K, V = (t.flatten(2).transpose(1, 2) for t in (qq, kk, vv))        # [b][positions][channels]
    P = torch.softmax(Q @ K.transpose(1, 2), dim=-1)
    ref = (P @ V).transpose(1, 2).reshape(b, dv, hq, wq)
    ref.backward(go)
    qd, kd, vd = (dev(t, dtype).requires_grad_(True) for t in (qq, kk, vv))
    o = ops.attention_core(qd, kd, vd)
    o.backward(dev(go, dtype))
    tol = TOL[dtype]
    close(o, ref, tol, "o")
    close(qd.grad, qq.grad, tol, "dq")
    close(kd.grad, kk.grad, tol, "dk")
    close(vd.grad, vv.grad, tol, "dv")


def test_mask_concat_and_mul_bit_exact():
    ops.set_compute_dtype(torch.float32)
    feat = rnd(2, 64, 8, 8, seed=1)
    mask = (rnd(2, 1, 8, 8, seed=2) > 0).float()
    out = host(ops.mask_concat(feat.cuda(), mask.cuda()))
    ref = torch.cat((feat * mask, mask), dim=1)
    assert out.shape[1] == 68
    assert torch.equal(out[:, :65], ref)
    assert torch.equal(out[:, 65:], torch.zeros(2, 3, 8, 8))
    f2, m2 = rnd(3, 365, seed=3), (rnd(3, 365, seed=4) > 0).float()
    assert torch.equal(host(ops.mask_mul_2d(f2.cuda(), m2.cuda())), f2 * m2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_ingest_image_and_backward(dtype):
    ops.set_compute_dtype(dtype)
    img = torch.rand(2, 3, 16, 16) * 2 - 1
    scale, shift = (2.0, 3.0, 4.0), (0.1, -0.2, 0.3)
    x = img.cuda().requires_grad_(True)
    y = ops.ingest_image(x, dtype, scale, shift)
    cp = ops.pad_channels(3, dtype)
    ref = img * torch.tensor(scale)[None, :, None, None] + torch.tensor(shift)[None, :, None, None]
    close(y[:, :3], q(ref, dtype), 1e-6 if dtype == torch.float32 else 1e-2, "ingest")
    assert float(host(y)[:, 3:].abs().max()) == 0.0 and y.shape[1] == cp
    gy = rnd(2, cp, 16, 16, seed=1)
    y.backward(dev(gy, dtype))
    close(x.grad, q(gy[:, :3], dtype) * torch.tensor(scale)[None, :, None, None], TOL[dtype], "ingest backward")


# ----------------------------------------------------------------------------------------------
# discriminator head + losses
# ----------------------------------------------------------------------------------------------
def test_direct_grads_window_matches_autograd_accumulation():
    """SpectralNormBank.direct_grads: two forward passes (D(real), D(fake), model_wrapper.py:150-160) and one backward -
    weight_orig / bias gradients accumulated by the kernels into the bank's flat buffers equal the sums autograd forms;
    a third pass without zero_grad() keeps accumulating, zero_grad() opens a new window."""
    ops.set_compute_dtype(torch.float32)
    D = models.Discriminator(channel_factor=8).cuda()
    sd0 = {k: v.clone() for k, v in params.synth_state_dict(D.state_dict(), 3).items()}
    x1, x2 = rnd(2, 3, 256, 256, seed=1).cuda(), rnd(2, 3, 256, 256, seed=2).cuda()
    cls = torch.tensor([3, 200]).cuda()

    def run(direct, passes):
        D.load_state_dict(sd0)                       # same weights AND power-iteration vectors
        D._bank.direct_grads = direct
        D.zero_grad()
        out = []
        for _ in range(passes):
            loss = ops.sqerr_loss(D(x1, cls), 1.0) + ops.sqerr_loss(D(x2, cls), 0.0)
            loss.backward()
            out.append({n: p.grad.detach().clone() for n, p in D.named_parameters() if p.grad is not None})
        return out

    try:
        ref = run(False, 2)
        got = run(True, 2)
        assert D.layers[1].main_block[1].weight_orig.grad.data_ptr() == D._bank.w_views[D.layers[1].main_block[1]._sn_slot].data_ptr()
        got_new_window = run(True, 1)
    finally:
        D._bank.direct_grads = False
    for k in (0, 1):
        assert set(ref[k]) == set(got[k])
        for n in ref[k]:
            close(got[k][n], host(ref[k][n]), 2e-5, "pass %d %s" % (k, n), robust=False)
    for n in ref[0]:
        close(got_new_window[0][n], host(ref[0][n]), 2e-5, "after zero_grad %s" % n, robust=False)


def test_non_direct_bank_bias_gradients_with_deferred_reductions_bf16():
    """Round-5 ADVICE (medium): with the slab reductions of the streaming 1x1 / 8-channel weight-gradient launches deferred to the bank's
    backward node, a bank WITHOUT direct gradients handed its bias slot back to autograd before the queued reduction had run - autograd
    then summed the D(real) and D(fake) contributions of an unreduced slot.  16-bit storage (the split path: 256 x 256 maps, thousands
    of pixels per split), two passes, one backward: every bias and weight gradient of the autograd-accumulated run equals the
    direct-gradient run (same kernels, same order of the partial sums up to the final addition)."""
    ops.set_compute_dtype(torch.bfloat16)
    try:
        D = models.Discriminator(channel_factor=8).cuda()
        sd0 = {k: v.clone() for k, v in params.synth_state_dict(D.state_dict(), 3).items()}
        x1, x2 = rnd(4, 3, 256, 256, seed=1).cuda(), rnd(4, 3, 256, 256, seed=2).cuda()
        cls = torch.tensor([3, 200, 7, 11]).cuda()

        def run(direct):
            D.load_state_dict(sd0)
            D._bank.direct_grads = direct
            D.zero_grad()
            loss = ops.sqerr_loss(D(x1, cls), 1.0) + ops.sqerr_loss(D(x2, cls), 0.0)
            loss.backward()
            return {n: p.grad.detach().float().clone() for n, p in D.named_parameters() if p.grad is not None}
        try:
            ref = run(True)
            got = run(False)
        finally:
            D._bank.direct_grads = False
        assert set(ref) == set(got)
        for n in ref:
            scale = max(float(ref[n].abs().max()), 1e-12)
            err = float((got[n] - ref[n]).abs().max()) / scale
            assert err <= 2e-3, (n, err)           # (before the fix: the first layers' biases were off by whole partial sums)
    finally:
        ops.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dtype", DTYPES)
def test_discriminator_head_and_lsgan(dtype):
    ops.set_compute_dtype(dtype)
    D = models.Discriminator(channel_factor=8).cuda()
    sd = params.synth_state_dict(D.state_dict(), 3)
    D.load_state_dict(sd)
    S = O.make_state(sd)
    b = 4
    x = q(rnd(b, 128, seed=1), dtype).requires_grad_(True)
    cls = torch.tensor([5, 17, 5, 300])
    emb_w = O.sn_weight(S, "embedding", True)
    pred_ref = O.sn_linear(S, "classification", x, True) + x * emb_w[cls[:, None]]
    assert pred_ref.shape == (b, b, 128)
    loss_ref = O.lsgan_generator_loss(pred_ref)
    loss_ref.backward()
    bank = D._bank
    xd = dev(x, dtype).requires_grad_(True)
    bank.begin(True, dtype, xd.device)
    pred = ops.discriminator_head(xd, D.embedding, D.classification, cls.cuda())
    bank.end()
    loss = ops.sqerr_loss(pred, 1.0)
    loss.backward()
    tol = TOL[dtype]
    close(pred, pred_ref, tol, "pred")
    close(loss, loss_ref, tol, "loss")
    close(xd.grad, x.grad, 2 * tol, "dx")
    close(D.embedding.weight_orig.grad, S["embedding.weight_orig"].grad, 2 * tol, "dE")
    close(D.classification.weight_orig.grad, S["classification.weight_orig"].grad, 2 * tol, "dwc")
    close(D.classification.bias.grad, S["classification.bias"].grad, 2 * tol, "dbc")


@pytest.mark.parametrize("n", [1, 7, 4096, 51200, 51203, (1 << 18) + 5, 1 << 20])
def test_sqerr_loss_sizes(n):
    """0.5 * mean((p - t)^2) (lossfunction.py:137,164): the one-block form (up to 2^18 elements; vector body + scalar tail, also from
    an address that is not 16-byte aligned) and the multi-block form agree with torch in fp64."""
    ops.set_compute_dtype(torch.float32)
    base = torch.randn(n + 1, device="cuda", generator=torch.Generator(device="cuda").manual_seed(n))
    for off in (0, 1):
        p = base[off:off + n].clone() if off == 0 else base[1:1 + n]
        p = p.detach().requires_grad_(True)
        loss = ops.sqerr_loss(p, 1.0)
        loss.backward(torch.tensor(2.0, device="cuda"))
        want = 0.5 * ((p.detach().double() - 1.0) ** 2).mean()
        assert abs(float(loss) - float(want)) <= 2e-6 * max(1.0, float(want))
        close(p.grad, (2.0 * (p.detach() - 1.0) / n).cpu(), 1e-6, "dp")


@pytest.mark.parametrize("dtype", DTYPES)
def test_reconstruction_and_diversity_losses(dtype):
    ops.set_compute_dtype(dtype)
    shapes = [(2, 8, 16, 16), (2, 16, 8, 8), (2, 4096), (2, 365)]
    real = [q(rnd(*s, seed=i), dtype) for i, s in enumerate(shapes)]
    fake = [q(rnd(*s, seed=10 + i), dtype).requires_grad_(True) for i, s in enumerate(shapes)]
    masks = [(rnd(2, 1, 16, 16, seed=20) > 0).float(), torch.ones(2, 1, 8, 8), (rnd(2, 4096, seed=21) > 0).float(),
             torch.zeros(2, 365)]
    ref = O.semantic_reconstruction_loss(real, fake, masks)
    ref.backward()
    fd = [dev(f, dtype).requires_grad_(True) for f in fake]
    loss = ops.semantic_reconstruction_loss([dev(r, dtype) for r in real], fd, [m.cuda() for m in masks])
    assert loss.shape == (1,)
    loss.backward()
    close(loss, ref, 1e-5 if dtype == torch.float32 else 1e-3, "rec loss")
    for a, b_ in zip(fd, fake):
        close(a.grad, b_.grad, 1e-6 if dtype == torch.float32 else 1e-2, "rec grad")
    img = q(torch.tanh(rnd(4, 3, 32, 32, seed=5)), dtype).requires_grad_(True)
    z = rnd(4, 128, seed=6)
    dref = O.diversity_loss(img, z)
    dref.backward()
    imgd = dev(img, dtype).requires_grad_(True)
    dl = ops.diversity_loss(imgd, z.cuda())
    dl.backward()
    close(dl, dref, 1e-5 if dtype == torch.float32 else 1e-3, "div loss")
    close(imgd.grad, img.grad, 1e-5 if dtype == torch.float32 else 2e-2, "div grad")
    # the caller's weight folded into the kernels (ModelWrapper: w_rec, w_div), an upstream gradient != 1, and repeated calls: the
    # shared fp64 accumulators of the one-launch forms must be back at zero after every call
    for _ in range(2):
        fd2 = [dev(f, dtype).requires_grad_(True) for f in fake]
        lw = ops.semantic_reconstruction_loss([dev(r, dtype) for r in real], fd2, [m.cuda() for m in masks], weight=0.1)
        lw.backward(gradient=torch.full((1,), 3.0, device="cuda"))
        close(lw, 0.1 * ref.detach(), 1e-5 if dtype == torch.float32 else 1e-3, "weighted rec loss")
        for a_, b_ in zip(fd2, fake):
            close(a_.grad, 0.3 * b_.grad, 1e-6 if dtype == torch.float32 else 1e-2, "weighted rec grad")
        img2 = dev(img, dtype).requires_grad_(True)
        dw = ops.diversity_loss(img2, z.cuda(), weight=0.25)
        dw.backward(gradient=torch.tensor(2.0, device="cuda"))
        close(dw, 0.25 * dref.detach(), 1e-5 if dtype == torch.float32 else 1e-3, "weighted div loss")
        close(img2.grad, 0.5 * img.grad, 1e-5 if dtype == torch.float32 else 2e-2, "weighted div grad")
    assert float(ops._loss_acc(torch.device("cuda", torch.cuda.current_device())).abs().max()) == 0.0
    # only some levels need a gradient
    fd3 = [dev(f, dtype).requires_grad_(i % 2 == 0) for i, f in enumerate(fake)]
    ops.semantic_reconstruction_loss([dev(r, dtype) for r in real], fd3, [m.cuda() for m in masks]).backward()
    for i, (a_, b_) in enumerate(zip(fd3, fake)):
        if i % 2 == 0:
            close(a_.grad, b_.grad, 1e-6 if dtype == torch.float32 else 1e-2, "rec grad (subset)")
        else:
            assert a_.grad is None


# ----------------------------------------------------------------------------------------------
# VGG-16 pyramid: forward taps and the input-gradient chain
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
def test_vgg16_pyramid(dtype):
    ops.set_compute_dtype(dtype)
    V = models.VGG16().cuda().eval()
    sd = params.synth_state_dict(V.state_dict(), 2)
    V.load_state_dict(sd)
    S = O.make_state(sd, frozen=True)
    img = (torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(1)) * 2 - 1).requires_grad_(True)
    ref = O.vgg16_forward(S, img)
    gs = [rnd(*r.shape, seed=30 + i) for i, r in enumerate(ref)]
    sum((r * g).sum() for r, g in zip(ref, gs)).backward()
    x = img.detach().cuda().requires_grad_(True)
    feats = V(x)
    assert [tuple(f.shape) for f in feats] == [tuple(r.shape) for r in ref]
    tol = 5e-4 if dtype == torch.float32 else 6e-2
    for i, (f, r) in enumerate(zip(feats, ref)):
        close(f, r, tol, "feature %d" % i)
    sum((f.float() * dev(g, torch.float32)).sum() for f, g in zip(feats, gs)).backward()
    # 13 ReLU masks + 5 max-pool routings sit between the taps and the image: isolated decisions flip with summation order
    close(x.grad, img.grad, 6e-3 if dtype == torch.float32 else 0.15, "d image", robust=True)
    # the no-gradient pass fuses ReLU + MaxPool into the last convolution of a stage (pool2 = 2): bit-identical taps
    with torch.no_grad():
        feats_ng = V(x.detach())
    for i, (f, g) in enumerate(zip(feats, feats_ng)):
        if i < 4:       # stages 1-4 (256^2 .. 32^2) take the fused epilogue
            assert torch.equal(f.detach(), g), "fused max-pool epilogue changed feature %d" % i
        else:           # 16^2 stage at batch 1: split-K partial sums meet through fp32 atomics, order varies run to run
            close(g, host(f), 1e-5 if dtype == torch.float32 else 2e-2, "feature %d (no-grad pass)" % i, robust=False)


# ----------------------------------------------------------------------------------------------
# multi-tensor Adam (optim.py / sp_adam_multi) against torch.optim.Adam
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_multi_tensor_adam_matches_torch(wd):
    from semantic_pyramid_for_image_generation_amd import optim
    shapes = [(3,), (1,), (129, 7), (70000,), (300, 300), (65536,), (65537,), (16, 3, 3, 3)]
    g = torch.Generator().manual_seed(3)
    base = [torch.randn(*s, generator=g) for s in shapes]
    pa = [torch.nn.Parameter(b.clone().cuda()) for b in base]
    pb = [torch.nn.Parameter(b.clone().cuda()) for b in base]
    oa = torch.optim.Adam(pa, lr=1e-2, weight_decay=wd, foreach=False)
    ob = optim.Adam(pb, lr=1e-2, weight_decay=wd)
    flat = torch.empty(sum(b.numel() for b in base) + 1, device="cuda")        # gradients as misaligned views of one buffer
    for it in range(4):
        off = 1
        for i, (a, b) in enumerate(zip(pa, pb)):
            gr = torch.randn(a.shape, generator=g).cuda() * (0.1 + it)
            if it == 2 and i == 1:
                a.grad = b.grad = None                                          # a parameter that skips a step keeps its own count
                continue
            a.grad = gr.clone()
            view = flat[off:off + gr.numel()].view(gr.shape)
            view.copy_(gr)
            b.grad = view
            off += gr.numel()
        oa.step()
        ob.step()
    for a, b, s in zip(pa, pb, shapes):
        # same formula; the kernel may contract multiply-adds (1-2 ulp of the parameter per step)
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-6), (s, float((a - b).abs().max()))
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["state"].keys() == sb["state"].keys()
    for k in sa["state"]:
        assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"])
        assert torch.allclose(sa["state"][k]["exp_avg_sq"], sb["state"][k]["exp_avg_sq"], rtol=1e-5, atol=1e-12)
    oa.load_state_dict(sb)                                                      # checkpoint compatibility both ways
    ob.load_state_dict(sa)
