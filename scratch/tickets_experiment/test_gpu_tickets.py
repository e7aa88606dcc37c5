"""Ticket scratch (include/sempyr.h: sp_set_ticket_scratch): the entry points that fold their finalize / reduce pass into the
producing launch must give the result of the two-pass form (SP_TUNE_TICKETS = 0) - bit for bit where both sum in the same order -
on repeated launches (the counters go back to zero), and leave the registered scratch zeroed."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

from semantic_pyramid_for_image_generation_amd import _lib as L  # noqa: E402
from semantic_pyramid_for_image_generation_amd import ops  # noqa: E402


def _scratch():
    ops.stream()                                         # registers the current stream's scratch on first use
    t = ops._TICKET_SCRATCH[torch.cuda.current_stream().cuda_stream]
    assert t is not None, "ticket scratch is off (SP_TICKETS=0?)"
    return t


class two_pass:
    def __enter__(self):
        ops.set_tuning(ops.TUNE_TICKETS, 0)

    def __exit__(self, *exc):
        ops.set_tuning(ops.TUNE_TICKETS, -1)


def both(fn):
    """fn() under the two-pass forms, then twice with tickets; returns (reference, folded_first, folded_second)."""
    with two_pass():
        ref = fn()
    a = fn()
    b = fn()
    torch.cuda.synchronize()
    assert int(_scratch().abs().max()) == 0, "a launch left a ticket counter non-zero"
    return ref, a, b


@pytest.mark.parametrize("shape", [(20, 25088, 4096), (20, 768, 128), (20, 4096, 2048), (7, 136, 130), (32, 128, 16384), (20, 128, 20)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_linear_split_k_finalize_folded(shape, dt):
    """sp_linear_fwd_ws (reference: every nn.Linear of models.py - VGG classifier, latent / feature mappings, D's 768 -> 128): the
    last K-split of a 128-column block sums the slabs in split order and applies bias / residual / activation."""
    b, k, n = shape
    ops.set_compute_dtype(dt)
    try:
        torch.manual_seed(k + n)
        kp = ops.pad_to(k, 8)
        x = torch.randn(b, k, device="cuda").to(dt)
        w = torch.zeros(n, kp, device="cuda", dtype=dt)
        w[:, :k] = (torch.randn(n, k, device="cuda") * k ** -0.5).to(dt)
        bias = torch.randn(n, device="cuda")
        res = torch.randn(b, n, device="cuda").to(dt)

        def run():
            y = torch.full((b, n), 7.0, device="cuda", dtype=dt)
            ops.linear_launch(x, w.data_ptr(), kp, bias, res, y, b, k, n, ops.ACT_LRELU)
            return y
        ref, a, c = both(run)
        assert torch.equal(ref, a) and torch.equal(ref, c)
        want = torch.nn.functional.leaky_relu(x.float() @ w[:, :k].float().t() + bias + res.float(), 0.2)
        assert float((ref.float() - want).abs().max()) <= 0.05 * float(want.abs().max())
    finally:
        ops.set_compute_dtype(torch.float32)


def test_scratch_is_per_stream_and_unregistered_streams_run_two_pass():
    """A stream without scratch (here: scratch withdrawn) must still compute - through the two-pass forms."""
    t = _scratch()
    h = torch.cuda.current_stream().cuda_stream
    x = torch.randn(20, 768, device="cuda").bfloat16()
    w = (torch.randn(128, 768, device="cuda") * 0.03).bfloat16()
    ops.set_compute_dtype(torch.bfloat16)
    try:
        y1 = torch.empty(20, 128, device="cuda", dtype=torch.bfloat16)
        ops.linear_launch(x, w.data_ptr(), 768, None, None, y1, 20, 768, 128, ops.ACT_NONE)
        L.call("sp_set_ticket_scratch", None, 0, ctypes.c_void_p(h))
        y2 = torch.empty(20, 128, device="cuda", dtype=torch.bfloat16)
        ops.linear_launch(x, w.data_ptr(), 768, None, None, y2, 20, 768, 128, ops.ACT_NONE)
        torch.cuda.synchronize()
        assert torch.equal(y1, y2)
    finally:
        L.call("sp_set_ticket_scratch", ops.ptr(t), t.numel(), ctypes.c_void_p(h))
        ops.set_compute_dtype(torch.float32)
    with pytest.raises(L.SempyrError):
        L.call("sp_set_ticket_scratch", ctypes.c_void_p(t.data_ptr() + 1), 4, ctypes.c_void_p(h))
