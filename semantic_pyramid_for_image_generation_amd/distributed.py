"""One-process-per-GPU data parallelism for the training step (replaces nn.DataParallel, main.py:91-94).

Every rank holds the full G, D and frozen VGG and a shard of the batch.  The gradients of a network live in ONE contiguous
fp32 buffer (ops.SpectralNormBank.flat: the weight-gradient kernels and the batched spectral-norm backward write it, the
parameters' .grad are views of it), so the reducer all-reduces ranges of that buffer IN PLACE - no flatten / unflatten copies -
on a side stream (RCCL over xGMI on the GPU node; gloo in the CPU tests):

  * eager launches: the bank's layer groups report as they finish inside the backward pass (`on_group_done`), and each group's
    range goes to the side stream right then - the reduction of the late layers overlaps the backward of the early ones;
  * captured graphs (bench.py default): the ranges are enqueued bucket by bucket as soon as the graph that holds the backward
    has been launched;
  * in both modes ModelWrapper joins the discriminator's reduction only after the generator forward of the G phase (which does
    not read D), i.e. D's all-reduce hides under compute; G's reduction is joined before Adam(G).

Spectral-norm u/v evolve identically on all ranks (same weights -> same power iteration), BatchNorm statistics stay rank-local
like DataParallel replicas.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


class GradientReducer:
    def __init__(self, bucket_bytes: int = 32 << 20, process_group=None, single_rank_passthrough: bool = True) -> None:
        """single_rank_passthrough=False keeps the whole machinery (side stream, events, collectives, scaling) running in a
        group of ONE rank - the all-reduce is then the identity: how the single-GPU tests exercise the multi-GPU code path."""
        self.bucket_bytes = bucket_bytes
        self.group = process_group
        self.passthrough = single_rank_passthrough
        self._side = None
        self._pending = 0
        self.log: List[Tuple[int, int]] = []       # (start, stop) of every range reduced since the last join (tests)
        # timing (bench.py, N > 1): events around every collective on the side stream and around every join on the main stream
        self.timing = False
        self._marks: list = []                     # (first event, last event) pairs of the collectives since the last join
        self._joins: list = []                     # (tag, marks, event before the join's wait, event after it)

    def world_size(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def active(self) -> bool:
        """True if reductions are actually issued (more than one rank, or a one-rank group kept live for tests)."""
        return dist.is_available() and dist.is_initialized() and (self.world_size() > 1 or not self.passthrough)

    # -------------------------------------------------------------------------------------------- flat, in place
    def reduce_range(self, flat: torch.Tensor, start: int, stop: int) -> None:
        """Average flat[start:stop] over all ranks, in place, asynchronously: the range must be final on the CURRENT stream at
        the time of the call (GPU: an event orders the side stream behind it).  join() makes the results visible."""
        ws = self.world_size()
        if not self.active() or stop <= start:
            return
        view = flat[start:stop]
        self.log.append((start, stop))
        if not flat.is_cuda:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
            view.div_(ws)
            return
        if self._side is None:
            self._side = torch.cuda.Stream()
        ev = torch.cuda.Event()
        ev.record()                                   # the producers of this range, enqueued so far on the current stream
        self._side.wait_event(ev)
        with torch.cuda.stream(self._side):
            if self.timing:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            work.wait()                               # side stream waits for the collective; the host does not
            view.mul_(1.0 / ws)
            if self.timing:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                self._marks.append((e0, e1))
        self._pending += 1

    def reduce_flat(self, flat: torch.Tensor, ranges: Sequence[Tuple[int, int]]) -> None:
        for a, b in ranges:
            self.reduce_range(flat, a, b)

    def join(self, tag: str = "") -> None:
        """The current stream waits for every reduction enqueued so far."""
        if self._pending:
            if self.timing:
                m0 = torch.cuda.Event(enable_timing=True)
                m0.record()
            torch.cuda.current_stream().wait_stream(self._side)      # `flat` is persistent: no allocator hand-over to record
            if self.timing:
                m1 = torch.cuda.Event(enable_timing=True)
                m1.record()
                self._joins.append((tag, self._marks, m0, m1))
                self._marks = []
            self._pending = 0
        self.log = []

    def stats(self) -> dict:
        """Means over the joins recorded while `timing` was on, per tag: `allreduce_ms` = first collective's start to the last
        one's end on the side stream (scaling included), `exposed_ms` = how long the main stream actually waited at the join
        (what the overlap did not hide).  Synchronises."""
        torch.cuda.synchronize()
        acc = {}
        for tag, marks, m0, m1 in self._joins:
            a = acc.setdefault(tag, [0.0, 0.0, 0])
            if marks:
                a[0] += marks[0][0].elapsed_time(marks[-1][1])
            a[1] += m0.elapsed_time(m1)
            a[2] += 1
        self._joins = []
        return {tag: {"allreduce_ms": round(v[0] / v[2], 4), "exposed_ms": round(v[1] / v[2], 4), "joins": v[2]} for tag, v in acc.items() if v[2]}

    # -------------------------------------------------------------------------------------------- per-parameter (fallback)
    def reduce(self, params: Sequence[torch.Tensor]) -> None:
        """Average ``p.grad`` of loose parameters (gradients that do not live in a bank's flat buffer: a network without
        direct gradients, or a caller's own modules): bucketed flatten -> all-reduce -> scatter back, joined at once."""
        ws = self.world_size()
        if not self.active():
            return
        buckets, cur, size = [], [], 0
        for p in reversed(list(params)):          # reverse registration order ~ order in which backward finishes them
            if p.grad is None:
                continue
            cur.append(p)
            size += p.grad.numel() * p.grad.element_size()
            if size >= self.bucket_bytes:
                buckets.append(cur)
                cur, size = [], 0
        if cur:
            buckets.append(cur)
        for bucket in buckets:
            grads = [p.grad for p in bucket]
            flat = torch._utils._flatten_dense_tensors(grads)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(ws)
            for g, r in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
                g.copy_(r)


def flat_of(module) -> Optional[torch.Tensor]:
    bank = getattr(module, "_bank", None)
    return bank.flat if bank is not None and bank.direct_grads else None
