"""Multi-tensor Adam on the HIP path.

`Adam` IS a torch.optim.Adam (same constructor, param_groups, per-parameter state {step, exp_avg, exp_avg_sq} and
state_dict, so checkpoints written by the reference - model_wrapper.py:215-223 - load unchanged); only `.step()` is
replaced: every parameter of a group is updated by ONE launch of sp_adam_multi instead of torch's foreach kernels.
There is no fallback: parameters must be fp32 on the GPU, amsgrad / maximize / capturable are rejected.

Host cost: the chunk table (pointers, lengths) of a group is built ONCE per set of (parameter, gradient, moment) addresses and
cached; a step only refreshes the two bias-correction columns with vectorised numpy and uploads the table through pinned
memory.  The per-parameter ``state['step']`` tensors torch keeps are brought up to date lazily (state_dict(), or whenever the
cached plan is dropped) - round 1 incremented 286 CPU tensors and rebuilt the table every call (~5 ms of Python per step).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L
from .ops import ptr, stream

CHUNK = 65536
_DT = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i4"), ("step_size", "<f4"),
                ("inv_sqrt_bc2", "<f4"), ("reserved", "<f4")])


class _Plan:
    __slots__ = ("key", "params", "keep", "table", "idx", "steps", "total", "synced")


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        if amsgrad or kw.get("maximize") or kw.get("capturable") or kw.get("differentiable"):
            raise L.SempyrError("sempyr Adam supports the reference's configuration only (no amsgrad / maximize / capturable)")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, foreach=False, fused=False)
        self._ring = {}            # per group: two pinned host tables + the event of their last upload
        self._plans = {}           # per group: cached chunk table

    # ------------------------------------------------------------------------------------------ step bookkeeping
    def _sync_steps(self) -> None:
        """Writes the step counts the cached plans hold into torch's per-parameter ``state['step']`` tensors."""
        for plan in self._plans.values():
            if plan is not None and not plan.synced:
                for p, t in zip(plan.params, plan.steps):
                    self.state[p]["step"].fill_(float(t))
                plan.synced = True

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        self._plans = {}
        return super().load_state_dict(state_dict)

    def _build_plan(self, gi, group, key) -> _Plan:
        self._sync_steps()
        ps, gs, ms, vs, ns, steps, plist, keep = [], [], [], [], [], [], [], []
        for p in group["params"]:
            g = p.grad
            if g is None:
                continue
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise L.SempyrError("sempyr Adam needs contiguous fp32 GPU parameters (got %s %s)" % (p.device, p.dtype))
            if g.is_sparse:
                raise L.SempyrError("sempyr Adam does not support sparse gradients")
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            plist.append(p)
            ps.append(p.data_ptr()); gs.append(g.data_ptr()); ms.append(st["exp_avg"].data_ptr()); vs.append(st["exp_avg_sq"].data_ptr())
            ns.append(p.numel())
            steps.append(float(st["step"]))
        plan = _Plan()
        plan.key, plan.params, plan.keep = key, plist, keep
        plan.steps = np.asarray(steps, dtype=np.float64)
        plan.synced = True
        if not ps:
            plan.total, plan.table, plan.idx = 0, None, None
            return plan
        n = np.asarray(ns, dtype=np.int64)
        reps = (n + CHUNK - 1) // CHUNK
        total = int(reps.sum())
        idx = np.repeat(np.arange(len(ns)), reps)                          # tensor of each chunk
        first = np.cumsum(reps) - reps
        off = (np.arange(total) - first[idx]) * CHUNK                      # element offset of each chunk in its tensor
        tab = np.zeros(total, dtype=_DT)
        byte_off = (off * 4).astype(np.uint64)
        tab["p"] = np.asarray(ps, dtype=np.uint64)[idx] + byte_off
        tab["g"] = np.asarray(gs, dtype=np.uint64)[idx] + byte_off
        tab["m"] = np.asarray(ms, dtype=np.uint64)[idx] + byte_off
        tab["v"] = np.asarray(vs, dtype=np.uint64)[idx] + byte_off
        tab["n"] = np.minimum(n[idx] - off, CHUNK).astype(np.int32)
        plan.total, plan.table, plan.idx = total, tab, idx
        return plan

    @torch.no_grad()
    def step(self, closure=None, found_inf=None):
        """found_inf (optional, the fp16 mode's ops.LossScaler): a device float; the launch updates NOTHING when it is non-zero
        (sp_adam_multi_guarded) - the step is skipped on the device, without a host sync.  (The host-side step count that feeds the
        bias corrections still advances on a skipped step; torch's fused Adam takes it back.)"""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("maximize"):
                raise L.SempyrError("sempyr Adam: amsgrad / maximize are not supported")
            lr, (b1, b2), eps, wd = float(group["lr"]), group["betas"], float(group["eps"]), float(group["weight_decay"])
            # the plan stays valid while every parameter keeps its gradient (and moment) storage: true for the flat gradient
            # buffers of the training step and for captured graphs; fp32 / contiguity are checked when a plan is built
            key = tuple((p.data_ptr(), g.data_ptr(), g.dtype == torch.float32 and g.is_contiguous())
                        for p in group["params"] for g in (p.grad,) if g is not None)
            plan = self._plans.get(gi)
            if plan is None or plan.key != key:
                if any(not k[2] for k in key):
                    raise L.SempyrError("sempyr Adam needs contiguous fp32 gradients")
                plan = self._plans[gi] = self._build_plan(gi, group, key)
            if plan.total == 0:
                continue
            plan.steps += 1.0
            plan.synced = False
            tab = plan.table
            tab["step_size"] = (lr / (1.0 - b1 ** plan.steps)).astype(np.float32)[plan.idx]
            tab["inv_sqrt_bc2"] = (1.0 / np.sqrt(1.0 - b2 ** plan.steps)).astype(np.float32)[plan.idx]
            dev = plan.params[0].device
            with torch.cuda.device(dev):
                # the chunk table goes up through pinned memory (a pageable copy would make the host wait for the stream);
                # two tables alternate and each waits for its own previous upload before it is overwritten
                ring = self._ring.setdefault(gi, {"i": 0, "slots": [None, None]})
                ring["i"] ^= 1
                slot = ring["slots"][ring["i"]]
                nbytes = tab.nbytes
                if slot is None or slot[0].numel() < nbytes:
                    slot = [torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, pin_memory=True), None]
                    ring["slots"][ring["i"]] = slot
                if slot[1] is not None:
                    slot[1].synchronize()
                slot[0][:nbytes].numpy()[:] = tab.view(np.uint8)
                tab_dev = slot[0][:nbytes].to(dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                slot[1] = ev
                if found_inf is not None:
                    L.call("sp_adam_multi_guarded", ptr(tab_dev), plan.total, float(b1), float(b2), eps, wd, found_inf, stream())
                else:
                    L.call("sp_adam_multi", ptr(tab_dev), plan.total, float(b1), float(b2), eps, wd, stream())
        return loss
