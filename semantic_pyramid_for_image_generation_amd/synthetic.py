"""Synthetic batches with the reference's input contract (data.py:39-90, misc.py:13-97).

images  (B,3,256,256) float32 in [-1,1]          (data.py:53)
labels  (B,365)       int64 one-hot               (data.py:58-59)
masks   list of 7 float32 0/1 tensors, index i <-> VGG feature i (ascending depth):
        (B,1,128,128) (B,1,64,64) (B,1,32,32) (B,1,16,16) (B,1,8,8) (B,4096) (B,365)

Places365 and scikit-image are not available offline, so the spatial masks are unions of
1-4 random axis-aligned rectangles instead of skimage ``random_shapes``; the stage
distribution and the nearest-neighbour propagation to the finer levels follow
misc.get_masks_for_training (misc.py:28-55).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F

MASK_SHAPES: Tuple[Tuple[int, ...], ...] = ((1, 128, 128), (1, 64, 64), (1, 32, 32), (1, 16, 16), (1, 8, 8),
                                            (4096,), (365,))
NUM_CLASSES = 365


def masks_for_stage(stage: int, spatial: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """Per-sample masks (no batch dim).  ``stage`` counts from the DEEP end like the reference
    (0 = logits (365,), 6 = the (1,128,128) level).  With ``spatial`` (a 0/1 map at the level just
    finer than ``stage``) all finer levels get its nearest-neighbour upsampling, all deeper levels
    are zero (misc.py:47-55)."""
    rev = tuple(reversed(MASK_SHAPES))
    out = []
    for idx, shp in enumerate(rev):
        if idx == stage:
            out.append(torch.ones(shp))
        elif spatial is not None and idx > stage:
            out.append(F.interpolate(spatial[None, None].float(), size=shp[1:], mode="nearest")[0])
        else:
            out.append(torch.zeros(shp))
    out.reverse()
    return out


def random_rect_mask(size: int, g: torch.Generator) -> torch.Tensor:
    """0/1 map (1 = keep): background 1, 1-4 rectangles of zeros, like skimage shapes != 255."""
    m = torch.ones(size, size)
    n = int(torch.randint(1, 5, (1,), generator=g))
    lo = min(8, size // 2)
    for _ in range(n):
        h = int(torch.randint(lo, size + 1, (1,), generator=g))
        w = int(torch.randint(lo, size + 1, (1,), generator=g))
        y = int(torch.randint(0, size - h + 1, (1,), generator=g))
        x = int(torch.randint(0, size - w + 1, (1,), generator=g))
        m[y:y + h, x:x + w] = 0.0
    return m


def training_masks(g: torch.Generator, p_random_mask: float = 0.3) -> List[torch.Tensor]:
    """One sample's masks with the distribution of misc.get_masks_for_training."""
    choices = list(range(7)) + [0, 1]
    stage = choices[int(torch.randint(0, len(choices), (1,), generator=g))]
    spatial = None
    if float(torch.rand(1, generator=g)) < p_random_mask and 0 < stage < 6:
        size = tuple(reversed(MASK_SHAPES))[stage + 1][1]
        spatial = random_rect_mask(size, g)
    return masks_for_stage(stage, spatial)


def training_masks_device(batch_size: int, device, generator: Optional[torch.Generator] = None,
                          p_random_mask: float = 0.3, seed: Optional[int] = None) -> List[torch.Tensor]:
    """A whole batch of training masks generated ON THE DEVICE by one kernel launch (sp_training_masks): the
    reference builds them per sample in DataLoader workers (misc.py:13-68), which cannot feed a thousand images per second.
    Same contract as training_masks(): stage ~ choice([0..6, 0, 1]) counted from the deep end; with probability p_random_mask
    and 0 < stage < 6 a 0/1 map of 1-4 zero rectangles at the level just finer than the stage is nearest-upsampled to all finer
    levels; everything deeper than the stage is zero; values are exact 0.0 / 1.0.  ``seed``: the 64-bit seed of this batch
    (the kernel is counter based: no device RNG state, no host sync).  Without it one 63-bit draw is taken from ``generator``
    (a CPU generator: no device sync; re-seeding it restarts the mask sequence, and its state travels with it - round-3 ADVICE:
    a module-level counter keyed by id(generator) did neither); without a generator, from a module-private generator seeded once
    from torch.initial_seed() and the process group's rank (_private_generator)."""
    import ctypes
    from . import _lib as L
    from . import ops
    dev = torch.device(device)
    if dev.type != "cuda":
        raise L.SempyrError("training_masks_device: needs a CUDA/HIP device (no CPU path; tests use the oracle's restatement)")
    if seed is None:
        if generator is not None and generator.device.type != "cpu":
            raise L.SempyrError("training_masks_device: pass a CPU torch.Generator (a device generator would cost a host sync per batch)")
        if generator is None:
            generator = _private_generator()
        seed = int(torch.randint(0, (1 << 63) - 1, (1,), dtype=torch.int64, generator=generator))
    out = [torch.empty((batch_size,) + shp, dtype=torch.float32, device=dev) for shp in MASK_SHAPES]
    L.call("sp_training_masks", *[ops.ptr(t) for t in out], batch_size, ctypes.c_uint64(seed), float(p_random_mask), ops.stream())
    return out


_PRIVATE_GEN = [None]


def _private_generator() -> torch.Generator:
    """The module's own CPU generator for callers that pass none: seeded ONCE from torch.initial_seed() and the rank of the process
    group (torch.distributed when initialised, so that the ranks of a data-parallel job draw different masks) - drawing from torch's
    global generator on every batch perturbed every other consumer of it: DataLoader seeds, shuffling, user code (round-4 ADVICE)."""
    if _PRIVATE_GEN[0] is None:
        rank = 0
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                rank = dist.get_rank()
        except Exception:
            rank = 0
        g = torch.Generator(device="cpu")
        g.manual_seed((torch.initial_seed() + rank * 0x9E3779B97F4A7C15) & ((1 << 63) - 1))
        _PRIVATE_GEN[0] = g
    return _PRIVATE_GEN[0]


def bernoulli_masks(g: torch.Generator, p: float = 0.5) -> List[torch.Tensor]:
    return [(torch.rand(s, generator=g) < p).float() for s in MASK_SHAPES]


def stack_masks(per_sample: List[List[torch.Tensor]]) -> List[torch.Tensor]:
    return [torch.stack([m[i] for m in per_sample], dim=0) for i in range(len(MASK_SHAPES))]


def synthetic_batch(batch_size: int, seed: int, resolution: int = 256):
    """(images, labels, masks) on the CPU; deterministic in (batch_size, seed)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    images = torch.rand(batch_size, 3, resolution, resolution, generator=g) * 2.0 - 1.0
    cls = torch.randint(0, NUM_CLASSES, (batch_size,), generator=g)
    labels = F.one_hot(cls, NUM_CLASSES).to(torch.long)
    masks = stack_masks([training_masks(g) for _ in range(batch_size)])
    return images, labels, masks
