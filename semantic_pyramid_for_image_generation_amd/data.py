"""Batch contract of /root/reference/data.py (the JPEG/PIL producer itself is outside the hot path, SURVEY.md
section 2 row 5).  ``image_label_list_of_masks_collate_function`` stacks (image, one-hot long label, 7 masks)
samples exactly like data.py:68-90, minus the two ``requires_grad = True`` flags, which only create dead
gradients."""
from typing import List, Tuple

import torch


def image_label_list_of_masks_collate_function(batch: List[Tuple[torch.Tensor, torch.Tensor, List[torch.Tensor]]]) -> \
        Tuple[torch.Tensor, torch.Tensor, List[torch.Tensor]]:
    images = torch.stack([instance[0] for instance in batch], dim=0)
    labels = torch.stack([instance[1] for instance in batch], dim=0)
    masks = [torch.stack([instance[2][i] for instance in batch], dim=0) for i in range(len(batch[0][2]))]
    return images, labels, masks


def normalize_min_max_device(images: torch.Tensor, min_val: float = -1.0, max_val: float = 1.0, eps: float = 1e-6,
                             per_channel: bool = True) -> torch.Tensor:
    """``kornia.normalize_min_max(image[None], min_val=-1., max_val=1.)[0]`` of data.py:53 for a whole (B, C, H, W) fp32 batch ON THE
    DEVICE, one launch (sp_minmax_normalize): decoded images can be normalised where the training step consumes them instead of
    per sample in DataLoader workers.  per_channel=True is kornia's definition (min / max over each (image, channel) plane);
    False takes them over the whole image.  Bit-identical to the torch expression (tests/test_gpu_next_rows.py)."""
    from . import _lib as L
    from . import ops
    ops.require_gpu(images)
    if images.dim() != 4 or images.dtype != torch.float32:
        raise L.SempyrError("normalize_min_max_device: a float32 (B, C, H, W) batch is expected")
    x = images.contiguous()
    y = torch.empty_like(x)
    b, c, h, w = x.shape
    L.call("sp_minmax_normalize", ops.ptr(x), ops.ptr(y), b, c, h * w, float(min_val), float(max_val), float(eps), 1 if per_channel else 0,
           ops.stream())
    return y
