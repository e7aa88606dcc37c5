"""Mask generators and the metric logger of /root/reference/misc.py (the mask CONTRACT is an input of the
hot path; generation itself runs in the data pipeline).  scikit-image is optional: without
it the spatially varying masks use unions of random rectangles (synthetic.random_rect_mask)."""
import json
import os
import random
from typing import List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import synthetic

MASK_SHAPES = [(1, 128, 128), (1, 64, 64), (1, 32, 32), (1, 16, 16), (1, 8, 8), (4096,), (365,)]


def get_masks_for_training(mask_shapes: List[Tuple] = MASK_SHAPES, device: str = 'cpu', add_batch_size: bool = False,
                           p_random_mask: float = 0.3) -> List[torch.Tensor]:
    """misc.py:13-68."""
    rev = list(reversed(mask_shapes))
    selected_stage = random.choice(list(range(len(mask_shapes))) + [0, 1])
    spatial = (np.random.rand() < p_random_mask) and (0 < selected_stage < len(mask_shapes) - 1)
    random_mask = None
    if spatial:
        size = rev[selected_stage + 1][1]
        try:
            from skimage.draw import random_shapes
            img = random_shapes(rev[selected_stage + 1][1:], min_shapes=1, max_shapes=4, min_size=min(8, size // 2),
                                allow_overlap=True)[0][:, :, 0]
            random_mask = (torch.tensor(img, dtype=torch.float32)[None] == 255.0).float()
        except ImportError:
            g = torch.Generator().manual_seed(random.getrandbits(62))
            random_mask = synthetic.random_rect_mask(size, g)[None]
    masks = []
    for index, shape in enumerate(rev):
        if index == selected_stage:
            masks.append(torch.ones(shape, dtype=torch.float32, device=device))
        elif spatial and index > selected_stage:
            masks.append(F.interpolate(random_mask[None], size=shape[1:], mode='nearest')[0].to(device))
        else:
            masks.append(torch.zeros(shape, dtype=torch.float32, device=device))
    if add_batch_size:
        masks = [m.unsqueeze(dim=0) for m in masks]
    masks.reverse()
    return masks


def get_masks_for_inference(stage_index_to_choose: int, mask_shapes=tuple(MASK_SHAPES), device: str = 'cpu',
                            add_batch_size: bool = False) -> List[torch.Tensor]:
    """misc.py:78-97."""
    masks = []
    for index, shape in enumerate(reversed(mask_shapes)):
        fill = torch.ones if index == stage_index_to_choose else torch.zeros
        masks.append(fill(shape, dtype=torch.float32, device=device))
    if add_batch_size:
        masks = [m.unsqueeze(dim=0) for m in masks]
    masks.reverse()
    return masks


def get_masks_for_validation(mask_shapes=tuple(MASK_SHAPES), device: str = 'cpu', add_batch_size: bool = False):
    """misc.py:71-75."""
    return get_masks_for_inference(random.choice(range(len(mask_shapes))), mask_shapes, device, add_batch_size)


def normalize_0_1_batch(input: torch.Tensor) -> torch.Tensor:
    """misc.py:100-109: every sample of the batch mapped to [0, 1] by its own minimum / maximum."""
    flat = input.reshape(input.shape[0], -1)
    lo, hi = flat.min(dim=1)[0][:, None, None, None], flat.max(dim=1)[0][:, None, None, None]
    return (input - lo) / (hi - lo)


def image_grid(images: torch.Tensor, nrow: int = 8, padding: int = 2) -> torch.Tensor:
    """torchvision.utils.make_grid(images, nrow, padding, pad_value=0) for a (B, C, H, W) batch: (3, H', W') with `nrow` images per row."""
    images = images.detach().float().cpu()
    if images.shape[1] == 1:
        images = images.repeat(1, 3, 1, 1)
    b, c, h, w = images.shape
    xmaps = min(nrow, b)
    ymaps = (b + xmaps - 1) // xmaps
    hh, ww = h + padding, w + padding
    grid = torch.zeros((c, hh * ymaps + padding, ww * xmaps + padding), dtype=torch.float32)
    for k in range(b):
        y, x = divmod(k, xmaps)
        grid[:, y * hh + padding:y * hh + padding + h, x * ww + padding:x * ww + padding + w] = images[k]
    return grid


def save_image_grid(images: torch.Tensor, path: str, nrow: int = 8, padding: int = 2) -> None:
    """torchvision.utils.save_image(images, path, nrow=nrow) of model_wrapper.py:290-292 without torchvision (not a dependency of the
    hot path): make_grid's layout and save_image's rounding (x * 255 + 0.5, clamped to [0, 255], truncated), as an 8-bit RGB PNG."""
    import struct
    import zlib
    grid = image_grid(images, nrow, padding)
    rgb = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).contiguous().numpy()
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6))
                + chunk(b"IEND", b""))


def load_png_rgb8(path: str) -> "np.ndarray":
    """Decoder of what save_image_grid writes (8-bit RGB, filter 0 on every scanline): (H, W, 3) uint8 - for the tests."""
    import struct
    import zlib
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, tag = struct.unpack(">I", data[pos:pos + 4])[0], data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h = struct.unpack(">II", body[:8])
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + 3 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3).copy()


class Logger(object):
    """Dict-of-lists metric logger (misc.py:124-159)."""

    def __init__(self) -> None:
        self.metrics = dict()
        self.hyperparameter = dict()

    def log(self, metric_name: str, value: float) -> None:
        self.metrics.setdefault(metric_name, []).append(value)

    def save_metrics(self, path: str) -> None:
        with open(os.path.join(path, 'hyperparameter.txt'), 'w') as f:
            json.dump(self.hyperparameter, f)
        for name, values in self.metrics.items():
            torch.save(torch.tensor(values), os.path.join(path, '{}.pt'.format(name)))
