"""The four loss modules of /root/reference/lossfunction.py with identical forward signatures and
``__repr__`` strings (the reference logs them, model_wrapper.py:88-91), computing through libsempyr.so."""
from typing import List, Tuple

import torch
import torch.nn as nn

from . import ops


class SemanticReconstructionLoss(nn.Module):
    """lossfunction.py:8-68."""

    def __init__(self) -> None:
        super().__init__()
        self.max_pooling_2d = nn.MaxPool2d(2)
        self.max_pooling_1d = nn.MaxPool1d(2)

    def __repr__(self):
        return '{}, maxpool kernel size{}'.format(self.__class__.__name__, self.max_pooling_1d.kernel_size)

    def forward(self, features_real: List[torch.Tensor], features_fake: List[torch.Tensor],
                masks: List[torch.Tensor], weight: float = 1.0) -> torch.Tensor:
        """`weight` (not in the reference's signature): the factor model_wrapper.py:186 multiplies the loss by, folded into the
        kernels - the value returned is weight * loss."""
        assert len(features_real) == len(features_fake) == len(masks)
        return ops.semantic_reconstruction_loss(features_real, features_fake, masks, weight)


class DiversityLoss(nn.Module):
    """lossfunction.py:71-110."""

    def __init__(self) -> None:
        super().__init__()
        self.l1_loss = nn.L1Loss(reduction='mean')

    def __repr__(self):
        return self.__class__.__name__

    def forward(self, images_fake: torch.Tensor, latent_inputs: torch.Tensor, weight: float = 1.0) -> torch.Tensor:
        """`weight` (not in the reference's signature): model_wrapper.py:184's factor, folded into the kernels."""
        assert images_fake.shape[0] > 1
        return ops.diversity_loss(images_fake, latent_inputs, weight)


class LSGANGeneratorLoss(nn.Module):
    """lossfunction.py:115-137."""

    def __repr__(self):
        return str(self.__class__.__name__)

    def forward(self, prediction_fake: torch.Tensor) -> torch.Tensor:
        return ops.sqerr_loss(prediction_fake, 1.0)


class LSGANDiscriminatorLoss(nn.Module):
    """lossfunction.py:140-164."""

    def __repr__(self):
        return str(self.__class__.__name__)

    def forward(self, prediction_real: torch.Tensor, prediction_fake: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return ops.sqerr_loss(prediction_real, 1.0), ops.sqerr_loss(prediction_fake, 0.0)
