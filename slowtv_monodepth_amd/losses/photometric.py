"""Dense photometric errors (reference: `src/losses/photometric.py`), evaluated by HIP kernels (`smd_photo_error_*`)."""
from __future__ import annotations

import torch
import torch.nn as nn

__all__ = ['DenseL1Error', 'DenseL2Error', 'PhotoError']


class DenseL1Error(nn.Module):
    """|pred - target| averaged over channels (src/losses/photometric.py:11-14)."""
    def forward(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        from .. import functional as F
        return F.photo_error(pred, target, loss_name='l1')


class DenseL2Error(nn.Module):
    """Euclidean distance over channels, sqrt(clamp(sum (pred - target)^2, eps)) (src/losses/photometric.py:17-20)."""
    def forward(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        from .. import functional as F
        return F.photo_error(pred, target, loss_name='l2')


class PhotoError(nn.Module):
    """weight_ssim * SSIM error (3x3 mean filter, reflection padding) + (1 - weight_ssim) * L1 (src/losses/photometric.py:54-88);
    `weight_ssim=0` drops the SSIM term, `weight_ssim=1` the L1 term.  (The FUSED reconstruction kernels are built for the 0.85
    that `ReconstructionLoss` constructs, src/losses/reconstruction.py:38; this class runs on the un-fused operator.)"""
    def __init__(self, weight_ssim: float = 0.85):
        super().__init__()
        if not (0 <= weight_ssim <= 1): raise ValueError(f'Invalid SSIM weight. ({weight_ssim} vs. [0, 1])')
        self.weight_ssim, self.weight_l1 = weight_ssim, 1 - weight_ssim

    def forward(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        from .. import functional as F
        return F.photo_error(pred, target, loss_name='ssim', weight_ssim=self.weight_ssim)
