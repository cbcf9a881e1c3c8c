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
    """0.85 * SSIM error (3x3 mean filter, reflection padding) + 0.15 * L1 (src/losses/photometric.py:54-88).

    Only the reference's operating point `weight_ssim=0.85` (the one `ReconstructionLoss` builds,
    src/losses/reconstruction.py:38) is compiled into the kernels."""
    def __init__(self, weight_ssim: float = 0.85):
        super().__init__()
        if not (0 <= weight_ssim <= 1): raise ValueError(f'Invalid SSIM weight. ({weight_ssim} vs. [0, 1])')
        if abs(weight_ssim - 0.85) > 1e-12: raise NotImplementedError('the HIP photometric kernels are built for weight_ssim=0.85')
        self.weight_ssim, self.weight_l1 = weight_ssim, 1 - weight_ssim

    def forward(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        from .. import functional as F
        return F.photo_error(pred, target, loss_name='ssim')
