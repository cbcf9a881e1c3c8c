"""`ReconstructionLoss` — registry keys `img_recon`, `feat_recon`, `autoenc_recon` (reference: `src/losses/reconstruction.py:12-126`)."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..registry import register

__all__ = ['ReconstructionLoss']


@register(('img_recon', 'feat_recon', 'autoenc_recon'))
class ReconstructionLoss(nn.Module):
    """Photometric reconstruction loss between synthesised support views and the target.

    :param loss_name: 'ssim' (0.85 SSIM + 0.15 L1), 'l1', or 'l2' (Euclidean feature distance, used by `feat_recon`).
    :param use_min: minimum over the support views per pixel (Monodepth2) instead of their mean.
    :param use_automask: drop pixels whose un-warped support frame already matches the target better (Monodepth2).
    :param mask_name: None, 'explainability' (err * mask) or 'uncertainty' (err * exp(-mask) + mask): how a predictive per-support
        weighting mask (b,n,h,w) passed to `forward` / `compute_photo` enters the error (reconstruction.py:46-57).

    When called through `handlers.image_recon` on 3-channel images with 'ssim'/'l1' and without masks the whole warp + error +
    reduction runs as ONE fused kernel; any other channel count, 'l2', a predictive mask, or calling the module directly on
    already-warped tensors (`crit(pred, target, source)`) runs the un-fused HIP operators.
    """
    def __init__(self, loss_name: str = 'ssim', use_min: bool = False, use_automask: bool = False, mask_name: str | None = None):
        super().__init__()
        if mask_name not in {'explainability', 'uncertainty', None}: raise ValueError(f'Invalid mask type: {mask_name}')
        if loss_name not in {'ssim', 'l1', 'l2'}: raise KeyError(loss_name)
        self.loss_name, self.use_min, self.use_automask, self.mask_name = loss_name, use_min, use_automask, mask_name
        self.noise_seed = 0  # advanced on every call so the in-kernel tie-break noise differs between steps

    def next_seed(self) -> int:
        self.noise_seed += 1
        return self.noise_seed

    def compute_photo(self, pred: torch.Tensor, target: torch.Tensor, mask=None) -> torch.Tensor:
        """(*n,b,c,h,w) predictions vs (b,c,h,w) target -> reduced error (b,1,h,w) (reconstruction.py:79-96).

        The per-support errors are differentiable (`functional.photo_error`), the reduced map returned here is not (the
        reduction kernel's selection is applied without a graph): it serves the comparisons of `handlers.depth_regr`
        (src/core/handlers.py:236-254), which the reference also evaluates for its value only."""
        from .. import functional as F
        if self.mask_name and mask is None: raise ValueError("Must provide a 'mask' when masking...")
        if pred.ndim == 4: pred = pred[None]
        n, b = pred.shape[:2]
        err = F.photo_error(pred.flatten(0, 1), target[None].expand_as(pred).flatten(0, 1), loss_name=self.loss_name)
        err = err.view(n, b, *err.shape[-2:])
        return F.recon_reduce(err, None, use_min=self.use_min, mask=mask, mask_name=self.mask_name)[1].unsqueeze(1)

    def forward(self, pred: torch.Tensor, target: torch.Tensor, source: torch.Tensor | None = None, mask=None, noise=None):
        """:return: (loss (), {'automask': (b,1,h,w) bool} if automasking)"""
        from .. import functional as F
        if self.mask_name and mask is None: raise ValueError("Must provide a 'mask' when masking...")
        if self.use_automask and source is None: raise ValueError("Must provide the original 'source' images when automasking...")
        if pred.ndim == 4: pred = pred[None]
        n, b = pred.shape[:2]
        tgt = target[None].expand_as(pred).flatten(0, 1)
        err_warp = F.photo_error(pred.flatten(0, 1), tgt, loss_name=self.loss_name).view(n, b, *pred.shape[-2:])
        err_static = None
        if self.use_automask:
            if source.ndim == 4: source = source[None]
            err_static = F.photo_error(source.flatten(0, 1), tgt, loss_name=self.loss_name).view(n, b, *pred.shape[-2:])
        loss, err, sel = F.recon_reduce(err_warp, err_static, use_min=self.use_min, noise=noise, seed=self.next_seed(),
                                        mask=mask if self.mask_name else None, mask_name=self.mask_name)
        ld = {'automask': (sel != 255).unsqueeze(1)} if self.use_automask else {}
        return loss, ld
