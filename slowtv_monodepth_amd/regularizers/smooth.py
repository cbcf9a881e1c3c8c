"""`SmoothReg` — registry key `disp_smooth` (reference: `src/regularizers/smooth.py:51-97`)."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..registry import register

__all__ = ['SmoothReg']


@register('disp_smooth')
class SmoothReg(nn.Module):
    """First-order (optionally edge-aware) smoothness of the mean-normalised disparity, as one HIP launch per phase.

    :param use_edges: down-weight disparity gradients by exp(-|image gradient|) (Monodepth).
    :param use_laplacian: second-order differences (DVSO) instead of first-order ones (src/regularizers/smooth.py:33-48).
    :param use_blur: Gaussian pre-blur through `kornia.filters.gaussian_blur2d` (smooth.py:21): kornia is absent from the build image,
        so neither golden vectors nor a pinned oracle can be produced for it — refused rather than shipped unverified.
    """
    def __init__(self, use_edges: bool = False, use_laplacian: bool = False, use_blur: bool = False) -> None:
        super().__init__()
        if use_blur: raise NotImplementedError('use_blur needs kornia.filters.gaussian_blur2d, which cannot be pinned in this build (SURVEY.md §2.1)')
        self.use_edges, self.use_laplacian, self.use_blur = use_edges, use_laplacian, use_blur

    def forward(self, disp: torch.Tensor, img: torch.Tensor):
        """disp (b,1,h,w), img (b,3,h',w') (resized to the disparity's size exactly as `handlers.disp_smooth` does when
        h', w' differ) -> (loss, {'disp_grad', 'image_grad'})."""
        from .. import functional as F
        loss, dg, ig = F.disp_smooth_fused({0: disp}, img, use_edges=self.use_edges, want_aux=True, use_laplacian=self.use_laplacian)
        return loss, {'disp_grad': dg, 'image_grad': ig}
