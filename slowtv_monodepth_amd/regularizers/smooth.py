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
    :param use_laplacian / use_blur: reference options that no BASELINE configuration enables; not accelerated.
    """
    def __init__(self, use_edges: bool = False, use_laplacian: bool = False, use_blur: bool = False) -> None:
        super().__init__()
        if use_laplacian or use_blur: raise NotImplementedError('use_laplacian / use_blur are outside the accelerated path (SURVEY.md §2.1)')
        self.use_edges, self.use_laplacian, self.use_blur = use_edges, use_laplacian, use_blur

    def forward(self, disp: torch.Tensor, img: torch.Tensor):
        """disp (b,1,h,w), img (b,3,h',w') (resized to the disparity's size exactly as `handlers.disp_smooth` does when
        h', w' differ) -> (loss, {'disp_grad', 'image_grad'})."""
        from .. import functional as F
        loss, dg, ig = F.disp_smooth_fused({0: disp}, img, use_edges=self.use_edges, want_aux=True)
        return loss, {'disp_grad': dg, 'image_grad': ig}
