"""Small tensor helpers whose exact semantics the loss path depends on (reference: `src/tools/ops.py`)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

__all__ = ['eps', 'mean_normalize', 'interpolate_like', 'expand_dim', 'eye_like', 'standardize', 'unstandardize']

_MEAN = (0.485, 0.456, 0.406)
_STD = (0.229, 0.224, 0.225)


def eps(x: torch.Tensor | None = None) -> float:
    """Machine epsilon of `x.dtype` (float32 if None): 1.1920929e-07 on this path (src/tools/ops.py:63-66)."""
    return torch.finfo(torch.float32 if x is None else x.dtype).eps


def mean_normalize(x: torch.Tensor, dim=(2, 3)) -> torch.Tensor:
    """x / clamp(mean_dim(x), eps) (src/tools/ops.py:279-286)."""
    return x/x.mean(dim=dim, keepdim=True).clamp(min=eps(x))


def interpolate_like(input: torch.Tensor, other: torch.Tensor, mode: str = 'nearest', align_corners: bool = False) -> torch.Tensor:
    """Resize `input` to the spatial size of `other` (src/tools/ops.py:311-314)."""
    if mode == 'nearest': align_corners = None
    return F.interpolate(input, size=other.shape[-2:], mode=mode, align_corners=align_corners)


def expand_dim(x: torch.Tensor, num, dim=0, insert: bool = False) -> torch.Tensor:
    """Expand (optionally freshly inserted) dimension(s) `dim` to size(s) `num` (src/tools/ops.py:317-344)."""
    if isinstance(num, int):
        if isinstance(dim, int): num, dim = [num], [dim]
        else: num = [num]*len(dim)
    elif len(num) != len(dim):
        raise ValueError(f'Non-matching expansion and dims. ({len(num)} vs. {len(dim)})')
    if insert:
        for d in dim: x = x.unsqueeze(d)
    sizes = [-1]*x.ndim
    for k, d in zip(num, dim): sizes[d] = k
    return x.expand(sizes)


def eye_like(x: torch.Tensor) -> torch.Tensor:
    """Identity matrices shaped like `x` (*, n, n) (src/tools/ops.py:292-308)."""
    if x.ndim < 2: raise ValueError(f'Input must have at least two dimensions! Got "{x.ndim}"')
    n, n2 = x.shape[-2:]
    if n != n2: raise ValueError(f'Input last two dimensions must be square (*, n, n)! Got "{x.shape}"')
    return torch.eye(n, dtype=x.dtype, device=x.device).expand_as(x).clone()


def standardize(x: torch.Tensor, mean=_MEAN, std=_STD) -> torch.Tensor:
    """ImageNet standardisation of (*, 3, h, w) images (src/tools/ops.py:250-257)."""
    shape = [1]*(x.ndim - 3) + [3, 1, 1]
    return (x - x.new_tensor(mean).view(shape))/x.new_tensor(std).view(shape)


def unstandardize(x: torch.Tensor, mean=_MEAN, std=_STD) -> torch.Tensor:
    shape = [1]*(x.ndim - 3) + [3, 1, 1]
    return x*x.new_tensor(std).view(shape) + x.new_tensor(mean).view(shape)
