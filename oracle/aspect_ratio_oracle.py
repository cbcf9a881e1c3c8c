"""CPU oracle of the aspect-ratio augmentation (reference: `src/core/aspect_ratio.py:35-166`).  TEST INFRASTRUCTURE ONLY.

Parity status: the RESIZE half (`resize_aug`, `sample_resize`), the sampling (`sample_crop`) and the intrinsics updates
(`centre_crop_K`, `resize_K`) are PINNED on vectors produced by importing the reference (`tests/golden/ar_*.npz`,
`make_golden.py: run_aspect_cases`).  The CROP half is "parity unpinned": the reference calls
`kornia.geometry.transform.center_crop(x, size, mode='bilinear', align_corners=False)` (aspect_ratio.py:78) and kornia 0.6.x is
not installed in the build image, so no vector of it can be produced here.  `center_crop` below restates kornia's published
algorithm: the source box is the integer window starting at `int(H/2 - h/2), int(W/2 - w/2)`, the destination box the full
output, and the warp between two boxes of equal size is a pure integer translation — a slice.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def center_crop(x: torch.Tensor, size) -> torch.Tensor:
    """kornia.geometry.transform.center_crop restated (see the module docstring): (..., H, W) -> (..., h, w)."""
    H, W = x.shape[-2:]
    h, w = int(size[0]), int(size[1])
    y0, x0 = int(H/2 - h/2), int(W/2 - w/2)
    return x[..., y0:y0 + h, x0:x0 + w]


def resize(x: torch.Tensor, size) -> torch.Tensor:
    """`F.interpolate(x, size, mode='bilinear', align_corners=False)` on the last two dims (aspect_ratio.py:141)."""
    lead = x.shape[:-2]
    return F.interpolate(x.reshape(-1, 1, *x.shape[-2:]), size=tuple(int(v) for v in size), mode='bilinear', align_corners=False).reshape(*lead, *size)


def centre_crop_K(K, new_shape, shape):
    """src/tools/geometry.py:233-246."""
    K = K.clone()
    K[..., 0, 2] *= new_shape[1]/shape[1]
    K[..., 1, 2] *= new_shape[0]/shape[0]
    return K


def resize_K(K, new_shape, shape):
    """src/tools/geometry.py:249-263."""
    K = K.clone()
    K[..., 0, :] *= new_shape[1]/shape[1]
    K[..., 1, :] *= new_shape[0]/shape[0]
    return K


def crop_resize(tensors, crop_shape, out_shape, K=None):
    """`crop_aug` then `resize_aug` on a list of (..., H, W) tensors and the intrinsics (aspect_ratio.py:67-151)."""
    sh = tuple(tensors[0].shape[-2:])
    outs = [resize(center_crop(t.float(), crop_shape), out_shape) for t in tensors]
    if K is not None: K = resize_K(centre_crop_K(K.float(), crop_shape, sh), out_shape, crop_shape)
    return outs, K
