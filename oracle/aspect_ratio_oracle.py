"""CPU oracle of the aspect-ratio augmentation (reference: `src/core/aspect_ratio.py:35-166`).  TEST INFRASTRUCTURE ONLY.

Parity status: the RESIZE half (`resize_aug`, `sample_resize`), the sampling (`sample_crop`) and the intrinsics updates
(`centre_crop_K`, `resize_K`) are PINNED on vectors produced by importing the reference (`tests/golden/ar_*.npz`,
`make_golden.py: run_aspect_cases`).  The CROP half is "parity unpinned": the reference calls
`kornia.geometry.transform.center_crop(x, size, mode='bilinear', align_corners=False)` (aspect_ratio.py:78) and kornia (pinned
`kornia=0.6.10` in the reference's `docker/environment.yml`) is not installed in the build image, so no vector of it can be produced
here.  `center_crop` below restates kornia 0.6.10's published call chain with torch's own `affine_grid` / `grid_sample`:

  `center_crop` (kornia/geometry/transform/crop2d.py): source box = the integer window starting at `int(W/2 - w/2), int(H/2 - h/2)`,
      corners `start .. start + size - 1`; destination box `0 .. size - 1`  ->  `crop_by_boxes`
  `crop_by_boxes`: `get_perspective_transform(src_box, dst_box)` (here: the exact translation `dst = src - start` it solves for)
      ->  `crop_by_transform_mat`  ->  `warp_affine(x, M[:, :2], (h, w), mode, padding_mode='zeros', align_corners)`
  `warp_affine` (imgwarp.py): `normalize_homography(M, (H, W), (h, w))` = `N_dst @ M @ inv(N_src)` with
      `normal_transform_pixel(n)`: `norm = 2 pix/(n - 1) - 1` — the align_corners=TRUE convention, whatever `align_corners` is —
      then `F.affine_grid(inv(.)[:, :2], [B, C, h, w], align_corners)` and `F.grid_sample(x, grid, mode, 'zeros', align_corners)`.

With the caller's `align_corners=False` the two conventions do not cancel: output column i is sampled at

    x(i) = ((i + 0.5)(w - 1)/w + x0) * W/(W - 1) - 0.5          (and likewise in y)

— a bilinear resample that is slightly zoomed about the window (W=640, w=320, x0=160: column 0 reads x = 160.25), with zero padding
for taps outside the image — and NOT the slice `x0 + i`, which it is only for `align_corners=True` (`test_oracle_golden.py` checks
that reduction, the one anchor available here).  The same (n - 1) quirk as `ViewSynth` (SURVEY.md:296-298).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _normal_transform_pixel(height: int, width: int, dtype) -> torch.Tensor:
    """kornia.geometry.conversions.normal_transform_pixel: pixel -> [-1, 1] with the (n - 1) denominators."""
    eps = 1e-14
    t = torch.tensor([[1.0, 0.0, -1.0], [0.0, 1.0, -1.0], [0.0, 0.0, 1.0]], dtype=dtype)
    t[0, 0] = t[0, 0]*2.0/(eps if width == 1 else width - 1.0)
    t[1, 1] = t[1, 1]*2.0/(eps if height == 1 else height - 1.0)
    return t


def crop_window(shape, size):
    """-> (y0, x0): kornia's `center_crop` start offsets (truncation, crop2d.py)."""
    H, W = int(shape[0]), int(shape[1])
    h, w = int(size[0]), int(size[1])
    return int(H/2 - h/2), int(W/2 - w/2)


def center_crop(x: torch.Tensor, size, align_corners: bool = False, mode: str = 'bilinear') -> torch.Tensor:
    """kornia.geometry.transform.center_crop(x, size, mode, padding_mode='zeros', align_corners) restated (module docstring):
    (..., H, W) -> (..., h, w).  The reference passes `align_corners=False` (src/core/aspect_ratio.py:78)."""
    H, W = x.shape[-2:]
    h, w = int(size[0]), int(size[1])
    y0, x0 = crop_window((H, W), (h, w))
    lead = x.shape[:-2]
    src = x.reshape(-1, 1, H, W)
    dt = src.dtype
    M = torch.tensor([[1.0, 0.0, -float(x0)], [0.0, 1.0, -float(y0)], [0.0, 0.0, 1.0]], dtype=dt)     # dst_pix <- src_pix: what get_perspective_transform solves for
    dst_norm_trans_src_norm = _normal_transform_pixel(h, w, dt) @ (M @ torch.linalg.inv(_normal_transform_pixel(H, W, dt)))   # normalize_homography
    src_norm_trans_dst_norm = torch.linalg.inv(dst_norm_trans_src_norm)
    theta = src_norm_trans_dst_norm[None, :2, :].expand(src.shape[0], 2, 3)
    grid = F.affine_grid(theta, [src.shape[0], 1, h, w], align_corners=align_corners)
    out = F.grid_sample(src, grid, mode=mode, padding_mode='zeros', align_corners=align_corners)
    return out.reshape(*lead, h, w)


def crop_source_coords(n_out: int, start: int, n_in: int, dtype=torch.float64) -> torch.Tensor:
    """Closed form of where `center_crop(align_corners=False)` samples along one axis: x(i) = ((i + 0.5)(n_out - 1)/n_out + start) n_in/(n_in - 1) - 0.5."""
    i = torch.arange(n_out, dtype=dtype)
    return ((i + 0.5)*(n_out - 1)/n_out + start)*n_in/(n_in - 1) - 0.5


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
    # (the not-applied branch of the augmentation resizes without a crop, aspect_ratio.py:60: `crop_shape` == input shape means "no crop")
    crop = (lambda t: t) if tuple(int(v) for v in crop_shape) == sh else (lambda t: center_crop(t, crop_shape))
    outs = [resize(crop(t.float()), out_shape) for t in tensors]
    if K is not None: K = resize_K(centre_crop_K(K.float(), crop_shape, sh), out_shape, crop_shape)
    return outs, K
