"""Pose / intrinsics prologue and the `ViewSynth` operator (reference: `src/tools/geometry.py`).

`T_from_AAt` on GPU tensors is the `smd_pose_*` kernel pair (the training step goes through
`functional.pose_matrices` / `functional.intrinsics` directly, which also fold in the inverse transforms and K^-1);
on host tensors — dataset poses, fixtures — it is the same formula in PyTorch.  `to_scaled` / `to_inv` / `resize_K` /
`build_K` are the reference's small helpers kept for API parity (the training step uses the K0 and intrinsics kernels).
`ViewSynth` is the class-level drop-in whose forward/backward are HIP kernels (`smd_view_synth_*`).  The fused fast
path (`handlers.image_recon`) never instantiates point clouds or sampling grids at all.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops

__all__ = ['to_scaled', 'to_inv', 'T_from_AAt', 'resize_K', 'build_K', 'ViewSynth']


def to_inv(depth: torch.Tensor) -> torch.Tensor:
    """depth <-> disparity: `(d > 0) / d.clamp(min=eps)` (src/tools/geometry.py:86-90)."""
    return (depth > 0)/depth.clamp(min=ops.eps(depth))


def to_scaled(disp: torch.Tensor, min: float = 0.01, max: float | None = 100) -> tuple[torch.Tensor, torch.Tensor]:
    """Sigmoid disparity -> (scaled disparity, depth) (src/tools/geometry.py:62-76)."""
    if min <= 0: raise ValueError(f'Min depth must be greater than 0. ({min})')
    if max and (max < min): raise ValueError(f'Max depth must be greater than min. ({max} vs. {min})')
    i_max, i_min = 1/min, (1/max) if max else 0
    disp = (i_max - i_min)*disp + i_min
    return disp, to_inv(disp)


def T_from_AAt(aa: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """Axis-angle (*,3) + translation (*,3) -> (*,4,4) by Rodrigues' formula (src/tools/geometry.py:181-209)."""
    s1, s2 = aa.shape, t.shape
    if s1[-1] != 3: raise ValueError(f'Incorrect `axisangle` shape. ({s1} vs. (*, 3)')
    if s2[-1] != 3: raise ValueError(f'Incorrect `t` shape. ({s2} vs. (*, 3)')
    if s1 != s2: raise ValueError(f'Non-matching shapes. ({s1} vs. {s2}')
    if aa.is_cuda and aa.dtype == torch.float32 and t.dtype == torch.float32:
        from . import functional as F
        return F.pose_matrices(aa.reshape(-1, 3), t.reshape(-1, 3)).reshape(*s1[:-1], 4, 4)
    angle = aa.norm(p=2, dim=-1, keepdim=True)
    x, y, z = (aa/angle.clip(min=ops.eps(angle))).unbind(-1)
    o = torch.zeros_like(x)
    W = torch.stack([o, -z, y, z, o, -x, -y, x, o], dim=-1).unflatten(-1, (3, 3))
    ang = angle.unsqueeze(-1)
    R = torch.eye(3, dtype=aa.dtype, device=aa.device) + W*ang.sin() + (W @ W)*(1 - ang.cos())
    top = torch.cat((R, t.unsqueeze(-1)), dim=-1)                                   # (*,3,4)
    bottom = aa.new_tensor([0, 0, 0, 1]).expand(*s1[:-1], 1, 4)
    return torch.cat((top, bottom), dim=-2)


def build_K(fs: torch.Tensor, cs: torch.Tensor) -> torch.Tensor:
    """Normalised focal lengths / principal point (b,2) -> (b,4,4) (src/networks/pose.py:60-73)."""
    o, l = torch.zeros_like(fs[:, 0]), torch.ones_like(fs[:, 0])
    rows = [torch.stack([fs[:, 0], o, cs[:, 0], o], -1), torch.stack([o, fs[:, 1], cs[:, 1], o], -1),
            torch.stack([o, o, l, o], -1), torch.stack([o, o, o, l], -1)]
    return torch.stack(rows, dim=-2)


def resize_K(K: torch.Tensor, new_shape: tuple[int, int], shape: tuple[int, int] | None = None) -> torch.Tensor:
    """Scale rows 0/1 of (*,4,4) intrinsics by the width/height ratio (src/tools/geometry.py:249-263)."""
    if shape is None: shape = (1, 1)
    scale = K.new_tensor([new_shape[1]/shape[1], new_shape[0]/shape[0], 1, 1]).view(4, 1)
    return K*scale


class ViewSynth(nn.Module):
    """Warp `input` from the support view into the target view given target depth, relative pose and intrinsics
    (`ViewSynth.forward`, src/tools/geometry.py:353-391).  Forward and backward run as HIP kernels; gradients flow to
    `input`, `depth`, `T`, `K` and `K_inv`.

    :param shape: (h, w) of the maps this instance will warp.
    """
    def __init__(self, shape: tuple[int, int]):
        super().__init__()
        self.shape = tuple(int(s) for s in shape)

    def forward(self, input: torch.Tensor, depth: torch.Tensor, T: torch.Tensor, K: torch.Tensor, K_inv: torch.Tensor | None = None):
        """:return: (input_warp (b,c,h,w), depth_warp (b,1,h,w), mask_valid (b,1,h,w) bool)"""
        from . import functional as F
        if tuple(depth.shape[-2:]) != self.shape: raise ValueError(f'ViewSynth built for {self.shape}, got depth {tuple(depth.shape)}')
        if K_inv is None: K_inv = torch.linalg.inv(K)
        return F.view_synth(input, depth, T, K, K_inv)
