"""Synthetic three-frame batches with the layout and value ranges of the reference's dataloader output
(`(x, y, m)` of src/datasets/base_mde.py:158-184; KITTI intrinsics of src/datasets/kitti_raw.py:76-81).

There is no dataset on the GPU box, so training throughput is measured on device-resident synthetic triplets:
a smooth random texture (sum of sinusoids + 5 % noise, so SSIM is non-degenerate) and copies of it shifted by a
few pixels as support frames (so the warps matter and roughly half the pixels survive automasking).
"""
from __future__ import annotations

import math

import torch

from . import ops

__all__ = ['texture', 'kitti_K', 'make_batch']


def texture(gen: torch.Generator, b: int, h: int, w: int, shift=(0.0, 0.0), device='cpu', waves: int = 6, coeffs=None):
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32, device=device),
                            torch.arange(w, dtype=torch.float32, device=device), indexing='ij')
    xs, ys = xs + shift[0], ys + shift[1]
    if coeffs is None:
        coeffs = dict(f=torch.rand(waves, 2, generator=gen, device=device)*0.35 + 0.02,
                      ph=torch.rand(waves, b, 3, 1, 1, generator=gen, device=device)*2*math.pi,
                      amp=torch.rand(waves, b, 3, 1, 1, generator=gen, device=device),
                      noise=torch.rand(b, 3, h, w, generator=gen, device=device))
    img = torch.zeros(b, 3, h, w, device=device)
    for k in range(waves):
        img = img + coeffs['amp'][k]*torch.sin(coeffs['f'][k, 0]*xs + coeffs['f'][k, 1]*ys + coeffs['ph'][k])
    lo, hi = img.amin(dim=(2, 3), keepdim=True), img.amax(dim=(2, 3), keepdim=True)
    img = 0.95*(img - lo)/(hi - lo).clamp(min=1e-6) + 0.05*coeffs['noise']
    return img.clamp(0, 1), coeffs


def kitti_K(b: int, h: int, w: int, device='cpu') -> torch.Tensor:
    K = torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=torch.float32, device=device)
    return K[None].repeat(b, 1, 1)


def make_batch(b: int, h: int, w: int, supp_idxs=(-1, 1), seed: int = 42, device='cpu'):
    """-> (x, y, m): x = network inputs (ImageNet-standardised), y = loss inputs (raw [0,1] images, K), m = metadata."""
    gen = torch.Generator(device=device).manual_seed(seed)
    imgs, coeffs = texture(gen, b, h, w, device=device)
    supp = []
    for k, i in enumerate(supp_idxs):
        dx = (2.0 + k)*(1 if i > 0 else -1)
        supp.append(texture(gen, b, h, w, shift=(dx, 0.5*dx), device=device, coeffs=coeffs)[0])
    supp = torch.stack(supp)
    x = {'imgs': ops.standardize(imgs), 'supp_imgs': ops.standardize(supp), 'supp_idxs': torch.tensor(list(supp_idxs))}
    y = {'imgs': imgs, 'supp_imgs': supp, 'K': kitti_K(b, h, w, device)}
    m = {'supp': [str(i) for i in supp_idxs]}
    return x, y, m
