"""GPU-side aspect-ratio augmentation (reference: `src/core/aspect_ratio.py:35-166`; call site `src/core/trainer.py:54-60, 106`).

Same sampling as the reference — the same draws from `random` and `torch`'s generators in the same order (`MonoDepthModule.step`
calls this on every training step, as the reference's `training_step` does, so `random.random()` is drawn even at probability 0) —
and the same `m['augs']` bookkeeping; what differs is what runs underneath: the reference materialises the crop of all 2(1+n)b images
with kornia's `center_crop(mode='bilinear', align_corners=False)` and then resizes it with `F.interpolate`; here both steps, for every
image tensor of the batch and for the intrinsics, are ONE launch (`smd_crop_resize`, `csrc/smd_aspect.hip`).  The crop is kornia's
re-sampled window (NOT a slice: its (n - 1)-normalised warp under align_corners=False grids samples 0.25 px off the integer window at
640 -> 320), restated from kornia 0.6.10's published call chain — kornia is absent from the build image, so that half is "parity
unpinned"; the resize half, the sampling and the K updates are pinned on reference fixtures (`oracle/aspect_ratio_oracle.py`).
"""
from __future__ import annotations

import random

import torch

__all__ = ['aspect_ratio_aug', 'sample_crop', 'sample_resize', 'LABELS', 'RATIOS', 'RATIO2LABEL', 'LABEL2RATIO']

LABELS = [
    '6/13', '9/16', '3/5', '2/3', '4/5', '1/1',  # portrait
    '5/4', '4/3', '3/2', '14/9', '5/3', '16/9', '2/1', '24/10', '33/10', '18/5',  # landscape
]
RATIOS = [int(a)/int(b) for a, b in (l.split('/') for l in LABELS)]
RATIO2LABEL = dict(zip(RATIOS, LABELS))
LABEL2RATIO = dict(zip(LABELS, RATIOS))


def _num_pix(shape) -> int:
    assert len(shape) == 2
    return shape[0]*shape[1]


def _closest_multiple(i, n: int = 32) -> int: return round(i/n)*n


def sample_crop(shape, min: float = 0.5, max: float = 1.0):
    """Centre-crop size with a random aspect ratio; at least one side within [min, max] of the input (aspect_ratio.py:93-123).
    Draws: `torch.randint` twice (10 candidate heights, 10 candidate widths), `random.choice` of the ratio, `random.choice` of
    one valid candidate.  -> ((h, w), ratio)"""
    assert max >= min
    n = 10
    hs = torch.randint(int(shape[0]*min), int(shape[0]*max), (n,))
    ws = torch.randint(int(shape[1]*min), int(shape[1]*max), (n,))
    r = random.choice(RATIOS)
    hs, ws = torch.cat((hs, (ws/r).long())), torch.cat(((r*hs).long(), ws))
    valid = ((hs >= 0) & (hs <= shape[0]) & (ws >= 0) & (ws <= shape[1])).nonzero().flatten().tolist()
    i = random.choice(valid)
    return (int(hs[i]), int(ws[i])), r


def sample_resize(shape, ref_shape, eps: float = 0.8):
    """Resize target of the crop: its aspect ratio, at most `eps` of `ref_shape`'s pixels, sides multiples of 32 (aspect_ratio.py:154-166)."""
    mul = 32
    n, n_ref = _num_pix(shape), _num_pix(ref_shape)
    r = (n_ref/n)**0.5
    res = [_closest_multiple(r*i, n=mul) for i in shape]
    while _num_pix(res) > n_ref*eps: res = [i - mul for i in res]
    return res


def _hip_resample(tensors, crop_shape, out_shape, K):
    from . import functional as F
    return F.crop_resize(tensors, crop_shape, out_shape, K)


@torch.no_grad()
def aspect_ratio_aug(batch, p: float = 1.0, crop_min: float = 0.5, crop_max: float = 1.0, ref_shape=None, *, resample=None):
    """Change the aspect ratio of the training images: random centre crop, then a resize to at most 80 % of `ref_shape`'s pixels
    (aspect_ratio.py:35-64).  With probability 1 - p only the resize to `ref_shape`'s pixel count happens (if it differs).
    Returns the batch (dicts updated in place, tensors replaced).  `resample(tensors, crop_shape, out_shape, K) -> (tensors, K)`
    is the operator (default: the HIP kernel; the CPU tests inject the oracle's)."""
    x, y, m = batch
    resample = resample or _hip_resample
    sh = tuple(x['imgs'].shape[-2:])
    if random.random() > p:
        if not ref_shape or tuple(ref_shape) == sh: return batch
        crop_shape, res_shape = sh, sample_resize(sh, ref_shape, eps=1)
    else:
        ref = tuple(ref_shape) if ref_shape else sh
        crop_shape, ratio = sample_crop(sh, crop_min, crop_max)
        m.setdefault('augs', []).append(f'{list(sh)} -> {crop_shape} -> {RATIO2LABEL[ratio]}')
        res_shape = sample_resize(crop_shape, ref, eps=0.8)
    m.setdefault('augs', []).append(str(res_shape))
    if 'depth_hints' in y:   # aspect_ratio.py:146-149
        raise RuntimeError('Geometric augmentation should not be combined with depth hints... Interpolating depth is not well defined.')
    keys = [(x, 'imgs'), (y, 'imgs'), (x, 'supp_imgs'), (y, 'supp_imgs')] + ([(y, 'depth')] if 'depth' in y else [])
    outs, K = resample([d[k] for d, k in keys], crop_shape, tuple(res_shape), y.get('K'))
    for (d, k), o in zip(keys, outs): d[k] = o
    if K is not None: y['K'] = K
    return x, y, m
