"""Multi-scale loss handlers (reference: `src/core/handlers.py`): the level at which the fused kernels plug in.

`image_recon` and `disp_smooth` keep the reference's signatures and return values, but instead of expanding every
tensor to (n, S*b, ...) and chaining ViewSynth -> ReconstructionLoss (handlers.py:45-62), they hand the un-expanded
tensors to one fused HIP forward (and, through autograd, one fused backward).
"""
from __future__ import annotations

import torch

from . import functional as F

__all__ = ['image_recon', 'disp_smooth', 'ScaleDict']


class ScaleDict(dict):
    """{scale: (b,1,h,w)} whose values are views of ONE scale-major tensor `.stacked` (S,b,1,h,w) — what the K0 kernel
    writes — so the handler can pass it on without the `torch.stack` copy of handlers.py:48."""
    stacked: torch.Tensor

    @classmethod
    def from_stack(cls, keys, stacked: torch.Tensor) -> 'ScaleDict':
        out = cls({k: stacked[i] for i, k in enumerate(keys)})
        out.stacked = stacked
        return out


def image_recon(crit, synth, depths: dict, masks, imgs: torch.Tensor, supp_imgs: torch.Tensor, Ts: torch.Tensor, Ks: torch.Tensor,
                *, K_inv: torch.Tensor | None = None, noise: torch.Tensor | None = None, want_warp: bool = True):
    """Reconstruction loss over all scales and supports (src/core/handlers.py:14-67).

    :param crit: `ReconstructionLoss` (its loss_name / use_min / use_automask select the kernel flags).
    :param synth: `ViewSynth` for the image size (kept for signature parity; the fused kernel does its own projection).
    :param depths: {s: (b,1,h,w)} up-sampled depth per scale (a `ScaleDict` avoids one copy).
    :param masks: must be None (predictive masks are outside the accelerated path).
    :param imgs: (b,3,h,w) target; supp_imgs: (n,b,3,h,w); Ts: (n,b,4,4); Ks: (b,4,4).
    :param K_inv: optional (b,4,4) inverse intrinsics when the caller already has them (`functional.intrinsics`).
    :param noise: optional (S*b,1,h,w) replacement for the reference's `randn_like` tie-break draw (reconstruction.py:72).
    :return: (loss, {'supp_imgs_warp': (n,b,3,h,w) of scale 0 [, 'automask': (b,1,h,w) bool of scale 0]})
    """
    if masks is not None: raise NotImplementedError('predictive masks are outside the accelerated path')
    if synth is not None and tuple(synth.shape) != tuple(imgs.shape[-2:]):
        raise ValueError(f'ViewSynth built for {synth.shape}, images are {tuple(imgs.shape[-2:])}')
    stacked = getattr(depths, 'stacked', None)
    if stacked is None: stacked = torch.stack(list(depths.values()))
    flags = F.recon_flags(crit.loss_name, crit.use_min, crit.use_automask)
    loss, err, sel, warp0 = F.image_recon_fused(stacked, imgs, supp_imgs, Ts, Ks, K_inv, flags=flags, noise=noise, seed=crit.next_seed(),
                                                want_warp=want_warp)
    ld = {}
    if crit.use_automask: ld['automask'] = sel[0] != 255
    if want_warp: ld['supp_imgs_warp'] = warp0
    return loss, ld


def disp_smooth(crit, disps: dict, imgs: torch.Tensor, *, want_aux: bool = True):
    """Smoothness over the raw (not up-sampled) multi-scale disparities: mean_s(loss_s / 2^s) (src/core/handlers.py:262-281).

    :param want_aux: also produce the two logging maps (one extra small launch); the training loop turns this off.
    :return: (loss, {'disp_grad', 'image_grad'} of scale 0)
    """
    loss, dg, ig = F.disp_smooth_fused(disps, imgs, use_edges=crit.use_edges, want_aux=want_aux)
    return loss, ({'disp_grad': dg, 'image_grad': ig} if want_aux and dg is not None else {})
