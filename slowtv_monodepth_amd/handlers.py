"""Multi-scale loss handlers (reference: `src/core/handlers.py`): the level at which the fused kernels plug in.

`image_recon` and `disp_smooth` keep the reference's signatures and return values, but instead of expanding every
tensor to (n, S*b, ...) and chaining ViewSynth -> ReconstructionLoss (handlers.py:45-62), they hand the un-expanded
tensors to one fused HIP forward (and, through autograd, one fused backward).

The other ViewSynth users (`feat_recon`, `autoenc_recon`, `stereo_const`, `depth_regr`) keep the
reference's structure and run on the un-fused HIP operators (`view_synth`, `photo_error`, `recon_reduce`,
`regression_loss`), which take any channel count.
"""
from __future__ import annotations

from collections.abc import Mapping

import torch

from . import functional as F

__all__ = ['image_recon', 'disp_smooth', 'feat_recon', 'autoenc_recon', 'stereo_const', 'depth_regr', 'ScaleDict', 'LazyDepths']


class ScaleDict(dict):
    """{scale: (b,1,h,w)} whose values are views of ONE scale-major tensor `.stacked` (S,b,1,h,w) — what the K0 kernel
    writes — so the handler can pass it on without the `torch.stack` copy of handlers.py:48."""
    stacked: torch.Tensor

    @classmethod
    def from_stack(cls, keys, stacked: torch.Tensor) -> 'ScaleDict':
        out = cls({k: stacked[i] for i, k in enumerate(keys)})
        out.stacked = stacked
        return out


class LazyDepths(Mapping):
    """`fwd['depth_up']` = {scale: (b,1,h,w)} that has not been computed yet: it remembers the network's disparities and the
    to-depth parameters.  The fused `image_recon` handler consumes it WITHOUT a K0 launch (the fused kernel up-samples and
    converts the rows it is about to warp, and hands back the depth stack it wrote); any other access (`depths[s]`, `.stacked`,
    iteration over values) materialises it through the K0 kernel, so every consumer of `depth_up` sees the usual tensors."""

    def __init__(self, keys, disps, size, min_depth, max_depth):
        self.keys_, self.disps, self.size, self.min_depth, self.max_depth = list(keys), list(disps), tuple(size), min_depth, max_depth
        self._stack = None

    @property
    def pending(self) -> bool: return self._stack is None

    def adopt(self, stacked: torch.Tensor) -> None: self._stack = stacked

    @property
    def stacked(self) -> torch.Tensor:
        if self._stack is None: self._stack, _ = F.disp_to_depth(self.disps, self.size, self.min_depth, self.max_depth)
        return self._stack

    def __getitem__(self, k): return self.stacked[self.keys_.index(k)]
    def __iter__(self): return iter(self.keys_)
    def __len__(self): return len(self.keys_)


def image_recon(crit, synth, depths: dict, masks, imgs: torch.Tensor, supp_imgs: torch.Tensor, Ts: torch.Tensor, Ks: torch.Tensor,
                *, K_inv: torch.Tensor | None = None, noise: torch.Tensor | None = None, want_warp: bool = True, prepared=None):
    """Reconstruction loss over all scales and supports (src/core/handlers.py:14-67).

    :param crit: `ReconstructionLoss` (its loss_name / use_min / use_automask select the kernel flags).
    :param synth: `ViewSynth` for the image size (kept for signature parity; the fused kernel does its own projection).
    :param depths: {s: (b,1,h,w)} up-sampled depth per scale (a `ScaleDict` avoids one copy).
    :param masks: None, or {s: (b,n,h,w)} up-sampled predictive weighting masks for a criterion built with `mask_name` — then the
        un-fused operators run (warp, per-support errors, masked reduction), as the reference does (src/core/handlers.py:47, 62).
    :param imgs: (b,3,h,w) target; supp_imgs: (n,b,3,h,w); Ts: (n,b,4,4); Ks: (b,4,4).
    :param K_inv: optional (b,4,4) inverse intrinsics when the caller already has them (`functional.intrinsics`).
    :param noise: optional (S*b,1,h,w) replacement for the reference's `randn_like` tie-break draw (reconstruction.py:72).
    :param prepared: optional `functional.PreparedFrames` for (imgs, supp_imgs): the frame-only half of the forward, launched ahead
        of time (the trainer does it on a side stream under the networks); ignored if it was built for something else.
    :return: (loss, {'supp_imgs_warp': (n,b,3,h,w) of scale 0 [, 'automask': (b,1,h,w) bool of scale 0]})
    """
    if synth is not None and tuple(synth.shape) != tuple(imgs.shape[-2:]):
        raise ValueError(f'ViewSynth built for {synth.shape}, images are {tuple(imgs.shape[-2:])}')
    if imgs.shape[1] != 3 or crit.loss_name == 'l2' or masks is not None or getattr(crit, 'mask_name', None):   # features / Euclidean error / predictive masks: un-fused operators
        return _image_recon_generic(crit, depths, imgs, supp_imgs, Ts, Ks, K_inv, noise, want_warp, masks)
    flags = F.recon_flags(crit.loss_name, crit.use_min, crit.use_automask)
    if isinstance(depths, LazyDepths) and depths.pending:   # K0 fused: no up-sampling launch, the kernel writes the depth stack
        if prepared is not None and not prepared.matches(imgs, supp_imgs, flags, [d.shape[-2] for d in depths.disps], [d.shape[-1] for d in depths.disps]): prepared = None
        loss, err, sel, warp0, depth_up = F.image_recon_fused_disp(depths.disps, imgs, supp_imgs, Ts, Ks, K_inv, flags=flags, min_depth=depths.min_depth,
                                                                   max_depth=depths.max_depth, noise=noise, seed=crit.next_seed(), want_warp=want_warp, want_err=False,
                                                                   prepared=prepared)
        depths.adopt(depth_up)
    else:
        stacked = getattr(depths, 'stacked', None)
        if stacked is None: stacked = torch.stack(list(depths.values()))
        if prepared is not None and not prepared.matches(imgs, supp_imgs, flags, None, None): prepared = None
        loss, err, sel, warp0 = F.image_recon_fused(stacked, imgs, supp_imgs, Ts, Ks, K_inv, flags=flags, noise=noise, seed=crit.next_seed(),
                                                    want_warp=want_warp, want_err=False, prepared=prepared)
    ld = {}
    if crit.use_automask: ld['automask'] = sel[0] != 255
    if want_warp: ld['supp_imgs_warp'] = warp0
    return loss, ld


def _expand_views(depths: dict, imgs, supp_imgs, Ts, Ks, K_inv):
    """The (n, S*b, ...) expansion of handlers.py:45-56 (views only; the operators read them through `.contiguous()`)."""
    n, S, b = supp_imgs.shape[0], len(depths), imgs.shape[0]
    dep = torch.stack(list(depths.values())).flatten(0, 1)                                   # (S*b,1,h,w)
    tgt = imgs[None].expand(S, *imgs.shape).flatten(0, 1)                                    # (S*b,c,h,w)
    src = supp_imgs[:, None].expand(n, S, *supp_imgs.shape[1:]).flatten(1, 2)                # (n,S*b,c,h,w)
    T = Ts[:, None].expand(n, S, b, 4, 4).flatten(0, 2)
    K = Ks[None, None].expand(n, S, b, 4, 4).flatten(0, 2)
    Ki = K_inv[None, None].expand(n, S, b, 4, 4).flatten(0, 2)
    return n, S, b, dep, tgt, src, T, K, Ki


def _image_recon_generic(crit, depths, imgs, supp_imgs, Ts, Ks, K_inv, noise, want_warp, masks=None):
    if K_inv is None: K_inv = torch.linalg.inv(Ks) if Ks.requires_grad else F.inv_intrinsics(Ks)
    n, S, b, dep, tgt, src, T, K, Ki = _expand_views(depths, imgs, supp_imgs, Ts, Ks, K_inv)
    warp = F.view_synth(src.flatten(0, 1), dep[None].expand(n, *dep.shape).flatten(0, 1), T, K, Ki)[0].unflatten(0, (n, S*b))
    mask = torch.stack(list(masks.values())).flatten(0, 1) if masks is not None else None    # (S*b,n,h,w), handlers.py:47
    loss, ld = crit(warp, tgt, source=src, mask=mask, noise=noise)
    out = {}
    if crit.use_automask: out['automask'] = ld['automask'].unflatten(0, (S, b))[0]
    if want_warp: out['supp_imgs_warp'] = warp.unflatten(1, (S, b))[:, 0]
    return loss, out


def feat_recon(crit, synth, depths: dict, masks, feats, supp_feats, Ts: torch.Tensor, Ks: torch.Tensor, *, noise=None):
    """Feature-metric reconstruction loss on the finest depth map only (src/core/handlers.py:70-119).

    :param feats: (b,c,hf,wf) target encoder features, or the encoder's list of multi-scale features (the 1/4-scale
        entry `[-4]` is used, as in the reference); supp_feats: (n,b,c,hf,wf) or the matching list.
    :return: (loss, {'supp_feats_warp': (n,b,c,h,w)})
    """
    if isinstance(feats, (list, tuple)): feats, supp_feats = feats[-4], supp_feats[-4]
    size = tuple(depths[0].shape[-2:])
    with torch.no_grad():   # features are detached and resized to the depth map (handlers.py:102-110)
        n = supp_feats.shape[0]
        feats = torch.nn.functional.interpolate(feats.detach().float(), size=size, mode='bilinear', align_corners=False)
        supp_feats = torch.nn.functional.interpolate(supp_feats.detach().float().flatten(0, 1), size=size, mode='bilinear',
                                                     align_corners=False).unflatten(0, (n, -1))
    loss, ld = image_recon(crit, synth, {0: depths[0]}, ({0: masks[0]} if masks is not None else None), feats, supp_feats, Ts, Ks, noise=noise, want_warp=True)
    return loss, {'supp_feats_warp': ld['supp_imgs_warp']}


def autoenc_recon(crit, preds: dict, targets: torch.Tensor, supp_preds: dict, supp_targets: torch.Tensor):
    """Autoencoder reconstruction of the target and support frames at every scale (src/core/handlers.py:122-149).
    preds {s: (b,3,h,w)}, supp_preds {s: (n,b,3,h,w)}, supp_targets (n,b,3,h,w) -> (loss, {})."""
    S = len(preds)
    p = torch.stack(list(preds.values())).flatten(0, 1)
    sp = torch.stack(list(supp_preds.values())).flatten(0, 2)
    t = targets[None].expand(S, *targets.shape).flatten(0, 1)
    st = supp_targets[None].expand(S, *supp_targets.shape).flatten(0, 2)
    loss, _ = crit(torch.cat((p, sp)), torch.cat((t, st)))
    return loss, {}


def stereo_const(crit, synth, disps: dict, depths: dict, disps_stereo: dict, depths_stereo: dict, T_stereo: torch.Tensor, K: torch.Tensor):
    """Virtual-stereo consistency (src/core/handlers.py:152-198): each view's disparity warped into the other one.
    :return: (loss, {'disps_warp', 'stereo_disps_warp': (b,1,h,w) of scale 0})"""
    S = len(disps)
    d = torch.stack(list(disps.values())).flatten(0, 1); dep = torch.stack(list(depths.values())).flatten(0, 1)
    ds = torch.stack(list(disps_stereo.values())).flatten(0, 1); deps = torch.stack(list(depths_stereo.values())).flatten(0, 1)
    N = T_stereo.shape[0]
    T = T_stereo.float()[None].expand(S, N, 4, 4).flatten(0, 1)
    T_all = torch.cat((T, torch.linalg.inv(T)))
    K_all = K.float()[None, None].expand(2, S, *K.shape).flatten(0, 2)
    Ki_all = (torch.linalg.inv(K) if K.requires_grad else F.inv_intrinsics(K.float()))[None, None].expand(2, S, *K.shape).flatten(0, 2)
    all_disps = torch.cat((ds, d))
    warp = F.view_synth(all_disps, torch.cat((dep, deps)), T_all, K_all, Ki_all)[0]
    loss, _ = crit(all_disps, warp)
    sw, dw = warp.chunk(2)
    return loss, {'disps_warp': dw.unflatten(0, (S, -1))[0], 'stereo_disps_warp': sw.unflatten(0, (S, -1))[0]}


def depth_regr(crit, synth, photo, depths: dict, targets: torch.Tensor, imgs: torch.Tensor, supp_imgs: torch.Tensor,
               Ts: torch.Tensor, Ks: torch.Tensor):
    """Proxy-depth regression with the Depth-Hints automask (src/core/handlers.py:201-259).

    :param photo: `ReconstructionLoss.compute_photo` of the img_recon criterion (src/core/trainer.py:430).
    :param targets: (b,1,h,w) proxy depth, 0 where missing.
    :return: (loss, {'mask_regr': (b,1,h,w) bool of scale 0 [, 'automask_hints' is folded into it]})
    """
    S, n, b = len(depths), supp_imgs.shape[0], imgs.shape[0]
    dep = torch.stack(list(depths.values())).flatten(0, 1)
    tg = targets[None].expand(S, *targets.shape).flatten(0, 1).contiguous()
    masks = tg > 0
    if crit.use_automask:
        with torch.no_grad():   # the mask is a comparison: nothing differentiable flows through either warp
            K_inv = F.inv_intrinsics(Ks.detach().float())
            _, _, _, _, im, src, T, K, Ki = _expand_views(depths, imgs, supp_imgs, Ts.detach(), Ks.detach(), K_inv)
            ex = lambda z: z[None].expand(n, *z.shape).flatten(0, 1)
            hints_warp = F.view_synth(src.flatten(0, 1), ex(tg), T, K, Ki)[0].unflatten(0, (n, S*b))
            pred_warp = F.view_synth(src.flatten(0, 1), ex(dep.detach()), T, K, Ki)[0].unflatten(0, (n, S*b))
            masks = masks & (photo(pred_warp, im) > photo(hints_warp, im))
    loss, ld = crit(dep, tg, masks)
    return loss, {'mask_regr': ld['mask_regr'].unflatten(0, (S, -1))[0]}


def disp_smooth(crit, disps: dict, imgs: torch.Tensor, *, want_aux: bool = True, prepared=None):
    """Smoothness over the raw (not up-sampled) multi-scale disparities: mean_s(loss_s / 2^s) (src/core/handlers.py:262-281).

    :param want_aux: also produce the two logging maps (one extra small launch); the training loop turns this off.
    :param prepared: optional `functional.PreparedFrames` carrying the edge weights of `imgs` for this pyramid (frame-only, launched ahead).
    :return: (loss, {'disp_grad', 'image_grad'} of scale 0)
    """
    if getattr(crit, 'use_blur', False): loss, dg, ig = F.disp_smooth_blurred(disps, imgs, use_edges=crit.use_edges, want_aux=want_aux)
    else: loss, dg, ig = F.disp_smooth_fused(disps, imgs, use_edges=crit.use_edges, want_aux=want_aux, use_laplacian=getattr(crit, 'use_laplacian', False), prepared=prepared)
    return loss, ({'disp_grad': dg, 'image_grad': ig} if want_aux and dg is not None else {})
