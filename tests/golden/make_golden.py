#!/usr/bin/env python3
"""Generate golden vectors for the view-synthesis loss hot path by IMPORTING the reference.

Runs ONLY in the build container (needs /root/reference); never on the GPU box.  The reference has no
tests of its own for this path, so these fixtures are the parity pins: seeded inputs plus
the outputs/gradients the reference's own code produces for them.

    PYTHONPATH=/root/reference python tests/golden/make_golden.py

Reference entry points driven here (all paths relative to /root/reference):
  * src/core/trainer.py:280-348  MonoDepthModule.forward_postprocess  (upsample + to_depth + Ts stacking)
  * src/core/trainer.py:350-472  MonoDepthModule.forward_loss         (handlers.image_recon / disp_smooth)
  * src/tools/geometry.py:353-391 ViewSynth.forward, :181-209 T_from_AAt, :62-90 to_scaled/to_inv
  * src/losses/photometric.py:54-88 PhotoError, src/losses/reconstruction.py:79-96 compute_photo
  * src/regularizers/smooth.py:51-97 SmoothReg
  * src/networks/pose.py:60-73 PoseNet.build_K (+ geometry.py:249-263 resize_K)
  * src/core/handlers.py:70-259 feat_recon / autoenc_recon / stereo_const / depth_regr, src/losses/regression.py:40-75

The fixture files hold data only (inputs + expected outputs); no reference source text is stored.
"""
import importlib.abc
import importlib.machinery
import math
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = Path('/root/reference')
OUT = Path(__file__).resolve().parent


# --------------------------------------------------------------------------------------------------
# Import shim: the reference imports several third-party packages at module scope that are absent in
# this image and never executed on the loss path.  Hand out empty placeholder modules for them.
# --------------------------------------------------------------------------------------------------
_ABSENT = ('cv2', 'skimage', 'kornia', 'timm', 'torchmetrics', 'pytorch_lightning', 'wandb', 'lmdb', 'h5py',
           'torchvision', 'lightning', 'tensorboard', 'albumentations', 'mmcv')


class _Placeholder(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith('__'): raise AttributeError(name)
        cls = type(name, (), {'__init__': lambda self, *a, **k: None,
                              '__class_getitem__': classmethod(lambda c, i: c)})
        setattr(self, name, cls)
        return cls


class _AbsentFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split('.')[0] in _ABSENT:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)

    def create_module(self, spec): return _Placeholder(spec.name)

    def exec_module(self, module): pass


def import_reference():
    if not REF.is_dir(): raise SystemExit('reference checkout not found; fixtures can only be regenerated in the build container')
    sys.meta_path.insert(0, _AbsentFinder())
    sys.path.insert(0, str(REF))
    sys.path.insert(0, str(OUT))       # exact_inputs.py
    import src  # noqa: F401
    from src.core import handlers
    from src.core.trainer import MonoDepthModule
    from src.losses import PhotoError, ReconstructionLoss
    from src.losses.photometric import DenseL1Error, DenseL2Error
    from src.losses.regression import RegressionLoss
    from src.networks.pose import PoseNet
    from src.regularizers import SmoothReg
    from src.tools import ViewSynth, T_from_AAt, resize_K, to_inv, to_scaled
    from src.utils import MultiLevelTimer
    return types.SimpleNamespace(**locals())


# --------------------------------------------------------------------------------------------------
# Synthetic inputs (also restated in slowtv_monodepth_amd/synthetic.py; kept independent here)
# --------------------------------------------------------------------------------------------------
def texture(g, b, h, w, shift=(0.0, 0.0)):
    """Smooth multi-sinusoid RGB texture in [0, 1] + 5% uniform noise, optionally shifted by (dx, dy) pixels."""
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
    xs, ys = xs + shift[0], ys + shift[1]
    img = torch.zeros(b, 3, h, w)
    for _ in range(6):
        fx, fy = (torch.rand(2, generator=g)*0.35 + 0.02).tolist()
        ph = (torch.rand(b, 3, 1, 1, generator=g)*2*math.pi)
        amp = torch.rand(b, 3, 1, 1, generator=g)
        img += amp*torch.sin(fx*xs + fy*ys + ph)
    img = (img - img.amin(dim=(2, 3), keepdim=True))/(img.amax(dim=(2, 3), keepdim=True) - img.amin(dim=(2, 3), keepdim=True))
    img = 0.95*img + 0.05*torch.rand(b, 3, h, w, generator=g)
    return img.clamp(0, 1)


def make_inputs(seed, b, h, w, n, scales, pose_scale=0.01, learn_K=False, flat_patch=False):
    g = torch.Generator().manual_seed(seed)
    st = g.get_state()
    imgs = texture(g, b, h, w)
    supp = []
    for i in range(n):
        g.set_state(st)  # same texture, shifted
        dx = float(2 + i)*(1 if i % 2 else -1)
        s_img = texture(g, b, h, w, shift=(dx, 0.5*dx))
        supp.append(s_img)
    supp = torch.stack(supp)
    if flat_patch:  # saturated region -> exact ties between warped and static error (automask noise matters)
        imgs[:, :, : h//3, : w//3] = 1.0
        supp[:, :, :, : h//3 + 2, : w//3 + 2] = 1.0
    g.manual_seed(seed + 1)
    disp = {s: (0.05 + 0.9*torch.rand(b, 1, max(h >> s, 1), max(w >> s, 1), generator=g)) for s in scales}
    aa = pose_scale*torch.randn(n, b, 3, generator=g)
    t = pose_scale*10*torch.randn(n, b, 3, generator=g)
    K = torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=torch.float32)
    K = K[None].repeat(b, 1, 1)
    out = dict(imgs=imgs, supp_imgs=supp, disp=disp, aa=aa, t=t, K=K)
    if learn_K:
        out['fs'] = torch.tensor([0.58, 1.92])[None].repeat(b, 1)*(1 + 0.1*torch.randn(b, 2, generator=g))
        out['cs'] = torch.tensor([0.5, 0.5])[None].repeat(b, 1) + 0.05*torch.randn(b, 2, generator=g)
    return out


# --------------------------------------------------------------------------------------------------
def make_inputs_compact(seed, b, h, w, n, scales, pose_scale, learn_K):
    """BASELINE-resolution cases: images / disparities / tie-break noise from `exact_inputs.make_inputs_exact` (regenerated by
    the tests, not stored); the few pose / intrinsics numbers are drawn here and stored."""
    from exact_inputs import frame_shifts, make_inputs_exact
    out = make_inputs_exact(seed, b, h, w, n, scales, learn_K=learn_K)
    g = torch.Generator().manual_seed(seed + 1)
    # a camera motion that roughly explains the frames' shifts at depth ~1.2 (so that min-reprojection and the automask are
    # both live: a third to a half of the pixels prefer a warped frame) + a small random rotation / translation
    sh = torch.tensor(frame_shifts(n), dtype=torch.float32)                                   # (n, 2) pixels
    t0 = torch.stack([-sh[:, 0]*1.2/(0.58*w), -sh[:, 1]*1.2/(1.92*h), torch.zeros(n)], -1)     # (n, 3)
    out['aa'] = 0.1*pose_scale*torch.randn(n, b, 3, generator=g)
    out['t'] = t0[:, None] + 0.1*pose_scale*torch.randn(n, b, 3, generator=g)
    if learn_K:
        out['fs'] = torch.tensor([0.58, 1.92])[None].repeat(b, 1)*(1 + 0.1*torch.randn(b, 2, generator=g))
        out['cs'] = torch.tensor([0.5, 0.5])[None].repeat(b, 1) + 0.05*torch.randn(b, 2, generator=g)
    return out


def run_trainer_case(R, name, *, seed, b, h, w, n, scales, supp_idxs, loss_kw, smooth_kw, min_depth, max_depth,
                     always_fwd_pose=True, learn_K=False, pose_scale=0.01, flat_patch=False, w_smooth=0.001, compact=0):
    """`compact` = sampling stride of the stored full-resolution maps (0: the small-case layout with inputs and whole maps)."""
    if compact: inp = make_inputs_compact(seed, b, h, w, n, scales, pose_scale, learn_K)
    else: inp = make_inputs(seed, b, h, w, n, scales, pose_scale=pose_scale, learn_K=learn_K, flat_patch=flat_patch)
    leaves = {f'disp_{s}': d.clone().requires_grad_(True) for s, d in inp['disp'].items()}
    aa = inp['aa'].clone().requires_grad_(True)
    t = inp['t'].clone().requires_grad_(True)
    leaves.update(aa=aa, t=t)

    Ts = R.T_from_AAt(aa=aa.flatten(0, 1), t=t.flatten(0, 1)).unflatten(0, (n, b))
    fwd = {'disp': {s: leaves[f'disp_{s}'] for s in scales}}
    for i, T in zip(supp_idxs, Ts):
        fwd[f'T_{i}'] = T.inverse() if (always_fwd_pose and i < 0) else T

    y = {'imgs': inp['imgs'], 'supp_imgs': inp['supp_imgs'], 'K': inp['K']}
    if learn_K:
        fs = inp['fs'].clone().requires_grad_(True); cs = inp['cs'].clone().requires_grad_(True)
        leaves.update(fs=fs, cs=cs)
        fwd['K'] = R.resize_K(R.PoseNet.build_K(fs, cs), (h, w))
    x = {'imgs': inp['imgs'], 'supp_idxs': torch.tensor(supp_idxs)}

    crit_recon = R.ReconstructionLoss(**loss_kw)
    losses = {'img_recon': crit_recon}
    weights = {'img_recon': torch.tensor(1.0)}
    if smooth_kw is not None:
        losses['disp_smooth'] = R.SmoothReg(**smooth_kw)
        weights['disp_smooth'] = torch.tensor(w_smooth)

    if min_depth or max_depth: to_depth = lambda d: R.to_scaled(d, min_depth, max_depth)[1]
    else: to_depth = R.to_inv

    ns = types.SimpleNamespace(losses=losses, weights=weights, synth=R.ViewSynth((h, w)),
                               timer=R.MultiLevelTimer(name='golden'), to_depth=to_depth)

    # Record the tie-break noise drawn inside apply_automask (reconstruction.py:72).
    noise_log = []
    orig = torch.randn_like

    def rec(tensor, *a, **k):
        if compact:      # the regenerable stand-in for the tie-break draw (exact_inputs.py)
            assert tensor.shape == inp['noise'].shape, (tensor.shape, inp['noise'].shape)
            out = inp['noise'].to(tensor)
        else: out = orig(tensor, *a, **k)
        noise_log.append(out.clone()); return out

    torch.manual_seed(seed + 7)
    torch.randn_like = rec
    try:
        fwd = R.MonoDepthModule.forward_postprocess(ns, fwd, x, y)
        loss, ld = R.MonoDepthModule.forward_loss(ns, fwd, x, y)
    finally:
        torch.randn_like = orig
    loss.backward()

    rec_ = {'meta_b': b, 'meta_h': h, 'meta_w': w, 'meta_n': n, 'meta_scales': np.array(scales),
            'meta_supp_idxs': np.array(supp_idxs), 'meta_always_fwd_pose': int(always_fwd_pose),
            'meta_min_depth': float(min_depth or 0), 'meta_max_depth': float(max_depth or 0),
            'meta_learn_K': int(learn_K), 'meta_w_smooth': float(w_smooth if smooth_kw is not None else -1),
            'meta_loss_name': str(loss_kw.get('loss_name', 'ssim')), 'meta_use_min': int(loss_kw.get('use_min', False)),
            'meta_use_automask': int(loss_kw.get('use_automask', False)),
            'meta_use_edges': int((smooth_kw or {}).get('use_edges', False))}
    if compact:
        return save_compact(R, name, rec_, inp, fwd, loss, ld, leaves, ns, crit_recon, seed, compact, noise_log)
    rec_.update({f'in_{k}': v for k, v in inp.items() if k != 'disp'})
    rec_.update({f'in_disp_{s}': d for s, d in inp['disp'].items()})
    if noise_log: rec_['in_noise'] = noise_log[0]
    rec_.update({f'out_depth_up_{s}': v for s, v in fwd['depth_up'].items()})
    rec_.update({f'out_disp_up_{s}': v for s, v in fwd['disp_up'].items()})
    rec_['out_Ts'] = fwd['Ts']
    if learn_K: rec_['out_K'] = fwd['K']
    rec_['out_loss'] = loss
    for k, v in ld.items(): rec_[f'out_{k}'] = v
    for k, v in leaves.items(): rec_[f'grad_{k}'] = v.grad

    # Finer-grained intermediates from the class-level reference modules (all scales).
    with torch.no_grad():
        S = len(scales)
        depths = torch.stack([fwd['depth_up'][s] for s in scales]).flatten(0, 1)  # (S*b,1,h,w)
        Ks = fwd.get('K', y['K'])
        warp_all, errs = [], []
        for i in range(n):
            wi = ns.synth(input=inp['supp_imgs'][i].repeat(S, 1, 1, 1), depth=depths,
                          T=fwd['Ts'][i].repeat(S, 1, 1), K=Ks.repeat(S, 1, 1))
            warp_all.append(wi)
        rec_['mid_warp'] = torch.stack([w_[0] for w_ in warp_all])  # (n,S*b,3,h,w)
        rec_['mid_depth_warp'] = torch.stack([w_[1] for w_ in warp_all])
        rec_['mid_mask_valid'] = torch.stack([w_[2] for w_ in warp_all])
        rec_['mid_err_warp'] = crit_recon.compute_photo(rec_['mid_warp'], inp['imgs'].repeat(S, 1, 1, 1))  # (S*b,1,h,w)
        rec_['mid_err_static'] = crit_recon.compute_photo(inp['supp_imgs'], inp['imgs'])  # (b,1,h,w), no noise
    save(name, rec_)
    print(f'{name}: loss={loss.item():.8f}  ' + ' '.join(f'{k}={v.item():.6f}' for k, v in ld.items() if v.ndim == 0))


def save_compact(R, name, rec_, inp, fwd, loss, ld, leaves, ns, crit_recon, seed, stride, noise_log):
    """BASELINE-resolution layout: nothing that `exact_inputs.make_inputs_exact(seed, ...)` regenerates is stored (only its bit
    checksums); scalars, matrices and every gradient whole; boolean maps bit-packed; full-resolution float maps sampled every
    `stride` pixels in both directions (keys `outs_*` / `mids_*`)."""
    from exact_inputs import bit_checksum
    assert len(noise_log) == 1 and torch.equal(noise_log[0], inp['noise'])
    scales = list(inp['disp'])
    S, n = len(scales), inp['supp_imgs'].shape[0]
    rec_.update(meta_compact=1, meta_seed=seed, meta_stride=stride)
    for k in ('aa', 't', 'fs', 'cs', 'K'):
        if k in inp: rec_[f'in_{k}'] = inp[k]
    for k in ('imgs', 'supp_imgs', 'noise', 'K'): rec_[f'chk_{k}'] = np.int64(bit_checksum(inp[k]))
    for s, d in inp['disp'].items(): rec_[f'chk_disp_{s}'] = np.int64(bit_checksum(d))
    sub = lambda v: v[..., ::stride, ::stride].contiguous()
    pack = lambda v: np.packbits(v.detach().cpu().numpy().astype(bool).reshape(-1))
    rec_['out_Ts'] = fwd['Ts']
    if 'K' in fwd: rec_['out_K'] = fwd['K']
    rec_['out_loss'] = loss
    for k, v in ld.items():
        if v.ndim == 0: rec_[f'out_{k}'] = v
        elif v.dtype == torch.bool: rec_[f'bits_{k}'] = pack(v); rec_[f'shape_{k}'] = np.array(v.shape)
        else: rec_[f'outs_{k}'] = sub(v)
    for s in scales:
        rec_[f'outs_depth_up_{s}'] = sub(fwd['depth_up'][s]); rec_[f'outs_disp_up_{s}'] = sub(fwd['disp_up'][s])
    for k, v in leaves.items(): rec_[f'grad_{k}'] = v.grad
    with torch.no_grad():   # class-level reference modules on every scale: the error maps and decisions the trainer-level call does not return
        depths = torch.stack([fwd['depth_up'][s] for s in scales]).flatten(0, 1)
        Ks = fwd.get('K', inp['K'])
        warp = torch.stack([ns.synth(input=inp['supp_imgs'][i].repeat(S, 1, 1, 1), depth=depths, T=fwd['Ts'][i].repeat(S, 1, 1),
                                     K=Ks.repeat(S, 1, 1))[0] for i in range(n)])                       # (n,S*b,3,h,w)
        tgt = inp['imgs'].repeat(S, 1, 1, 1)
        err_warp = crit_recon.compute_photo(warp, tgt)                                                      # (S*b,1,h,w)
        rec_['mids_err_warp'] = sub(err_warp)
        rec_['mid_err_warp_mean'] = err_warp.double().mean()
        if crit_recon.use_automask:
            real = torch.randn_like
            torch.randn_like = lambda x, **k: inp['noise'].to(x)
            try: err, am = crit_recon.apply_automask(err_warp, inp['supp_imgs'].repeat(1, S, 1, 1, 1), tgt)
            finally: torch.randn_like = real
            rec_['mids_err'] = sub(err)
            # which candidate every pixel's gradient goes to: arg-min over the supports of the reference's own per-support errors, 255 where
            # the identity error won (lets the tests compare GRADIENTS under the reference's routing when rounding decides a near-tie otherwise)
            per = crit_recon._photo(warp.flatten(0, 1), tgt[None].expand_as(warp).flatten(0, 1)).squeeze(1).unflatten(0, warp.shape[:2]).permute(1, 0, 2, 3)
            assert torch.equal(per.min(dim=1, keepdim=True)[0], err_warp)
            rec_['out_sel_all'] = torch.where(am, per.argmin(dim=1, keepdim=True), torch.full_like(am, 255, dtype=torch.long)).to(torch.uint8)
            assert torch.allclose(err.mean(), ld['loss_img_recon'] if 'loss_img_recon' in ld else err.mean(), rtol=1e-5)
    save(name, rec_)
    sz = (OUT/f'{name}.npz').stat().st_size
    am = ld.get('automask')
    print(f'{name}: loss={loss.item():.8f}  ' + ' '.join(f'{k}={v.item():.6f}' for k, v in ld.items() if v.ndim == 0)
          + (f'  automask share {am.float().mean().item():.3f}' if am is not None else '') + f'  [{sz/1e6:.2f} MB]')


def run_op_cases(R):
    """Stand-alone operator vectors: ViewSynth, PhotoError, SmoothReg, T_from_AAt, to_scaled/to_inv."""
    g = torch.Generator().manual_seed(1234)
    b, h, w, c = 3, 20, 28, 5
    feat = torch.rand(b, c, h, w, generator=g).requires_grad_(True)
    depth = (0.5 + 20*torch.rand(b, 1, h, w, generator=g)).requires_grad_(True)
    aa = (0.05*torch.randn(b, 3, generator=g)).requires_grad_(True)
    t = (0.5*torch.randn(b, 3, generator=g)); t[0, 2] = -3.0  # push some points behind the z clamp
    t.requires_grad_(True)
    K = torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]])[None].repeat(b, 1, 1)
    K = (K*(1 + 0.05*torch.rand(b, 4, 4, generator=g))).requires_grad_(True)
    T = R.T_from_AAt(aa, t)
    warp, dwarp, valid = R.ViewSynth((h, w))(feat, depth, T, K)
    gw = torch.randn(warp.shape, generator=g)
    gd = torch.randn(dwarp.shape, generator=g)
    ((warp*gw).sum() + (dwarp*gd).sum()).backward()
    save('op_view_synth', dict(in_input=feat, in_depth=depth, in_aa=aa, in_t=t, in_K=K, in_gw=gw, in_gd=gd,
                               out_T=T, out_warp=warp, out_depth_warp=dwarp, out_mask_valid=valid,
                               grad_input=feat.grad, grad_depth=depth.grad, grad_aa=aa.grad, grad_t=t.grad, grad_K=K.grad))

    pred = torch.rand(4, 3, 17, 23, generator=g).requires_grad_(True)
    tgt = (pred.detach() + 0.2*torch.randn(4, 3, 17, 23, generator=g)).clamp(0, 1)
    tgt[0] = pred.detach()[0]  # identical pair -> zero error, exercises the clamp at 0
    err = R.PhotoError()(pred, tgt)
    ge = torch.randn(err.shape, generator=g)
    (err*ge).sum().backward()
    save('op_photo_error', dict(in_pred=pred, in_target=tgt, in_ge=ge, out_err=err, grad_pred=pred.grad))

    for use_edges in (True, False):
        disp = (0.05 + 0.9*torch.rand(3, 1, 12, 20, generator=g)).requires_grad_(True)
        img = texture(g, 3, 12, 20)
        l, ld = R.SmoothReg(use_edges=use_edges)(disp, img)
        l.backward()
        save(f'op_smooth_edges{int(use_edges)}', dict(in_disp=disp, in_img=img, out_loss=l, out_disp_grad=ld['disp_grad'],
                                                       out_image_grad=ld['image_grad'], grad_disp=disp.grad))

    aa = torch.randn(6, 3, generator=g); aa[0] = 0; aa[1] *= 1e-4; aa[2] *= 3
    t = torch.randn(6, 3, generator=g)
    aa.requires_grad_(True); t.requires_grad_(True)
    T = R.T_from_AAt(aa, t)
    gT = torch.randn(T.shape, generator=g)
    (T*gT).sum().backward()
    save('op_T_from_AAt', dict(in_aa=aa, in_t=t, in_gT=gT, out_T=T, grad_aa=aa.grad, grad_t=t.grad))

    d = torch.rand(2, 1, 6, 9, generator=g); d[0, 0, 0, 0] = 0.0; d[0, 0, 0, 1] = 1.0
    sd, dep = R.to_scaled(d, 0.1, 100)
    sd2, dep2 = R.to_scaled(d, 0.01, None)
    save('op_to_depth', dict(in_disp=d, out_scaled_disp=sd, out_depth=dep, out_scaled_disp_nomax=sd2, out_depth_nomax=dep2,
                             out_inv=R.to_inv(d)))


def run_handler_cases(R):
    """The other ViewSynth users: generic-channel photometric errors, RegressionLoss, and the
    feat_recon / autoenc_recon / stereo_const / depth_regr handlers, with the reference's own autograd gradients."""
    g = torch.Generator().manual_seed(4321)
    # --- dense errors on C != 3
    for name, mod, c in (('op_photo_l2_c7', R.DenseL2Error(), 7), ('op_photo_l1_c4', R.DenseL1Error(), 4), ('op_photo_ssim_c5', R.PhotoError(), 5)):
        pred = torch.rand(3, c, 13, 19, generator=g).requires_grad_(True)
        tgt = (pred.detach() + 0.2*torch.randn(3, c, 13, 19, generator=g)).clamp(0, 1)
        tgt[0, :, :4] = pred.detach()[0, :, :4]   # identical block: |d| = 0 (sign / sqrt-clamp branches)
        err = mod(pred, tgt)
        ge = torch.randn(err.shape, generator=g)
        (err*ge).sum().backward()
        save(name, dict(in_pred=pred, in_target=tgt, in_ge=ge, out_err=err, grad_pred=pred.grad))

    # --- RegressionLoss
    for loss_name in ('l1', 'log_l1', 'berhu'):
        for invert in (False, True):
            pred = (0.5 + 10*torch.rand(2, 1, 11, 14, generator=g)).requires_grad_(True)
            tgt = 0.5 + 10*torch.rand(2, 1, 11, 14, generator=g)
            tgt[0, 0, :3] = 0.0                                             # invalid proxy depth
            mask = tgt > 0
            for tag, m in (('', None), ('_mask', mask)):
                pred.grad = None
                l, ld = R.RegressionLoss(loss_name=loss_name, invert=invert)(pred, tgt, m)
                l.backward()
                save(f'op_regr_{loss_name}{"_inv" if invert else ""}{tag}',
                     dict(in_pred=pred, in_target=tgt, in_mask=(mask if m is not None else torch.ones_like(mask)), meta_has_mask=int(m is not None),
                          out_loss=l, out_err=ld['err_regr'], grad_pred=pred.grad.clone()))

    def poses(n, b, scale=0.02):
        aa = (scale*torch.randn(n, b, 3, generator=g)).requires_grad_(True)
        t = (4*scale*torch.randn(n, b, 3, generator=g)).requires_grad_(True)
        return aa, t, R.T_from_AAt(aa.flatten(0, 1), t.flatten(0, 1)).unflatten(0, (n, b))

    def intr(b, h, w):
        return torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]])[None].repeat(b, 1, 1)

    # --- feat_recon: c=6 features at 1/4 resolution, l2 error, min + automask
    b, h, w, n, c = 2, 24, 40, 2, 6
    depth = (1 + 20*torch.rand(b, 1, h, w, generator=g)).requires_grad_(True)
    feats = torch.randn(b, c, h//4, w//4, generator=g); supp_feats = feats[None] + 0.3*torch.randn(n, b, c, h//4, w//4, generator=g)
    aa, t, Ts = poses(n, b); K = intr(b, h, w)
    crit = R.ReconstructionLoss(loss_name='l2', use_min=True, use_automask=True)
    noise = torch.randn(b, 1, h, w, generator=g)
    real_randn = torch.randn_like
    torch.randn_like = lambda x, **k: noise.to(x)            # pin the tie-break draw (reconstruction.py:72)
    try: l, ld = R.handlers.feat_recon(crit, R.ViewSynth((h, w)), {0: depth}, None, feats, supp_feats, Ts, K)
    finally: torch.randn_like = real_randn
    l.backward()
    save('hd_feat_recon', dict(in_depth=depth, in_feats=feats, in_supp_feats=supp_feats, in_aa=aa, in_t=t, in_K=K, in_noise=noise,
                               out_Ts=Ts, out_loss=l, out_supp_feats_warp=ld['supp_feats_warp'], grad_depth=depth.grad, grad_aa=aa.grad, grad_t=t.grad))

    # --- autoenc_recon: 2 scales, ssim, mean
    b, h, w, n = 2, 16, 24, 2
    tgt = texture(g, b, h, w); stg = torch.stack([texture(g, b, h, w, shift=(1.5*i, 0.5)) for i in range(n)])
    preds = {s: (tgt + 0.1*torch.randn(b, 3, h, w, generator=g)).clamp(0, 1).requires_grad_(True) for s in (0, 1)}
    spreds = {s: (stg + 0.1*torch.randn(n, b, 3, h, w, generator=g)).clamp(0, 1).requires_grad_(True) for s in (0, 1)}
    l, _ = R.handlers.autoenc_recon(R.ReconstructionLoss(loss_name='ssim', use_min=False), preds, tgt, spreds, stg)
    l.backward()
    save('hd_autoenc_recon', dict(in_targets=tgt, in_supp_targets=stg, out_loss=l,
                                  **{f'in_pred_{s}': v for s, v in preds.items()}, **{f'in_supp_pred_{s}': v for s, v in spreds.items()},
                                  **{f'grad_pred_{s}': v.grad for s, v in preds.items()}, **{f'grad_supp_pred_{s}': v.grad for s, v in spreds.items()}))

    # --- stereo_const: 2 scales, horizontal baseline
    b, h, w = 2, 20, 36
    mk = lambda: {s: (0.05 + 0.9*torch.rand(b, 1, h, w, generator=g)).requires_grad_(True) for s in (0, 1)}
    disps, disps_st = mk(), mk()
    depths = {s: R.to_scaled(d, 0.1, 100)[1] for s, d in disps.items()}
    depths_st = {s: R.to_scaled(d, 0.1, 100)[1] for s, d in disps_st.items()}
    T_st = torch.eye(4)[None].repeat(b, 1, 1); T_st[:, 0, 3] = torch.tensor([0.54, -0.54])
    K = intr(b, h, w)
    l, ld = R.handlers.stereo_const(R.RegressionLoss(loss_name='l1'), R.ViewSynth((h, w)), disps, depths, disps_st, depths_st, T_st, K)
    l.backward()
    save('hd_stereo_const', dict(in_T_stereo=T_st, in_K=K, out_loss=l, out_disps_warp=ld['disps_warp'], out_stereo_disps_warp=ld['stereo_disps_warp'],
                                 **{f'in_disp_{s}': v for s, v in disps.items()}, **{f'in_disp_stereo_{s}': v for s, v in disps_st.items()},
                                 **{f'grad_disp_{s}': v.grad for s, v in disps.items()}, **{f'grad_disp_stereo_{s}': v.grad for s, v in disps_st.items()}))

    # --- depth_regr: proxy depth with holes, Depth-Hints automask through the img_recon criterion's compute_photo
    b, h, w, n = 2, 20, 32, 2
    for tag, kw in (('berhu', dict(loss_name='berhu', invert=False, use_automask=True)),
                    ('log_l1_inv', dict(loss_name='log_l1', invert=True, use_automask=True)),
                    ('l1_noauto', dict(loss_name='l1', invert=False, use_automask=False))):
        imgs = texture(g, b, h, w); supp = torch.stack([texture(g, b, h, w, shift=(2.0*(i - 0.5), 0.3)) for i in range(n)])
        disps = {s: (0.05 + 0.9*torch.rand(b, 1, h, w, generator=g)).requires_grad_(True) for s in (0, 1)}
        depths = {s: R.to_scaled(d, 0.1, 100)[1] for s, d in disps.items()}
        hints = 0.5 + 30*torch.rand(b, 1, h, w, generator=g); hints[:, :, :4] = 0.0
        aa, t, Ts = poses(n, b); K = intr(b, h, w)
        photo = R.ReconstructionLoss(loss_name='ssim', use_min=True, use_automask=True).compute_photo
        l, ld = R.handlers.depth_regr(R.RegressionLoss(**kw), R.ViewSynth((h, w)), photo, depths, hints, imgs, supp, Ts.detach(), K)
        l.backward()
        save(f'hd_depth_regr_{tag}', dict(in_imgs=imgs, in_supp_imgs=supp, in_hints=hints, in_Ts=Ts, in_K=K, out_loss=l, out_mask_regr=ld['mask_regr'],
                                          meta_loss_name=kw['loss_name'], meta_invert=int(kw['invert']), meta_use_automask=int(kw['use_automask']),
                                          **{f'in_disp_{s}': v for s, v in disps.items()}, **{f'grad_disp_{s}': v.grad for s, v in disps.items()}))



def run_option_cases(R):
    """Reference options the accelerated path gained in round 3: `PhotoError(weight_ssim != 0.85)`
    (src/losses/photometric.py:65-88), `ReconstructionLoss(mask_name='explainability' | 'uncertainty')` with a predictive
    mask (src/losses/reconstruction.py:46-57, :70-71) and `SmoothReg(use_laplacian=True)` (src/regularizers/smooth.py:33-48)."""
    g = torch.Generator().manual_seed(2468)
    for tag, wgt in (('w0', 0.0), ('w03', 0.3), ('w1', 1.0)):
        pred = torch.rand(3, 3, 15, 21, generator=g).requires_grad_(True)
        tgt = (pred.detach() + 0.2*torch.randn(3, 3, 15, 21, generator=g)).clamp(0, 1)
        tgt[0, :, :5] = pred.detach()[0, :, :5]
        err = R.PhotoError(weight_ssim=wgt)(pred, tgt)
        ge = torch.randn(err.shape, generator=g)
        (err*ge).sum().backward()
        save(f'op_photo_{tag}', dict(in_pred=pred, in_target=tgt, in_ge=ge, meta_weight_ssim=wgt, out_err=err, grad_pred=pred.grad))

    n, b, h, w = 3, 2, 14, 18
    for mask_name, use_min, use_automask, ch in (('explainability', True, True, n), ('uncertainty', True, True, n), ('uncertainty', False, False, n),
                                                 ('explainability', False, True, 1)):
        tgt = texture(g, b, h, w)
        src = torch.stack([texture(g, b, h, w, shift=(1.5*(i + 1), -0.7*(i + 1))) for i in range(n)])
        pred = (src + 0.05*torch.randn(n, b, 3, h, w, generator=g)).clamp(0, 1).requires_grad_(True)
        if mask_name == 'explainability': mask = torch.rand(b, ch, h, w, generator=g)            # sigmoid-like weights
        else: mask = 0.3*torch.randn(b, ch, h, w, generator=g)                                    # log-variance-like
        mask.requires_grad_(True)
        crit = R.ReconstructionLoss(loss_name='ssim', use_min=use_min, use_automask=use_automask, mask_name=mask_name)
        noise_log = []
        orig = torch.randn_like

        def rec(tensor, *a, **k):
            out = orig(tensor, *a, **k); noise_log.append(out.clone()); return out
        torch.manual_seed(99)
        torch.randn_like = rec
        try: loss, ld = crit(pred, tgt, source=src, mask=mask)
        finally: torch.randn_like = orig
        loss.backward()
        rec_ = dict(in_pred=pred, in_target=tgt, in_source=src, in_mask=mask, meta_mask_name=mask_name, meta_use_min=int(use_min),
                    meta_use_automask=int(use_automask), out_loss=loss, grad_pred=pred.grad, grad_mask=mask.grad)
        if noise_log: rec_['in_noise'] = noise_log[0]
        if 'automask' in ld: rec_['out_automask'] = ld['automask']
        save(f'op_recon_mask_{mask_name[:5]}_min{int(use_min)}_auto{int(use_automask)}_c{ch}', rec_)

    for use_edges in (True, False):
        disp = (0.05 + 0.9*torch.rand(3, 1, 12, 20, generator=g)).requires_grad_(True)
        img = texture(g, 3, 12, 20)
        l, ld = R.SmoothReg(use_edges=use_edges, use_laplacian=True)(disp, img)
        l.backward()
        save(f'op_smooth_lap_edges{int(use_edges)}', dict(in_disp=disp, in_img=img, out_loss=l, out_disp_grad=ld['disp_grad'],
                                                           out_image_grad=ld['image_grad'], grad_disp=disp.grad))



def run_aspect_cases(R):
    """GPU-side aspect-ratio augmentation: everything of `src/core/aspect_ratio.py` that runs without
    kornia — the crop / resize shape sampling under seeded generators, `resize_aug` on a whole batch (images, depth, K), the
    not-applied branch of `aspect_ratio_aug` with a `ref_shape`, and `centre_crop_K`.  (`KT.center_crop` itself cannot be run
    here: kornia is absent; that half of the oracle is restated from kornia's published algorithm, "parity unpinned".)"""
    import random
    import importlib
    AR = importlib.import_module('src.core.aspect_ratio')
    geo = importlib.import_module('src.tools.geometry')
    rows = []
    for seed, shape, lo, hi in ((0, (376, 1242), 0.5, 1.0), (1, (376, 1242), 0.5, 1.0), (2, (720, 1280), 0.5, 1.0), (3, (192, 640), 0.6, 0.9),
                                (4, (384, 640), 0.5, 1.0), (5, (376, 1242), 0.3, 0.7), (6, (720, 1280), 0.5, 1.0), (7, (96, 128), 0.5, 1.0)):
        random.seed(seed); torch.manual_seed(seed)
        crop, r = AR.sample_crop(shape, lo, hi)
        res = AR.sample_resize(crop, (192, 640), eps=0.8)
        res1 = AR.sample_resize(shape, (192, 640), eps=1)
        rows.append([seed, shape[0], shape[1], lo, hi, crop[0], crop[1], r, res[0], res[1], res1[0], res1[1]])
    g = torch.Generator().manual_seed(77)
    b, n, h, w = 2, 1, 24, 40
    mk = lambda *s: torch.rand(*s, generator=g)
    x = {'imgs': mk(b, 3, h, w), 'supp_imgs': mk(n, b, 3, h, w)}
    y = {'imgs': mk(b, 3, h, w), 'supp_imgs': mk(n, b, 3, h, w), 'depth': 1 + 9*mk(b, 1, h, w),
         'K': torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]])[None].repeat(b, 1, 1)}
    rec = {f'in_x_{k}': v.clone() for k, v in x.items()}
    rec.update({f'in_y_{k}': v.clone() for k, v in y.items()})
    xo, yo, mo = AR.resize_aug(({k: v.clone() for k, v in x.items()}, {k: v.clone() for k, v in y.items()}, {}), ref_shape=(32, 96), eps=0.8)
    rec.update({f'out_x_{k}': v for k, v in xo.items()}); rec.update({f'out_y_{k}': v for k, v in yo.items()})
    rec['meta_res_shape'] = np.array(xo['imgs'].shape[-2:]); rec['meta_augs'] = str(mo['augs'])
    random.seed(11)
    xo2, yo2, mo2 = AR.aspect_ratio_aug(({k: v.clone() for k, v in x.items()}, {k: v.clone() for k, v in y.items()}, {}), p=0.0, ref_shape=(32, 64))
    rec.update({f'out2_x_{k}': v for k, v in xo2.items()}); rec.update({f'out2_y_{k}': v for k, v in yo2.items()})
    rec['out_centre_crop_K'] = geo.centre_crop_K(y['K'], (17, 30), (h, w))
    rec['sampling'] = np.array(rows, dtype=np.float64)
    save('ar_reference', rec)


def run_baseline_cases(R, kbr):
    """The reference at the resolutions BASELINE.json quotes: cfg 2's 192x640 with two supports and cfg 4/5's
    384x640 with learned intrinsics and four supports, one sample each, compact layout."""
    run_trainer_case(R, 'train_kbr_192x640', seed=2024, b=1, h=192, w=640, n=2, scales=[0, 1, 2, 3], supp_idxs=[-1, 1], compact=4, **kbr)
    run_trainer_case(R, 'train_learnK_n4_384x640', seed=2025, b=1, h=384, w=640, n=4, scales=[0, 1, 2, 3], supp_idxs=[-2, -1, 1, 2],
                     learn_K=True, compact=8, **kbr)


def run_decoder_case(R):
    """§8f rank 4: the reference's `MonodepthDecoder` (src/networks/decoders/monodepth.py) on named, seeded weights / encoder features / output
    gradients (exact_inputs.py): its four sigmoid disparities, the gradients w.r.t. every encoder feature, and per parameter the gradient's sum,
    sum of magnitudes and — for the small ones (heads, the thin last stage, biases) — the gradient itself."""
    from exact_inputs import DECODER_KW, bit_checksum, decoder_feats, decoder_out_grads, decoder_state
    from src.networks.decoders.monodepth import MonodepthDecoder as RefDec
    dec = RefDec(**DECODER_KW)
    holder = torch.nn.Module(); holder.decoders = torch.nn.ModuleDict({'disp': dec})
    shapes = {k: tuple(v.shape) for k, v in holder.state_dict().items()}
    state = decoder_state(shapes)
    holder.load_state_dict(state, strict=True)
    feats = [f.requires_grad_(True) for f in decoder_feats()]
    gouts = decoder_out_grads()
    out = dec(feats)
    sum((out[i]*gouts[i]).sum() for i in out).backward()
    rec = {'meta_keys': np.array(sorted(shapes)), 'chk_state': np.int64(sum(bit_checksum(v) for v in state.values())),
           'chk_feats': np.int64(sum(bit_checksum(f.detach()) for f in feats)), 'chk_gouts': np.int64(sum(bit_checksum(v) for v in gouts.values()))}
    for i, o in out.items(): rec[f'out_{i}'] = o
    for j, f in enumerate(feats): rec[f'gfeat_{j}'] = f.grad
    named = dict(holder.named_parameters())
    stats = []
    for k in sorted(shapes):
        g = named[k].grad.double()
        stats.append([g.sum().item(), g.abs().sum().item()])
        if g.numel() <= 5000: rec['gparam_' + k] = named[k].grad
    rec['gparam_stats'] = np.array(stats)
    save('net_decoder_64x96', rec)
    print('net_decoder_64x96: out_0 mean', out[0].mean().item(), 'params', len(shapes))


def save(name, rec):
    arrs = {}
    for k, v in rec.items():
        if isinstance(v, torch.Tensor): v = v.detach().cpu().numpy()
        arrs[k] = np.asarray(v)
    np.savez_compressed(OUT/f'{name}.npz', **arrs)


def main():
    torch.set_num_threads(8)
    R = import_reference()
    if '--handlers-only' in sys.argv:   # regenerate just the §8f rank-3 fixtures
        run_handler_cases(R)
        return
    if '--options-only' in sys.argv:    # regenerate just the round-3 option fixtures
        run_option_cases(R)
        return
    if '--aspect-only' in sys.argv:
        run_aspect_cases(R)
        return
    kbr = dict(loss_kw=dict(loss_name='ssim', use_min=True, use_automask=True), smooth_kw=dict(use_edges=True),
               min_depth=0.1, max_depth=100)
    if '--baseline-only' in sys.argv:
        run_baseline_cases(R, kbr)
        return
    if '--decoder-only' in sys.argv:     # regenerate just the §8f rank-4 decoder fixture (round 5)
        run_decoder_case(R)
        return
    run_baseline_cases(R, kbr)
    kbr = dict(loss_kw=dict(loss_name='ssim', use_min=True, use_automask=True), smooth_kw=dict(use_edges=True),
               min_depth=0.1, max_depth=100)
    # cfg-1-shaped miniature of the headline configuration
    run_trainer_case(R, 'train_kbr_24x32', seed=42, b=2, h=24, w=32, n=2, scales=[0, 1, 2, 3], supp_idxs=[-1, 1], **kbr)
    # larger, always_fwd_pose False (kbr), ties in a saturated patch
    run_trainer_case(R, 'train_kbr_96x128', seed=195, b=1, h=96, w=128, n=2, scales=[0, 1, 2, 3], supp_idxs=[-1, 1],
                     always_fwd_pose=False, flat_patch=True, **kbr)
    # learned intrinsics (cfg 4 family), 4 supports (cfg 5 family), 2 scales, bigger motion
    run_trainer_case(R, 'train_learnK_n4_40x56', seed=335, b=2, h=40, w=56, n=4, scales=[0, 1], supp_idxs=[-2, -1, 1, 2],
                     learn_K=True, pose_scale=0.03, **kbr)
    # mean-reprojection, no automask, single support / single scale, no depth range (to_inv), plain smoothness
    run_trainer_case(R, 'train_mean_n1_s1_33x47', seed=7, b=3, h=33, w=47, n=1, scales=[0], supp_idxs=[1],
                     loss_kw=dict(loss_name='ssim', use_min=False, use_automask=False), smooth_kw=dict(use_edges=False),
                     min_depth=None, max_depth=None)
    # min without automask; l1 photometric; odd sizes
    run_trainer_case(R, 'train_min_noauto_25x38', seed=11, b=2, h=25, w=38, n=3, scales=[0, 1, 2], supp_idxs=[-1, 1, 2],
                     loss_kw=dict(loss_name='ssim', use_min=True, use_automask=False), smooth_kw=None, min_depth=0.1, max_depth=100)
    run_trainer_case(R, 'train_l1_automask_24x32', seed=13, b=2, h=24, w=32, n=2, scales=[0, 2], supp_idxs=[-1, 1],
                     loss_kw=dict(loss_name='l1', use_min=True, use_automask=True), smooth_kw=dict(use_edges=True),
                     min_depth=0.1, max_depth=100)
    # large motion: z clamp / border clamp active
    run_trainer_case(R, 'train_bigmotion_24x32', seed=21, b=2, h=24, w=32, n=2, scales=[0, 1], supp_idxs=[-1, 1],
                     pose_scale=0.15, **kbr)
    run_op_cases(R)
    run_handler_cases(R)
    run_option_cases(R)
    run_aspect_cases(R)
    run_decoder_case(R)


if __name__ == '__main__':
    main()
