"""`MonoDepthModule` — the training step around the hot path (reference: `src/core/trainer.py:17-472`).

Same cfg sections (`net`, `loss`, `optimizer`, `scheduler`, `trainer`), same phase structure
(`forward` -> `forward_postprocess` -> `forward_loss`), same `fwd` / `loss_dict` keys.  What differs is what runs
underneath: the pose prologue, post-process and loss phases are ~15 HIP launches forward + backward instead of ~350 ATen
kernels, and the reference's per-phase `cuda.synchronize()` timers (src/utils/timers.py:178,195) are replaced by HIP
events that are only read when someone asks for them.  The Lightning shell is replaced by `fit()` in `train.py`.
"""
from __future__ import annotations

import copy
import os
from contextlib import contextmanager, nullcontext

import torch
import torch.nn as nn

from . import parsers
from .geometry import ViewSynth

__all__ = ['MonoDepthModule', 'HipLossBackend', 'EventTimer']


class HipLossBackend:
    """The product loss path: K0 kernel + fused handlers.  (Tests may inject another object with the same three methods,
    e.g. the CPU oracle, to exercise the host logic on machines without a GPU; the product never does.)"""
    def __init__(self): self._unserved = set()   # configurations `loss_path` found the single-node operator does not serve: not tried again

    def postprocess(self, disps: dict, size, min_depth, max_depth, want_disp_up=True):
        from . import functional as F
        from .handlers import LazyDepths, ScaleDict
        keys = list(disps.keys())
        if not want_disp_up:   # K0 fused into the reconstruction kernel: nothing is launched here (handlers.LazyDepths)
            return None, LazyDepths(keys, [disps[k].float() for k in keys], size, min_depth, max_depth)
        depth_up, disp_up = F.disp_to_depth([disps[k].float() for k in keys], size, min_depth, max_depth, want_disp_up=want_disp_up)
        return ScaleDict.from_stack(keys, disp_up), ScaleDict.from_stack(keys, depth_up)

    def image_recon(self, crit, synth, depths, masks, imgs, supp_imgs, Ts, Ks, want_warp=True, K_inv=None, prepared=None):
        from . import handlers
        self.last_sel, self.last_path = None, 'handlers: image_recon + disp_smooth as separate autograd nodes'
        return handlers.image_recon(crit, synth, depths, masks, imgs, supp_imgs, Ts.float(), Ks.float(), K_inv=K_inv, want_warp=want_warp, prepared=prepared)

    def inv_intrinsics(self, K):
        """K^-1 of the dataset's intrinsics.  A loader that hands out the SAME tensor object again (a fixed-camera dataset collated once, the
        synthetic batches of the bench) gets the inverse of the previous step back: keyed on the object (kept alive here, so that its address
        cannot be recycled for another batch's K) and its version counter."""
        from . import functional as F
        if not K.is_cuda: return None
        hit = self.__dict__.get('_kinv_cache')
        if hit is not None and hit[0] is K and hit[1] == K._version: return hit[2]
        Ki = F.inv_intrinsics(K.float())
        self.__dict__['_kinv_cache'] = (K, K._version, Ki)
        return Ki

    def prepare_frames(self, crit, imgs, supp_imgs, pyramid, stream, smooth_edges=False):
        """The frame-only half of the reconstruction forward (texel repack, target window sums, identity error of the automask:
        everything `handlers.image_recon` needs that no network output enters), enqueued on `stream` so that it runs under the
        networks instead of after them.  None when the fused operator will not be used for these tensors."""
        from . import functional as F
        if not imgs.is_cuda or imgs.shape[1] != 3 or crit.loss_name == 'l2' or imgs.dtype != torch.float32: return None
        if getattr(crit, 'mask_name', None): return None   # a masked criterion takes the un-fused operators (handlers.image_recon): nothing would read the buffer
        return F.image_recon_prep(imgs, supp_imgs, flags=F.recon_flags(crit.loss_name, crit.use_min, crit.use_automask), pyramid=pyramid, stream=stream,
                                  smooth_edges=smooth_edges)

    def invert_mask(self, invert, device):
        """python bool list -> cached uint8 device tensor (None if no pose is inverted): the same flags every step, so the device copy is kept
        (and no host-to-device copy happens inside a HIP-graph capture)."""
        if not any(invert): return None
        key = (tuple(invert), device)
        cache = self.__dict__.setdefault('_invert_masks', {})
        inv = cache.get(key)
        if inv is None: inv = cache[key] = torch.tensor(list(invert), dtype=torch.uint8).to(device)
        return inv

    def pose_matrices(self, aa, t, invert):
        """(N,3),(N,3) + python bool list -> (N,4,4); one launch for Rodrigues + the backward-in-time inverses."""
        from . import functional as F
        return F.pose_matrices(aa.float(), t.float(), self.invert_mask(invert, aa.device))

    def intrinsics(self, fs, cs, size):
        from . import functional as F
        return F.intrinsics(fs.float(), cs.float(), size)

    def disp_smooth(self, crit, disps, imgs, want_aux=True, prepared=None):
        from . import handlers
        return handlers.disp_smooth(crit, {k: d.float() for k, d in disps.items()}, imgs, want_aux=want_aux, prepared=prepared)

    def loss_path(self, crit, reg, depths, disps, imgs, supp_imgs, Ts, Ks, K_inv, w_recon, w_smooth, pose=None, intrinsics=None, prepared=None):
        """`img_recon` + `disp_smooth` + their weighted sum as ONE autograd node (`functional.loss_path_fused`: 1 launch forward, 3 backward).
        -> (loss, l_recon, l_smooth), or None when the operator does not serve this configuration (the caller then runs the two handlers)."""
        from . import functional as F
        from ._lib import Unsupported
        from .handlers import LazyDepths
        if not (isinstance(depths, LazyDepths) and depths.pending and imgs.is_cuda and imgs.shape[1] == 3 and imgs.dtype == torch.float32): return None
        if crit.loss_name != 'ssim' or getattr(crit, 'mask_name', None): return None
        if not reg.use_edges or getattr(reg, 'use_laplacian', False) or getattr(reg, 'use_blur', False): return None
        # what the operator is known not to serve, checked before anything is allocated or a tie-break seed is drawn (a consumed seed would shift the noise
        # sequence of the handlers' path that then runs): more supports than one pass takes, a single pyramid level; and any configuration that raised once
        n_supp, cfg_key = supp_imgs.shape[0], (supp_imgs.shape[0], len(depths.disps), tuple(imgs.shape), crit.use_min, crit.use_automask, intrinsics is not None, pose is not None)
        if n_supp > F.supports_per_pass() or len(depths.disps) < 2 or cfg_key in self._unserved: return None
        flags = F.recon_flags(crit.loss_name, crit.use_min, crit.use_automask)
        dl = [d.float() for d in depths.disps]
        if prepared is not None and not prepared.matches(imgs, supp_imgs, flags, [d.shape[-2] for d in dl], [d.shape[-1] for d in dl]): prepared = None
        try:
            loss, l_rec, l_sm, _sel, depth_up = F.loss_path_fused(dict(zip(depths.keys_, dl)), imgs, supp_imgs, Ts.float(), Ks.float(), K_inv, pose=pose, intrinsics=intrinsics,
                                                                  flags=flags, min_depth=depths.min_depth, max_depth=depths.max_depth, seed=crit.next_seed(),
                                                                  w_recon=w_recon, w_smooth=w_smooth, prepared=prepared)
        except Unsupported:
            self._unserved.add(cfg_key)
            return None
        self.last_sel, self.last_path = _sel, 'single node: smd_loss_path_fwd (1 launch) / smd_loss_path_bwd (3 launches)'   # (references only: what bench.py reports about the last step)
        depths.adopt(depth_up.detach())    # `fwd['depth_up']` for later readers (metrics, logging); no differentiable consumer may follow
        return loss, l_rec, l_sm


class EventTimer:
    """Nested phase timer with the reference's keys ('Total', 'Forward', 'Post-Process', 'Loss', 'Loss-<k>', 'Backward';
    src/core/trainer.py:100-102,170-175,384) that records HIP events instead of synchronising the device per phase."""
    def __init__(self, enabled: bool = False):
        self.enabled, self.events = enabled, {}

    @contextmanager
    def __call__(self, key: str):
        if not (self.enabled and torch.cuda.is_available()):
            yield
            return
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        try: yield
        finally:
            end.record()
            self.events[key] = (start, end)

    def to_dict(self) -> dict:
        if not self.events: return {}
        torch.cuda.synchronize()
        return {k: s.elapsed_time(e) for k, (s, e) in self.events.items()}

    def reset(self): self.events = {}


class MonoDepthModule(nn.Module):
    """Depth + pose networks, losses and the per-batch step.

    :param cfg: the reference's config dict (see cfg/default.yaml there); sections `net`, `loss`, `trainer` are read here.
    :param loss_backend: object providing postprocess / image_recon / disp_smooth (default: the HIP path).
    """
    def __init__(self, cfg: dict, loss_backend=None):
        super().__init__()
        self.cfg = cfg
        tcfg = cfg.get('trainer', {})
        self.nets = parsers.get_net(cfg['net'])
        self.losses, self.weights = parsers.get_loss(copy.deepcopy(cfg['loss']))
        self.backend = loss_backend or HipLossBackend()
        self._w_cache = {}     # the frozen loss weights as host numbers, re-read whenever a state dict (checkpoint, --resume) rewrote them: `_loss_weight`
        # Losses whose inputs only networks outside this package produce (the autoencoder network and the
        # virtual-stereo decoder head are out of scope; their handlers and criteria exist and are parity-tested on their own):
        # refuse at construction instead of failing with a KeyError in the middle of the first step.
        needs = {'feat_recon': ('autoencoder', 'an `autoencoder` network (fwd["autoenc_feats"])'),
                 'autoenc_recon': ('autoencoder', 'an `autoencoder` network (fwd["autoenc_imgs_up"])'),
                 'stereo_const': (None, 'a depth network with `use_virtual_stereo` (fwd["disp_stereo"], its up-sampled forms and y["T_stereo"])')}
        for k, (net_key, what) in needs.items():
            if k in self.losses and (net_key is None or net_key not in self.nets):
                raise NotImplementedError(f'loss "{k}" needs {what}, which this package does not build; the handler `handlers.{k}` can be called directly')
        self.synth = None
        self.scales = self.nets['depth'].out_scales
        self.n_scales = len(self.scales)
        self.min_depth, self.max_depth = tcfg.get('min_depth', None), tcfg.get('max_depth', None)
        self.always_fwd_pose = tcfg.get('always_fwd_pose', True)
        self.auto_scale_lr = tcfg.get('auto_scale_lr', False)
        # GPU-side aspect-ratio augmentation of the training batches (src/core/trainer.py:54-60, 106): off at probability 0
        self.ar_kwargs = dict(p=tcfg.get('aspect_ratio_aug_prob', 0.0), crop_min=tcfg.get('aspect_ratio_min', 0.5),
                              crop_max=tcfg.get('aspect_ratio_max', 1), ref_shape=tcfg.get('aspect_ratio_ref_shape', None))
        prec = str(tcfg.get('precision', 32))
        self.amp_dtype = {'32': None, '32-true': None, 'bf16': torch.bfloat16, 'bf16-mixed': torch.bfloat16}.get(prec, None)
        self.channels_last = bool(tcfg.get('channels_last', False))
        self.want_aux = bool(tcfg.get('log_images', False))  # supp_imgs_warp etc. are only for the image logger
        self.overlap_nets = bool(tcfg.get('overlap_nets', True))   # pose network on a second HIP stream, concurrent with the depth network
        self._side_streams = {}
        self._prep_streams = {}
        self._prepared = None
        # frame-only half of the reconstruction loss ahead of the networks: 'own' = on its own side stream at the start of the step,
        # 'pose' = on the pose network's side stream behind that network, False = inline, inside the loss (after the networks)
        self.prep_ahead = tcfg.get('prep_ahead', os.environ.get('SMD_PREP_AHEAD', 'pose'))
        if self.prep_ahead in ('0', 'false', 'False', 0): self.prep_ahead = False
        if self.prep_ahead in ('1', 'true', 'True', 1, True): self.prep_ahead = 'pose'   # measured at cfg 2: 'pose' 16.91-16.96 ms per step = inline (16.89-16.95); 'own' 17.13-17.23 (a third stream beside the two networks)
        if self.prep_ahead == 'pose' and not self.overlap_nets: self.prep_ahead = 'own'
        self.timer = EventTimer(enabled=bool(tcfg.get('profile_phases', False)))
        if self.channels_last: self.nets.to(memory_format=torch.channels_last)

    # ------------------------------------------------------------------------------------------------
    def _autocast(self, device_type):
        return torch.autocast(device_type, dtype=self.amp_dtype) if self.amp_dtype is not None else nullcontext()

    def forward(self, x: dict) -> dict:
        """Network forward (src/core/trainer.py:192-278): `disp` {s: (b,1,h/2^s,w/2^s)}, `T_{idx}` (b,4,4) per support,
        and with a `learn_K` pose net `K` (b,4,4), `fs`, `cs`."""
        fwd = {}
        imgs = x['imgs']
        if self.channels_last: imgs = imgs.contiguous(memory_format=torch.channels_last)
        idxs_all = [int(i) for i in x['supp_idxs']]
        # The two networks are independent until the loss.  Their deep stages launch kernels of a few dozen workgroups on a
        # 256-CU device, so the pose network is enqueued on a second HIP stream and runs concurrently with the depth
        # network (autograd replays each backward node on its forward stream, so the backward overlaps the same way).
        side = None
        if self.overlap_nets and imgs.is_cuda and 'pose' in self.nets and 'depth' in self.nets:
            main = torch.cuda.current_stream(imgs.device)
            side = self._side_streams.setdefault(imgs.device.index, torch.cuda.Stream(device=imgs.device))
            side.wait_stream(main)
            x['imgs'].record_stream(side); x['supp_imgs'].record_stream(side)
        for key in sorted(self.nets.keys(), key=lambda k: k != 'pose'):   # enqueue the side-stream branch first
            net = self.nets[key]
            if key == 'depth':
                with self._autocast(imgs.device.type): out = net(imgs)
                fwd.update(out)
            elif key == 'pose':
                with (torch.cuda.stream(side) if side is not None else nullcontext()):
                    produced = self._forward_pose(net, x, idxs_all)
                    if self.prep_ahead == 'pose' and side is not None and getattr(self, '_y', None) is not None:
                        self._prepared = self._prepare_frames(self._y, stream=side)
                        # the inverse of the dataset's intrinsics needs no network output either: same stream, same place (main waits
                        # for this stream before the loss); with a learned K the pose network's own K_inv is used instead
                        inv = getattr(self.backend, 'inv_intrinsics', None)
                        if inv is not None and 'K' not in produced and 'K' in self._y:
                            self._K_inv = inv(self._y['K'])
                            if self._K_inv is not None: self._K_inv.record_stream(main)
                if side is not None:
                    for k_, v in produced.items():
                        if isinstance(v, torch.Tensor): v.record_stream(main)
                    for v in (produced.get('_pose_leaves') or ())[:2]: v.record_stream(main)   # read by the loss path's backward on the main stream
                fwd.update(produced)
            else:
                raise KeyError(f'Unrecognized key: {key}.')
        if side is not None: main.wait_stream(side)
        return fwd

    def _forward_pose(self, net, x: dict, idxs_all) -> dict:
        out = {}
        inv = lambda i: self.always_fwd_pose and i < 0
        pairs = torch.stack([torch.cat([supp, x['imgs']] if inv(i) else [x['imgs'], supp], dim=1)
                             for i, supp in zip(idxs_all, x['supp_imgs']) if i != 0])   # (n,b,6,h,w)
        sh = pairs.shape[:2]
        pin = pairs.flatten(0, 1)
        if self.channels_last: pin = pin.contiguous(memory_format=torch.channels_last)
        with self._autocast(pin.device.type): pose = net(pin)
        pose = {k: v.float() for k, v in pose.items()}
        idxs = [i for i in idxs_all if i != 0]
        flags = [bool(inv(i)) for i in idxs for _ in range(sh[1])]
        aa, tr = pose['R'][:, 0].contiguous(), pose['t'][:, 0].contiguous()   # (strided views of the head's output: packed ONCE, for `pose_matrices` and for the loss path)
        Ts = self.backend.pose_matrices(aa, tr, flags).unflatten(0, sh)
        # for the fused loss path (its backward hands the gradients to the network's outputs directly): they travel with `fwd`, not on the module — a bare
        # `module.forward(x)` (validation, export) must not keep graph-attached tensors alive until the next call
        out['_Ts_all'], out['_pose_leaves'] = Ts, (aa, tr, flags, idxs)
        for i, T in zip(idxs, Ts): out[f'T_{i}'] = T
        if 'fs' in pose:
            out['fs'], out['cs'] = pose['fs'].unflatten(0, sh), pose['cs'].unflatten(0, sh)
            out['K'], out['K_inv'] = self.backend.intrinsics(out['fs'][0], out['cs'][0], x['imgs'].shape[-2:])  # first support's prediction only
        return out

    def forward_postprocess(self, fwd: dict, x: dict, y: dict) -> dict:
        """Upsample + to-depth of every scale in one launch, and stack the poses (src/core/trainer.py:280-348)."""
        disp_up, fwd['depth_up'] = self.backend.postprocess(fwd['disp'], tuple(x['imgs'].shape[-2:]), self.min_depth, self.max_depth,
                                                            want_disp_up=self.want_aux)
        if disp_up is not None: fwd['disp_up'] = disp_up   # only the image logger reads the un-scaled up-sampled disparity
        # a stereo support (index 0) brings its known pose with the batch instead of a predicted one (src/core/trainer.py:347)
        leaves, Ts_all = fwd.get('_pose_leaves'), fwd.get('_Ts_all')
        if (leaves is not None and Ts_all is not None and [int(i) for i in x['supp_idxs']] == list(leaves[3])
                and all(fwd[f'T_{i}'].data_ptr() == Ts_all[k].data_ptr() for k, i in enumerate(leaves[3]))):    # (this `fwd` is that call's)
            fwd['Ts'] = Ts_all     # every support's pose came out of ONE `pose_matrices` call, already stacked in this order: no copy
        else:
            fwd['Ts'] = torch.stack([(y['T_stereo'] if int(i) == 0 else fwd[f'T_{int(i)}']) for i in x['supp_idxs']])
        return fwd

    def forward_loss(self, fwd: dict, x: dict, y: dict):
        """Weighted sum of the configured losses (src/core/trainer.py:350-472); `loss_dict['loss_<k>']` per loss."""
        loss, loss_dict = 0., {}
        fused = self._forward_loss_fused(fwd, x, y)
        if fused is not None: return fused
        for k, crit in self.losses.items():
            with self.timer(f'Loss-{k}'):
                if k == 'img_recon':
                    kw = {'prepared': self._prepared} if self._prepared is not None else {}
                    K_inv = fwd.get('K_inv') if 'K' in fwd else getattr(self, '_K_inv', None)
                    if K_inv is None and 'K' not in fwd and hasattr(self.backend, 'inv_intrinsics'): K_inv = self.backend.inv_intrinsics(y['K'])   # (inline prep: same cache)
                    l, ld = self.backend.image_recon(crit, self.synth, fwd['depth_up'], fwd.get('mask_up'), y['imgs'], y['supp_imgs'],
                                                     fwd['Ts'], fwd.get('K', y['K']), want_warp=self.want_aux, K_inv=K_inv, **kw)
                elif k == 'disp_smooth':
                    kw = {'prepared': self._prepared} if self._prepared is not None else {}
                    l, ld = self.backend.disp_smooth(crit, fwd['disp'], y['imgs'], want_aux=self.want_aux, **kw)
                elif k == 'depth_regr':   # proxy-depth (Depth Hints) regression, src/core/trainer.py:425-433
                    if 'depth_hints' not in y: raise KeyError('Missing proxy depth prediction "depth_hints".')
                    from . import handlers
                    l, ld = handlers.depth_regr(crit, self.synth, self.losses['img_recon'].compute_photo, fwd['depth_up'], y['depth_hints'],
                                                y['imgs'], y['supp_imgs'], fwd['Ts'], fwd.get('K', y['K']))
                elif k == 'feat_recon':      # feature-metric reconstruction on an autoencoder's features, src/core/trainer.py:405-411
                    if 'autoenc_feats' not in fwd: raise KeyError('Missing autoencoder features "autoenc_feats" (no `autoencoder` network is configured).')
                    from . import handlers
                    l, ld = handlers.feat_recon(crit, self.synth, fwd['depth_up'], fwd.get('mask_up'), fwd['autoenc_feats'], fwd['supp_autoenc_feats'],
                                                fwd['Ts'], fwd.get('K', y['K']))
                elif k == 'autoenc_recon':   # src/core/trainer.py:413-418
                    if 'autoenc_imgs_up' not in fwd: raise KeyError('Missing autoencoder reconstructions "autoenc_imgs_up" (no `autoencoder` network is configured).')
                    from . import handlers
                    l, ld = handlers.autoenc_recon(crit, fwd['autoenc_imgs_up'], y['imgs'], fwd['supp_autoenc_imgs_up'], y['supp_imgs'])
                elif k == 'stereo_const':    # virtual-stereo consistency, src/core/trainer.py:420-428
                    if 'disp_stereo' not in fwd: raise KeyError('Missing virtual stereo prediction "disp_stereo".')
                    if 'T_stereo' not in y: raise KeyError('Missing stereo pair "T_stereo".')
                    from . import handlers
                    l, ld = handlers.stereo_const(crit, self.synth, fwd['disp_up'], fwd['depth_up'], fwd['disp_stereo_up'], fwd['depth_stereo_up'],
                                                  y['T_stereo'], fwd.get('K', y['K']))
                else:
                    raise ValueError(f'Missing loss key: "{k}"')
            loss = loss + self.weights[k]*l
            loss_dict[f'loss_{k}'] = l
            loss_dict.update(ld)
        return loss, loss_dict

    def _loss_weight(self, k: str) -> float:
        """`self.weights[k]` as a host float for the single-node loss path (the kernel takes the weights as scalars).  The `ParameterDict` is part of the state
        dict: `load_state_dict` / a reference checkpoint / `--resume` overwrite it in place, and the handlers' path (`self.weights[k]*l`) and the reference
        then use the checkpoint's values — so the cached number is keyed on the parameter's version counter and storage (one device read per change)."""
        p = self.weights[k]
        key = (p._version, p.data_ptr(), p.device)
        c = self._w_cache.get(k)
        if c is None or c[0] != key: c = self._w_cache[k] = (key, float(p.detach()))
        return c[1]

    def _forward_loss_fused(self, fwd: dict, x: dict, y: dict):
        """The kbr loss configuration — `img_recon` + `disp_smooth`, nothing else, no image logging — through ONE autograd node
        (`HipLossBackend.loss_path`).  None: not that configuration / not served; `forward_loss` then runs the handlers one by one."""
        fn = getattr(self.backend, 'loss_path', None)
        if fn is None or self.want_aux or set(self.losses.keys()) != {'img_recon', 'disp_smooth'} or fwd.get('mask_up') is not None: return None
        crit, reg = self.losses['img_recon'], self.losses['disp_smooth']
        learned = 'K' in fwd
        K_inv = fwd.get('K_inv') if learned else getattr(self, '_K_inv', None)
        if K_inv is None and not learned and hasattr(self.backend, 'inv_intrinsics'): K_inv = self.backend.inv_intrinsics(y['K'])
        pose, leaves = None, fwd.get('_pose_leaves')
        if leaves is not None and [int(i) for i in x['supp_idxs']] == list(leaves[3]):   # every support's pose is predicted (no stereo frame): Ts is exactly pose_matrices(aa, t)
            inv = self.backend.invert_mask(leaves[2], leaves[0].device) if hasattr(self.backend, 'invert_mask') else None
            pose = (leaves[0].float(), leaves[1].float(), inv)
        # (a stereo support brings a known pose: no pose leaves — then the intrinsics go in as the K, K_inv tensors and autograd carries their gradients;
        # the kernel's own chain rule for (fs, cs) rides on the pose chain's guest block and would fail in backward without it: ADVICE r5)
        intr = (fwd['fs'][0].float(), fwd['cs'][0].float()) if (learned and 'fs' in fwd and K_inv is not None and pose is not None) else None
        with self.timer('Loss-img_recon'):
            out = fn(crit, reg, fwd['depth_up'], fwd['disp'], y['imgs'], y['supp_imgs'], fwd['Ts'], fwd.get('K', y['K']), K_inv,
                     self._loss_weight('img_recon'), self._loss_weight('disp_smooth'), pose=pose, intrinsics=intr, prepared=self._prepared)
        if out is None: return None
        loss, l_rec, l_sm = out
        ld = {'loss_img_recon': l_rec, 'loss_disp_smooth': l_sm}
        out_ld = {f'loss_{k}': ld[f'loss_{k}'] for k in self.losses}
        sel = getattr(self.backend, 'last_sel', None)
        if crit.use_automask and sel is not None: out_ld['automask'] = sel[0] != 255    # the entry the handlers' path adds (handlers.image_recon): same keys whichever path ran
        return loss, out_ld

    def step(self, batch, mode: str = 'train'):
        """One forward pass + losses (src/core/trainer.py:115-190) -> (loss, loss_dict, fwd)."""
        if mode == 'train':   # `training_step`: `batch = self.ar_aug(batch)` on every step (trainer.py:106) — at p = 0 it only draws `random.random()`
            from .aspect_ratio import aspect_ratio_aug
            batch = aspect_ratio_aug(batch, **self.ar_kwargs, resample=getattr(self.backend, 'crop_resize', None))
        x, y, m = batch
        self.synth = ViewSynth(x['imgs'].shape[-2:])
        self._prepared = self._prepare_frames(y) if self.prep_ahead == 'own' else None
        self._y, self._K_inv = y, None
        try:
            with self.timer('Total'):
                with self.timer('Forward'): fwd = self.forward(x)
                with self.timer('Post-Process'): fwd = self.forward_postprocess(fwd, x, y)
                with self.timer('Loss'): loss, loss_dict = self.forward_loss(fwd, x, y)
        finally:
            # the prep-ahead state belongs to THIS step: a later `module.forward(x)` (validation, inference) must neither launch the
            # prep for the previous batch nor keep that batch (and its 150 MB packed buffer) alive
            self._y = self._prepared = self._K_inv = None
        return loss, loss_dict, fwd

    def _prepare_frames(self, y: dict, stream=None):
        """Launch the frame-only half of `img_recon` before the networks (it needs `y['imgs']`, `y['supp_imgs']` only).  With the
        decoder's pyramid (scale s = image size >> s) the K0 row table is built there too; if the networks then emit other sizes
        the handler falls back to an inline prep."""
        crit = self.losses['img_recon'] if 'img_recon' in self.losses else None
        fn = getattr(self.backend, 'prepare_frames', None)
        if not (self.prep_ahead and crit is not None and fn is not None and y['imgs'].is_cuda): return None
        dev = y['imgs'].device
        st = stream if stream is not None else self._prep_streams.setdefault(dev.index, torch.cuda.Stream(device=dev))
        h, w = y['imgs'].shape[-2:]
        pyramid = None if self.want_aux else [(max(h >> s, 1), max(w >> s, 1)) for s in self.scales]   # want_aux: depth comes from the K0 launch
        reg = self.losses['disp_smooth'] if 'disp_smooth' in self.losses else None                    # its edge weights are frame-only too
        edges = (bool(pyramid) and reg is not None and getattr(reg, 'use_edges', False) and not getattr(reg, 'use_laplacian', False)
                 and not getattr(reg, 'use_blur', False))    # (the blurred form computes its weights from the blurred image: nothing would read these)
        return fn(crit, y['imgs'], y['supp_imgs'], pyramid, st, **({'smooth_edges': True} if edges else {}))

    # ------------------------------------------------------------------------------------------------
    def configure_optimizers(self):
        """Optimizer + chained schedulers (src/core/trainer.py:82-92)."""
        out = {'optimizer': parsers.get_opt(self.nets, self.cfg['optimizer'])}
        if cfg := self.cfg.get('scheduler'):
            sch = parsers.get_sched(out['optimizer'], cfg)
            out['lr_scheduler'] = torch.optim.lr_scheduler.ChainedScheduler(list(sch.values()))
        return out
