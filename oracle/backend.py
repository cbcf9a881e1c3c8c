"""CPU loss backend built on the oracle — TEST INFRASTRUCTURE ONLY (see oracle/view_synth_oracle.py).

Implements the three methods `MonoDepthModule` expects from a loss backend so that (a) the host logic (trainer, DDP
sharding, gradient accumulation) can be exercised on machines without a GPU, and (b) `bench.py` can time the same
training step on the host cores as the CPU baseline.  The product never instantiates this class.
"""
from __future__ import annotations

import torch

from . import view_synth_oracle as O


class OracleBackend:
    def __init__(self, aten: bool = True): self.aten = aten

    def postprocess(self, disps, size, min_depth, max_depth, want_disp_up=True):
        disp_up, depth_up = O.disp_to_depth_up({k: d.float() for k, d in disps.items()}, size, min_depth, max_depth, aten=self.aten)
        return (disp_up if want_disp_up else None), depth_up

    def pose_matrices(self, aa, t, invert):
        T = O.T_from_AAt(aa.float(), t.float())
        if any(invert): T = torch.stack([torch.linalg.inv(Ti) if f else Ti for Ti, f in zip(T, invert)])
        return T

    def intrinsics(self, fs, cs, size):
        return O.resize_K(O.build_K(fs.float(), cs.float()), size), None   # K_inv: the oracle inverts K itself, like the reference

    def image_recon(self, crit, synth, depths, masks, imgs, supp_imgs, Ts, Ks, want_warp=True, K_inv=None):
        if masks is not None: raise NotImplementedError
        loss, ld, _ = O.image_recon(depths, imgs, supp_imgs, Ts.float(), Ks.float(), crit.loss_name, crit.use_min, crit.use_automask,
                                    noise=None, aten=self.aten)
        if not want_warp: ld.pop('supp_imgs_warp', None)
        return loss, ld

    def crop_resize(self, tensors, crop_shape, out_shape, K):
        from . import aspect_ratio_oracle as A
        return A.crop_resize(tensors, crop_shape, out_shape, K)

    def disp_smooth(self, crit, disps, imgs, want_aux=True):
        loss, ld = O.disp_smooth({k: d.float() for k, d in disps.items()}, imgs, crit.use_edges, aten=self.aten)
        return loss, (ld if want_aux else {})
