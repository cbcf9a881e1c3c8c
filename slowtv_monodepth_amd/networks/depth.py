"""`DepthNet` — registry key `depth` (reference: `src/networks/depth.py:16-156`)."""
from __future__ import annotations

import torch.nn as nn

from ..registry import DEC_REG, register
from .encoders import create_encoder

__all__ = ['DepthNet']


@register('depth')
class DepthNet(nn.Module):
    """Image -> multi-scale sigmoid disparity {s: (b,1,h/2^s,w/2^s)} + encoder features.

    Same constructor kwargs as the reference.  Virtual stereo, stereo blending and predictive masks are ablation
    features outside the accelerated configurations and raise `NotImplementedError` when enabled.
    """
    def __init__(self, enc_name: str = 'resnet18', pretrained: bool = True, dec_name: str = 'monodepth', out_scales=(0, 1, 2, 3),
                 mask_name=None, num_ch_mask=None, use_virtual_stereo: bool = False, use_stereo_blend: bool = False):
        super().__init__()
        if dec_name not in DEC_REG: raise KeyError(f'Invalid decoder. ({dec_name} vs. {list(DEC_REG)}')
        if mask_name not in {None, 'explainability', 'uncertainty'}: raise KeyError(f'Invalid mask. ({mask_name})')
        if mask_name is not None or use_virtual_stereo or use_stereo_blend:
            raise NotImplementedError('mask prediction / virtual stereo / stereo blending are outside the accelerated path')
        self.enc_name, self.pretrained, self.dec_name = enc_name, pretrained, dec_name
        self.out_scales = [out_scales] if isinstance(out_scales, int) else list(out_scales)
        self.mask_name, self.num_ch_mask = mask_name, num_ch_mask
        self.use_virtual_stereo, self.use_stereo_blend = use_virtual_stereo, use_stereo_blend
        self.encoder = create_encoder(enc_name, in_chans=3, pretrained=pretrained)
        self.num_ch_enc, self.enc_sc = self.encoder.feature_info.channels(), self.encoder.feature_info.reduction()
        self.decoders = nn.ModuleDict({'disp': DEC_REG[dec_name](
            num_ch_enc=self.num_ch_enc, enc_sc=self.enc_sc, upsample_mode='nearest', use_skip=True,
            out_sc=self.out_scales, out_ch=1, out_act='sigmoid')})

    def forward(self, x):
        feat = self.encoder(x)
        out = {'depth_feats': feat}
        for k, dec in self.decoders.items(): out[k] = dict(sorted(dec(feat).items()))
        return out
