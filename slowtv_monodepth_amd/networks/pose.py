"""`PoseNet` — registry key `pose` (reference: `src/networks/pose.py:13-135`)."""
from __future__ import annotations

import torch.nn as nn

from ..geometry import build_K
from ..registry import register
from .encoders import create_encoder

__all__ = ['PoseNet']


def _block(cin, cout, k, pad=0): return nn.Sequential(nn.Conv2d(cin, cout, k, 1, pad), nn.ReLU(inplace=True))


@register('pose')
class PoseNet(nn.Module):
    """Channel-concatenated image pair (b,6,h,w) -> axis-angle `R` and translation `t`, each (b,2,3) scaled by 0.01
    (only index 0 of the middle dim is used by the trainer); with `learn_K`, also normalised focal lengths `fs`
    (softplus) and principal point `cs` (sigmoid), each (b,2)."""
    def __init__(self, enc_name: str = 'resnet18', learn_K: bool = False, pretrained: bool = False):
        super().__init__()
        self.enc_name, self.learn_K, self.pretrained = enc_name, learn_K, pretrained
        self.n_imgs = 2
        self.encoder = create_encoder(enc_name, in_chans=3*self.n_imgs, pretrained=pretrained)
        self.n_ch_enc = self.encoder.feature_info.channels()
        self.n_ch_dec = 256
        self.pose_eps = 0.01
        self.squeeze = _block(self.n_ch_enc[-1], self.n_ch_dec, 1)
        self.decoders = nn.ModuleDict({'pose': self._head(6*self.n_imgs, nn.Unflatten(-1, (self.n_imgs, 6)))})
        if learn_K:
            self.decoders['focal'] = self._head(2, nn.Softplus())
            self.decoders['offset'] = self._head(2, nn.Sigmoid())

    build_K = staticmethod(build_K)

    def _head(self, cout, tail):
        c = self.n_ch_dec
        return nn.Sequential(_block(c, c, 3, 1), _block(c, c, 3, 1), nn.Conv2d(c, cout, 1), nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten(), tail)

    def forward(self, x):
        feat = self.squeeze(self.encoder(x)[-1])
        out = self.pose_eps*self.decoders['pose'](feat)
        out = {'R': out[..., :3], 't': out[..., 3:]}
        if self.learn_K:
            out['fs'] = self.decoders['focal'](feat)
            out['cs'] = self.decoders['offset'](feat)
        return out
