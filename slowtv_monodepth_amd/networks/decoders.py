"""Monodepth(2) decoder — registry key `monodepth` (reference: `src/networks/decoders/monodepth.py:14-89`)."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..registry import register

__all__ = ['MonodepthDecoder', 'ACT']

ACT = {'sigmoid': nn.Sigmoid(), 'relu': nn.ReLU(inplace=True), 'none': nn.Identity(), None: nn.Identity()}


def conv3x3(cin: int, cout: int) -> nn.Conv2d:
    """3x3 conv with reflection padding (src/networks/decoders/utils.py:44-46)."""
    return nn.Conv2d(cin, cout, 3, padding=1, padding_mode='reflect')


class ConvELU(nn.Sequential):
    def __init__(self, cin, cout): super().__init__(conv3x3(cin, cout), nn.ELU(inplace=True))


@register('monodepth')
class MonodepthDecoder(nn.Module):
    """Five up-convolution stages (256..16 channels) with encoder skips where a matching stride exists, and a
    3x3 output head per requested scale.

    :param num_ch_enc / enc_sc: channels and strides of the encoder features.
    :param out_sc: scales (as log2 stride) at which to emit a prediction; out_ch / out_act: its channels / activation.
    """
    def __init__(self, num_ch_enc, enc_sc, upsample_mode: str = 'nearest', use_skip: bool = True,
                 out_sc=(0, 1, 2, 3), out_ch: int = 1, out_act: str = 'sigmoid'):
        super().__init__()
        if out_act not in ACT: raise KeyError(f'Invalid activation key. ({out_act} vs. {tuple(ACT.keys())}')
        self.num_ch_enc, self.enc_sc = list(num_ch_enc), list(enc_sc)
        self.upsample_mode, self.use_skip, self.out_sc, self.out_ch = upsample_mode, use_skip, list(out_sc), out_ch
        self.act = ACT[out_act]
        self.num_ch_dec = [16, 32, 64, 128, 256]
        self.up0, self.up1, self.out = nn.ModuleDict(), nn.ModuleDict(), nn.ModuleDict()
        for i in range(4, -1, -1):
            cin = self.num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            self.up0[str(i)] = ConvELU(cin, self.num_ch_dec[i])
            cin = self.num_ch_dec[i]
            if use_skip and 2**i in self.enc_sc: cin += self.num_ch_enc[self.enc_sc.index(2**i)]
            self.up1[str(i)] = ConvELU(cin, self.num_ch_dec[i])
        for i in self.out_sc: self.out[str(i)] = conv3x3(self.num_ch_dec[i], out_ch)

    def forward(self, feat):
        x = feat[-1]
        amp_bf16 = torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
        if x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and (amp_bf16 or not torch.is_autocast_enabled()) and self.upsample_mode == 'nearest':
            return self._forward_glued(feat, torch.bfloat16 if amp_bf16 else None)
        out = {}
        for i in range(4, -1, -1):
            x = F.interpolate(self.up0[str(i)](x), scale_factor=2, mode=self.upsample_mode)
            if self.use_skip and 2**i in self.enc_sc: x = torch.cat((x, feat[self.enc_sc.index(2**i)]), 1)
            x = self.up1[str(i)](x)
            if i in self.out_sc: out[i] = self.act(self.out[str(i)](x))
        return out

    def _forward_glued(self, feat, out_dtype=None):
        """Same network, same parameters; the ops BETWEEN the convolutions (ELU, nearest x2, cat, reflection pad) run as
        the two gather kernels of `csrc/smd_decoder.hip`, each writing the next convolution's padded input, and the
        padded ELU output of a stage is shared by its output head and the next stage (the reference pads it twice)."""
        from .. import functional as HF
        def conv(m, xp):   # input already reflection-padded; the bias is added by the next glue kernel
            co, ci = m.weight.shape[:2]
            if (co % 32 == 0 and ci % 16 == 0) or (co == 16 and ci in (16, 32)):
                # smd_conv3x3_mfma_* (bf16 matrix cores; fp32 tensors: three-way split operands, fp32-class results; bf16 tensors under autocast: one piece) or, per
                # operator and shape by this box's A/B, MIOpen (the wide stages) / the f32-MFMA kernels smd_conv3x3_thin_* (the 16-channel last stage in fp32)
                return HF.conv3x3_wide(xp, m.weight.float())
            return F.conv2d(xp, m.weight)
        out = {}
        xp = HF.elu_pad(feat[-1], apply_elu=False, out_dtype=out_dtype)   # under bf16 autocast the glue writes bf16 for the bf16 convolutions
        for i in range(4, -1, -1):
            m0, m1 = self.up0[str(i)][0], self.up1[str(i)][0]
            skip = feat[self.enc_sc.index(2**i)] if (self.use_skip and 2**i in self.enc_sc) else None
            c = conv(m1, HF.elu_up_cat_pad(conv(m0, xp), skip, bias=m0.bias.float(), out_dtype=out_dtype))
            if i in self.out_sc or i > 0: xp = HF.elu_pad(c, bias=m1.bias.float(), apply_elu=True, out_dtype=out_dtype)
            if i in self.out_sc:
                m = self.out[str(i)]
                if self.out_ch == 1 and isinstance(self.act, (nn.Sigmoid, nn.Identity)):   # a one-channel head is a stencil: smd_conv3x3_head_* (fp32 or bf16 activation in, fp32 out)
                    out[i] = HF.conv3x3_head(xp, m.weight.float(), m.bias.float() if m.bias is not None else None, 'sigmoid' if isinstance(self.act, nn.Sigmoid) else None)
                else:
                    out[i] = self.act(F.conv2d(xp, m.weight, m.bias))
        return out
