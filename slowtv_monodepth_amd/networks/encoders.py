"""`features_only` image encoders in plain `torch.nn`.

The reference builds its encoders with `timm.create_model(name, features_only=True[, in_chans=6])`
(src/networks/depth.py:95-98, src/networks/pose.py:39-41).  timm is not available on the MI355X image, so the three
families the BASELINE configurations name are restated here with the same stage layout, the same multi-scale feature
list and the same `feature_info.channels()/reduction()` accessors.  Weights are randomly initialised (`pretrained`
is accepted for cfg compatibility; there is no network access to fetch ImageNet weights).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = ['create_encoder', 'FeatureInfo', 'ResNetFeatures', 'ConvNeXtFeatures', 'ENCODERS']


class FeatureInfo:
    def __init__(self, channels, reduction): self._c, self._r = list(channels), list(reduction)
    def channels(self): return list(self._c)
    def reduction(self): return list(self._r)


class BatchNormAct2d(nn.BatchNorm2d):
    """`nn.BatchNorm2d` (same parameters, buffers and state-dict keys) whose call can absorb the residual add and the ReLU
    that follow it in a ResNet block.  On the GPU in fp32 training this is the `smd_bn_*` kernel pair (3 sweeps forward,
    2 backward, for BN + add + ReLU together); anywhere else (CPU, eval, autocast, channels-last) it is the ATen composition."""
    fused_enabled = True   # class-wide switch (tests compare the two evaluations of the same network)

    def forward(self, x, residual=None, relu: bool = False):
        fused = (self.fused_enabled and x.is_cuda and self.training and x.dtype == torch.float32 and not torch.is_autocast_enabled() and self.affine
                 and self.track_running_stats and self.momentum is not None and x.is_contiguous())
        if not fused:
            y = super().forward(x)
            if residual is not None: y = y + residual
            return F.relu(y, inplace=True) if relu else y
        from .. import functional as HF
        self._pending_batches = getattr(self, '_pending_batches', 0) + 1   # num_batches_tracked, materialised lazily (see below)
        return HF.batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var, residual=residual,
                                 momentum=self.momentum, eps=self.eps, relu=relu)

    def flush_counter(self):
        """`num_batches_tracked` only matters with momentum=None (cumulative average), which the fused path excludes; it is
        kept exact but written when someone looks (state_dict / eval) instead of by one tiny launch per layer per step."""
        n = getattr(self, '_pending_batches', 0)
        if n and self.num_batches_tracked is not None: self.num_batches_tracked += n
        self._pending_batches = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self.flush_counter()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def train(self, mode: bool = True):
        if not mode: self.flush_counter()
        return super().train(mode)


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False); self.bn1 = BatchNormAct2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False); self.bn2 = BatchNormAct2d(cout)
        self.down = None
        if stride != 1 or cin != cout:
            self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), BatchNormAct2d(cout))

    def forward(self, x):
        idt = x if self.down is None else self.down(x)
        x = self.bn1(self.conv1(x), relu=True)
        return self.bn2(self.conv2(x), residual=idt, relu=True)


class Bottleneck(nn.Module):
    def __init__(self, cin, mid, stride):
        super().__init__()
        cout = mid*4
        self.conv1 = nn.Conv2d(cin, mid, 1, bias=False); self.bn1 = BatchNormAct2d(mid)
        self.conv2 = nn.Conv2d(mid, mid, 3, stride, 1, bias=False); self.bn2 = BatchNormAct2d(mid)
        self.conv3 = nn.Conv2d(mid, cout, 1, bias=False); self.bn3 = BatchNormAct2d(cout)
        self.down = None
        if stride != 1 or cin != cout:
            self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), BatchNormAct2d(cout))

    def forward(self, x):
        idt = x if self.down is None else self.down(x)
        x = self.bn1(self.conv1(x), relu=True)
        x = self.bn2(self.conv2(x), relu=True)
        return self.bn3(self.conv3(x), residual=idt, relu=True)


class ResNetFeatures(nn.Module):
    """ResNet trunk returning [stem (1/2), layer1 (1/4), layer2 (1/8), layer3 (1/16), layer4 (1/32)]."""
    def __init__(self, layers=(2, 2, 2, 2), bottleneck=False, in_chans=3):
        super().__init__()
        self.conv1 = nn.Conv2d(in_chans, 64, 7, 2, 3, bias=False); self.bn1 = BatchNormAct2d(64)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        widths, exp = (64, 128, 256, 512), (4 if bottleneck else 1)
        stages, cin = [], 64
        for i, (wd, nb) in enumerate(zip(widths, layers)):
            blocks = []
            for j in range(nb):
                stride = 2 if (j == 0 and i > 0) else 1
                blocks.append(Bottleneck(cin, wd, stride) if bottleneck else BasicBlock(cin, wd, stride))
                cin = wd*exp
            stages.append(nn.Sequential(*blocks))
        self.layers = nn.ModuleList(stages)
        self.feature_info = FeatureInfo([64] + [wd*exp for wd in widths], [2, 4, 8, 16, 32])
        for m in self.modules():
            if isinstance(m, nn.Conv2d): nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def forward(self, x):
        x = self.bn1(self.conv1(x), relu=True)
        feats = [x]
        if BatchNormAct2d.fused_enabled and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled() and x.is_contiguous():
            from .. import functional as HF
            x = HF.max_pool3x3s2(x)
        else:
            x = self.maxpool(x)
        for layer in self.layers:
            x = layer(x); feats.append(x)
        return feats


def _use_nchw_kernels(x) -> bool:
    return BatchNormAct2d.fused_enabled and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16, torch.float16)


class LayerNorm2d(nn.LayerNorm):
    """LayerNorm over the channel dim of an NCHW tensor (on the GPU: `smd_layernorm_cf_*`, no permutes)."""
    def forward(self, x):
        if _use_nchw_kernels(x):
            from .. import functional as HF
            with torch.autocast('cuda', enabled=False):
                return HF.layer_norm_cf(x.float().contiguous(), self.weight.float(), self.bias.float(), self.eps)
        return F.layer_norm(x.permute(0, 2, 3, 1), self.normalized_shape, self.weight, self.bias, self.eps).permute(0, 3, 1, 2)


class ConvNeXtBlock(nn.Module):
    def __init__(self, dim, ls_init=1e-6):
        super().__init__()
        self.dw = nn.Conv2d(dim, dim, 7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.fc1 = nn.Linear(dim, 4*dim); self.fc2 = nn.Linear(4*dim, dim)
        self.gamma = nn.Parameter(ls_init*torch.ones(dim))

    def forward(self, x):
        if _use_nchw_kernels(x): return self._forward_nchw(x)
        y = self.dw(x).permute(0, 2, 3, 1)
        y = self.fc2(F.gelu(self.fc1(self.norm(y))))*self.gamma
        return x + y.permute(0, 3, 1, 2)

    def _forward_nchw(self, x):
        """The same block without leaving NCHW (timm's `conv_mlp` formulation on the SAME parameters): depthwise 7x7 stencil
        and channel LayerNorm as HIP kernels in fp32 (autocast runs layer_norm in fp32 anyway), the MLP as two 1x1
        convolutions on views of the Linear weights, layer scale + residual as one addcmul.  No layout copies."""
        from .. import functional as HF
        to_bf16 = torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16   # the 1x1 convolution will run in bf16
        with torch.autocast('cuda', enabled=False):
            y = HF.dwconv7x7(x.float().contiguous(), self.dw.weight.float(), self.dw.bias.float())
            y = HF.layer_norm_cf(y, self.norm.weight.float(), self.norm.bias.float(), self.norm.eps,
                                 out_dtype=torch.bfloat16 if to_bf16 else torch.float32)
        y = F.conv2d(y, self.fc1.weight[:, :, None, None], self.fc1.bias)
        y = F.conv2d(F.gelu(y), self.fc2.weight[:, :, None, None], self.fc2.bias)
        return torch.addcmul(x, y, self.gamma.view(1, -1, 1, 1).to(y.dtype))


class ConvNeXtFeatures(nn.Module):
    """ConvNeXt trunk returning the four stage outputs at strides [4, 8, 16, 32]."""
    def __init__(self, depths=(3, 3, 9, 3), dims=(96, 192, 384, 768), in_chans=3):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(in_chans, dims[0], 4, 4), LayerNorm2d(dims[0], eps=1e-6))
        stages = []
        for i, (d, c) in enumerate(zip(depths, dims)):
            mods = []
            if i > 0: mods += [LayerNorm2d(dims[i - 1], eps=1e-6), nn.Conv2d(dims[i - 1], c, 2, 2)]
            mods += [ConvNeXtBlock(c) for _ in range(d)]
            stages.append(nn.Sequential(*mods))
        self.stages = nn.ModuleList(stages)
        self.feature_info = FeatureInfo(dims, [4, 8, 16, 32])
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None: nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.stem(x)
        feats = []
        for st in self.stages:
            x = st(x); feats.append(x)
        return feats


ENCODERS = {
    'resnet18': lambda c: ResNetFeatures((2, 2, 2, 2), False, c),
    'resnet34': lambda c: ResNetFeatures((3, 4, 6, 3), False, c),
    'resnet50': lambda c: ResNetFeatures((3, 4, 6, 3), True, c),
    'convnext_tiny': lambda c: ConvNeXtFeatures((3, 3, 9, 3), (96, 192, 384, 768), c),
    'convnext_small': lambda c: ConvNeXtFeatures((3, 3, 27, 3), (96, 192, 384, 768), c),
    'convnext_base': lambda c: ConvNeXtFeatures((3, 3, 27, 3), (128, 256, 512, 1024), c),
}


def create_encoder(name: str, in_chans: int = 3, pretrained: bool = False) -> nn.Module:
    """Stand-in for `timm.create_model(name, features_only=True, in_chans=...)`.

    `pretrained=True` (the reference's default for the depth network) cannot be honoured here — timm and its weight hub are not
    available — and silently training from scratch would be a different experiment, so it warns.  ImageNet / reference weights
    can be loaded afterwards from a timm or reference `state_dict` with `networks.checkpoint.load_reference_state_dict`."""
    if name not in ENCODERS: raise KeyError(f'Unknown encoder "{name}". Available: {sorted(ENCODERS)}')
    if pretrained:
        import warnings
        warnings.warn(f'create_encoder({name!r}, pretrained=True): no pretrained weights are available offline, the encoder is randomly initialised; '
                      'load a timm / reference state_dict with slowtv_monodepth_amd.networks.checkpoint.load_reference_state_dict', UserWarning, stacklevel=2)
    return ENCODERS[name](in_chans)
