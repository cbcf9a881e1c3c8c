"""Parameter-name bridge to the reference's checkpoints.

The networks here restate the reference's architectures with their own module layout (`networks/encoders.py` stands in for timm,
`networks/decoders.py` uses `up0 / up1 / out` dictionaries), so a reference `state_dict` — Lightning's `ckpt['state_dict']` with
the `nets.<key>.` prefix (`src/core/trainer.py:58-60`), or the bare module's — needs its keys translated:

    reference / timm name                               here
    decoders.disp.decoder.{2(4-i)+j}.conv.{w,b}         decoders.disp.up{j}.{i}.0.{w,b}        (monodepth.py:50-66: ModuleList order)
    decoders.disp.decoder.{10+k}.{w,b}                  decoders.disp.out.{out_sc[k]}.{w,b}
    encoder.layer{L}.{B}.downsample.{0,1}.*             encoder.layers.{L-1}.{B}.down.{0,1}.*  (timm ResNet, features_only)
    encoder.stem_0 / stem_1 (or stem.0 / stem.1)        encoder.stem.0 / stem.1                (timm ConvNeXt, features_only)
    encoder.stages_{S}.downsample.{0,1}.*               encoder.stages.{S}.{0,1}.*
    encoder.stages_{S}.blocks.{B}.{conv_dw,norm,mlp.fc1,mlp.fc2,gamma}
                                                        encoder.stages.{S}.{B+2*(S>0)}.{dw,norm,fc1,fc2,gamma}

`to_reference_state_dict` is the inverse; `load_reference_state_dict` / `reference_checkpoint` use them.  The decoder half is
checked against the reference's own `MonodepthDecoder` in tests/test_reference_boundary.py (build container); timm is not in
this image, so the encoder half is pinned by a round-trip test and by timm's published module names.
"""
from __future__ import annotations

import re

import torch
import torch.nn as nn

__all__ = ['from_reference_key', 'to_reference_key', 'load_reference_state_dict', 'to_reference_state_dict', 'reference_checkpoint', 'load_reference_checkpoint']


def _decoder_from_ref(rest: str, out_sc):
    m = re.fullmatch(r'decoder\.(\d+)\.(?:conv\.)?(weight|bias)', rest)
    if not m: return None
    idx, leaf = int(m.group(1)), m.group(2)
    if idx < 10: return f'up{idx % 2}.{4 - idx//2}.0.{leaf}'
    return f'out.{out_sc[idx - 10]}.{leaf}'


def _decoder_to_ref(rest: str, out_sc):
    m = re.fullmatch(r'up([01])\.(\d)\.0\.(weight|bias)', rest)
    if m: return f'decoder.{2*(4 - int(m.group(2))) + int(m.group(1))}.conv.{m.group(3)}'
    m = re.fullmatch(r'out\.(\d)\.(weight|bias)', rest)
    if m: return f'decoder.{10 + list(out_sc).index(int(m.group(1)))}.{m.group(2)}'
    return None


_BLOCK_LEAF = {'conv_dw': 'dw', 'norm': 'norm', 'mlp.fc1': 'fc1', 'mlp.fc2': 'fc2', 'gamma': 'gamma'}
_BLOCK_LEAF_INV = {v: k for k, v in _BLOCK_LEAF.items()}


def _encoder_from_ref(rest: str):
    m = re.fullmatch(r'layer(\d)\.(\d+)\.(.*)', rest)                      # timm ResNet
    if m: return f'layers.{int(m.group(1)) - 1}.{m.group(2)}.' + re.sub(r'^downsample\.', 'down.', m.group(3))
    m = re.fullmatch(r'stem[._](\d)\.(.*)', rest)                           # timm ConvNeXt
    if m: return f'stem.{m.group(1)}.{m.group(2)}'
    m = re.fullmatch(r'stages[._](\d)\.downsample\.(\d)\.(.*)', rest)
    if m: return f'stages.{m.group(1)}.{m.group(2)}.{m.group(3)}'
    m = re.fullmatch(r'stages[._](\d)\.blocks\.(\d+)\.(conv_dw|norm|mlp\.fc1|mlp\.fc2|gamma)(\..*)?', rest)
    if m:
        s = int(m.group(1))
        return f'stages.{s}.{int(m.group(2)) + (2 if s > 0 else 0)}.{_BLOCK_LEAF[m.group(3)]}{m.group(4) or ""}'
    return rest                                                             # conv1 / bn1 of the ResNet stem: same names


def _encoder_to_ref(rest: str, convnext: bool):
    if not convnext:
        m = re.fullmatch(r'layers\.(\d)\.(\d+)\.(.*)', rest)
        if m: return f'layer{int(m.group(1)) + 1}.{m.group(2)}.' + re.sub(r'^down\.', 'downsample.', m.group(3))
        return rest
    m = re.fullmatch(r'stem\.(\d)\.(.*)', rest)
    if m: return f'stem_{m.group(1)}.{m.group(2)}'
    m = re.fullmatch(r'stages\.(\d)\.(\d+)\.(.*)', rest)
    if m:
        s, j, leaf = int(m.group(1)), int(m.group(2)), m.group(3)
        if s > 0 and j < 2: return f'stages_{s}.downsample.{j}.{leaf}'
        head, _, tail = leaf.partition('.')
        return f'stages_{s}.blocks.{j - (2 if s > 0 else 0)}.{_BLOCK_LEAF_INV[head]}' + (f'.{tail}' if tail else '')
    return rest


def _split(key: str):
    """'(prefix)(encoder.|decoders.<name>.)(rest)' -> (prefix, part, rest); part is None for keys that need no translation."""
    m = re.match(r'(.*?)(encoder\.|decoders\.(?:disp|mask)\.)(.*)', key)
    return (m.group(1), m.group(2), m.group(3)) if m else (key, None, '')


def from_reference_key(key: str, out_sc=(0, 1, 2, 3)) -> str:
    prefix, part, rest = _split(key)
    if part is None: return key
    if part.startswith('decoders.'):
        new = _decoder_from_ref(rest, list(out_sc))
        return prefix + part + (new if new is not None else rest)
    return prefix + part + _encoder_from_ref(rest)


def to_reference_key(key: str, out_sc=(0, 1, 2, 3), convnext: bool = False) -> str:
    prefix, part, rest = _split(key)
    if part is None: return key
    if part.startswith('decoders.'):
        new = _decoder_to_ref(rest, list(out_sc))
        return prefix + part + (new if new is not None else rest)
    return prefix + part + _encoder_to_ref(rest, convnext)


def _out_sc(module: nn.Module):
    for m in module.modules():
        if hasattr(m, 'out_sc') and hasattr(m, 'up0'): return list(m.out_sc)
    return [0, 1, 2, 3]


def _is_convnext(module: nn.Module, key: str) -> bool:
    """Is the encoder that owns `key` a ConvNeXt trunk?  (a DepthNet and a PoseNet of one trainer may differ)"""
    if 'encoder.' not in key: return False
    owner_path = key.split('encoder.')[0].rstrip('.')
    try: owner = module.get_submodule(owner_path) if owner_path else module
    except AttributeError: return False
    return hasattr(getattr(owner, 'encoder', None), 'stages')


def load_reference_state_dict(module: nn.Module, state_dict: dict, strict: bool = True):
    """Load a reference `state_dict` (of the same kind of module: a network, the `nets` ModuleDict or the whole trainer)."""
    out_sc = _out_sc(module)
    return module.load_state_dict({from_reference_key(k, out_sc): v for k, v in state_dict.items()}, strict=strict)


def to_reference_state_dict(module: nn.Module) -> dict:
    out_sc = _out_sc(module)
    return {to_reference_key(k, out_sc, _is_convnext(module, k)): v for k, v in module.state_dict().items()}


def reference_checkpoint(trainer: nn.Module, epoch: int = 0, global_step: int = 0, optimizer=None, scheduler=None,
                         lightning_version: str = '2.0.1') -> dict:
    """A checkpoint dict in the layout of the reference's Lightning checkpoints: `state_dict` keyed as its LightningModule
    (`nets.<key>.…`, `weights.<loss>`) — what `MonoDepthModule.load_from_checkpoint` and every `state_dict` consumer read — plus
    the bookkeeping Lightning's restore path indexes: `epoch`, `global_step`, `optimizer_states`, `lr_schedulers`
    (`restore_lr_schedulers` reads `ckpt['lr_schedulers']`), `loops` and `callbacks` (empty: this package's loop has no Lightning
    loop / callback state to hand over, so a Lightning `fit(ckpt_path=...)` restarts its progress counters from `epoch` /
    `global_step`).  `lightning_version` defaults to the reference's pin (docker/environment.yml).  The optimizer state is
    `torch.optim` state as is: it round-trips through this package's `--resume`; loading it into the reference additionally
    needs its `get_opt` to create the parameter groups in the same order (timm's `create_optimizer_v2` there, not checked here)."""
    ckpt = {'state_dict': to_reference_state_dict(trainer), 'epoch': int(epoch), 'global_step': int(global_step),
            'pytorch-lightning_version': str(lightning_version), 'hyper_parameters': {'cfg': getattr(trainer, 'cfg', None)},
            'lr_schedulers': [scheduler.state_dict()] if scheduler is not None else [], 'loops': {}, 'callbacks': {}}
    if optimizer is not None: ckpt['optimizer_states'] = [optimizer.state_dict()]
    return ckpt


def load_reference_checkpoint(trainer: nn.Module, ckpt: dict | str, strict: bool = True):
    if not isinstance(ckpt, dict): ckpt = torch.load(ckpt, map_location='cpu', weights_only=False)
    return load_reference_state_dict(trainer, ckpt['state_dict'] if 'state_dict' in ckpt else ckpt, strict=strict)
