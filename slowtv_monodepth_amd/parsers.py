"""cfg dict -> networks / losses / optimizer / schedulers (reference: `src/tools/parsers.py`)."""
from __future__ import annotations

from collections import OrderedDict

import torch
from torch import nn, optim

from . import registry as reg

__all__ = ['get_net', 'get_loss', 'get_opt', 'get_sched']


def get_net(cfg: dict) -> nn.ModuleDict:
    """{'depth': {...}, 'pose': {...}} -> ModuleDict of `NET_REG[key](**kwargs)`; `None` entries are skipped
    (src/tools/parsers.py:36-69)."""
    reg.trigger_decoders(); reg.trigger_nets()
    nets = OrderedDict()
    for k, kw in cfg.items():
        if kw is None: continue
        try: nets[k] = reg.NET_REG[k](**kw)
        except NotImplementedError: raise
        except Exception as e: raise ValueError(f'Error using "{k}" in {list(reg.NET_REG)}') from e
    return nn.ModuleDict(nets)


def get_loss(cfg: dict) -> tuple[nn.ModuleDict, nn.ParameterDict]:
    """{'img_recon': {'weight': 1, ...}, ...} -> (losses, frozen weights); pops `weight` (default 1) from each entry and
    skips `None` entries, exactly as src/tools/parsers.py:72-106 (including mutating the cfg it is given)."""
    reg.trigger_losses()
    losses, weights = nn.ModuleDict(), nn.ParameterDict()
    for k, kw in cfg.items():
        if kw is None: continue
        weights[k] = nn.Parameter(torch.as_tensor(kw.pop('weight', 1)), requires_grad=False)
        losses[k] = reg.LOSS_REG[k](**kw)
    return losses, weights


_OPTS = {'adam': optim.Adam, 'adamw': optim.AdamW, 'sgd': optim.SGD, 'rmsprop': optim.RMSprop}


def get_opt(parameters, cfg: dict) -> optim.Optimizer:
    """Optimizer factory with the cfg keys of src/tools/parsers.py:205-243 (`type`|`opt`, `lr`, `weight_decay`,
    `frozen_bn`, `backbone_lr`).  The reference delegates to timm's `create_optimizer_v2`, whose default is to exempt
    biases and 1-d (norm) parameters from weight decay; that rule is restated here on `torch.optim`."""
    cfg = dict(cfg)
    if 'type' in cfg: cfg['opt'] = cfg.pop('type')
    elif 'opt' not in cfg: raise KeyError('Must provide a cfg key `type` or `opt` when instantiating an optimizer.')
    name = cfg.pop('opt').lower()
    if name not in _OPTS: raise KeyError(f'Unknown optimizer "{name}" ({sorted(_OPTS)})')
    is_module = isinstance(parameters, nn.Module)
    if cfg.pop('frozen_bn', False):
        if not is_module: raise ValueError('Cannot freeze batch norm parameters unless given nn.Module')
        for m in parameters.modules():
            if isinstance(m, nn.BatchNorm2d): m.requires_grad_(False)
    blr = cfg.pop('backbone_lr', False)
    wd = cfg.pop('weight_decay', 0.0)
    if blr and not is_module: raise ValueError('Cannot set backbone LR unless given nn.Module')
    if blr and blr == cfg['lr']: raise ValueError('Backbone LR must be different from the main LR')
    if is_module:
        groups = {}
        for n, p in parameters.named_parameters():
            if not p.requires_grad: continue
            no_decay = p.ndim <= 1 or n.endswith('.bias')
            is_bb = bool(blr) and 'encoder' in n
            groups.setdefault((no_decay, is_bb), []).append(p)
        params = []
        for (no_decay, is_bb), ps in groups.items():
            g = {'params': ps, 'weight_decay': 0.0 if no_decay else wd}
            if is_bb: g['lr'] = blr
            params.append(g)
    else:
        params = parameters
        cfg['weight_decay'] = wd
    if name in ('adam', 'adamw') and 'fused' not in cfg and 'foreach' not in cfg:
        ps = [p for g in params for p in g['params']] if is_module else list(params)
        if is_module and ps and all(p.is_cuda and p.is_floating_point() for p in ps):
            cfg['fused'] = True   # same update rule as the default (foreach) implementation, one launch instead of ~10 sweeps
    return _OPTS[name](params, **cfg)


def get_sched(opt: optim.Optimizer, cfg: dict) -> dict:
    """{'steplr': {...}, 'linear': {...}} -> {name: scheduler} from `SCHED_REG` (src/tools/parsers.py:246-269)."""
    out = {}
    for k, kw in cfg.items():
        if kw is None: continue
        if k not in reg.SCHED_REG: raise ValueError(f'Error using "{k}" in {list(reg.SCHED_REG)}')
        out[k] = reg.SCHED_REG[k](opt, **kw)
    return out
