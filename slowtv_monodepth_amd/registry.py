"""Name -> class registries for networks, losses, decoders, datasets and predictors.

Mirrors the operator surface of the reference's `src/registry.py:16-138`: `@register(name | (names...), type=None,
overwrite=False)`; the registry is guessed from the class-name suffix (`Net`, `Loss`/`Reg`, `Dataset`, `Pred(ictor)`,
`Dec(oder)`), duplicate keys raise `ValueError`, an unknown `type` raises `TypeError`, and classes defined in
`__main__` are skipped so that running a module as a script does not register it twice.
"""
from __future__ import annotations

import logging

import torch.optim.lr_scheduler as sched

__all__ = ['register', 'NET_REG', 'LOSS_REG', 'DATA_REG', 'SCHED_REG', 'PRED_REG', 'DEC_REG',
           'trigger_nets', 'trigger_losses', 'trigger_decoders']

log = logging.getLogger('slowtv_monodepth_amd.registry')

NET_REG: dict = {}
LOSS_REG: dict = {}
DATA_REG: dict = {}
PRED_REG: dict = {}
DEC_REG: dict = {}
SCHED_REG: dict = {  # src/registry.py:21-28
    'steplr': sched.StepLR, 'exp': sched.ExponentialLR, 'cos': sched.CosineAnnealingLR,
    'cos_warm': sched.CosineAnnealingWarmRestarts, 'plateau': sched.ReduceLROnPlateau, 'linear': sched.LinearLR,
}

_BY_TYPE = {'net': NET_REG, 'loss': LOSS_REG, 'data': DATA_REG, 'pred': PRED_REG, 'dec': DEC_REG}
_SUFFIX = (('Net', 'net'), ('Loss', 'loss'), ('Reg', 'loss'), ('Dataset', 'data'), ('Pred', 'pred'), ('Predictor', 'pred'),
           ('Dec', 'dec'), ('Decoder', 'dec'))


def _type_of(cls) -> str:
    for suffix, kind in _SUFFIX:
        if cls.__name__.endswith(suffix): return kind
    raise ValueError(f'Class matched no known patterns. ({cls.__name__} vs. {sorted({s for s, _ in _SUFFIX})})')


def register(name, type: str | None = None, overwrite: bool = False):
    """Class decorator adding `cls` under `name` (str or tuple of str) to the registry `type` (guessed if None)."""
    names = (name,) if isinstance(name, str) else tuple(name)

    def deco(cls):
        if cls.__module__ == '__main__':
            log.warning("Ignoring class '%s' created in the '__main__' module.", cls.__name__)
            return cls
        kind = type or _type_of(cls)
        if kind not in _BY_TYPE: raise TypeError(f'Invalid `type`. ({kind} vs. {set(_BY_TYPE)})')
        table = _BY_TYPE[kind]
        for key in names:
            if not overwrite and key in table:
                raise ValueError(f"'{key}' already in '{kind}' registry ({table[key]} vs. {cls}). Set `overwrite=True` to overwrite.")
            table[key] = cls
        return cls
    return deco


def trigger_nets() -> None:
    from . import networks  # noqa: F401


def trigger_decoders() -> None:
    from .networks import decoders  # noqa: F401


def trigger_losses() -> None:
    from . import losses, regularizers  # noqa: F401
