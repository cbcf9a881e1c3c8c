"""YAML config loading with the reference's merge rule (`src/utils/io.py:134-163`): configs are applied left to right,
dictionaries merge recursively, every other value (lists included) replaces the old one."""
from __future__ import annotations

from pathlib import Path

import yaml

__all__ = ['load_yaml', 'load_merge_yaml', 'merge_cfg', 'write_yaml']


def load_yaml(file) -> dict:
    with open(file) as f: return yaml.load(f, Loader=yaml.FullLoader)


def write_yaml(file, data: dict, mkdir: bool = False) -> None:
    file = Path(file)
    if mkdir: file.parent.mkdir(parents=True, exist_ok=True)
    with open(file, 'w') as f: yaml.dump(data, f, sort_keys=False)


def merge_cfg(old: dict, new: dict) -> dict:
    out = dict(old)
    for k, v in new.items():
        out[k] = merge_cfg(out[k], v) if (k in out and isinstance(v, dict) and isinstance(out[k], dict)) else v
    return out


def load_merge_yaml(*files) -> dict:
    """((((cfg1 <- cfg2) <- cfg3) ...) <- cfgN)."""
    cfgs = [load_yaml(f) for f in files]
    out = cfgs[0]
    for new in cfgs[1:]: out = merge_cfg(out, new)
    return out
