"""Shared pytest plumbing: the `gpu` marker, golden-fixture loading, and the path to the repo root."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
GOLDEN = ROOT/'tests'/'golden'
if str(ROOT) not in sys.path: sys.path.insert(0, str(ROOT))

if str(GOLDEN) not in sys.path: sys.path.insert(0, str(GOLDEN))       # exact_inputs.py (shared with make_golden.py)

TRAIN_CASES_SMALL = ['train_kbr_24x32', 'train_kbr_96x128', 'train_learnK_n4_40x56', 'train_mean_n1_s1_33x47',
                     'train_min_noauto_25x38', 'train_l1_automask_24x32', 'train_bigmotion_24x32']
# The reference run at the resolutions BASELINE.json quotes (one sample each; compact layout: inputs regenerated from the seed by
# tests/golden/exact_inputs.py, full-resolution maps stored every meta_stride-th pixel, masks bit-packed)
TRAIN_CASES_BASELINE = ['train_kbr_192x640', 'train_learnK_n4_384x640']
TRAIN_CASES = TRAIN_CASES_SMALL + TRAIN_CASES_BASELINE


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The package refuses to import without its HIP library (no fallback path); build it once if the tree is fresh.
    if not (ROOT/'slowtv_monodepth_amd'/'libsmd_hotpath.so').is_file():
        import subprocess
        subprocess.run(['make', '-C', str(ROOT/'slowtv_monodepth_amd'/'csrc'), '-j8'], check=True)


def load_golden(name: str) -> dict:
    """Load a fixture as {key: torch tensor | python scalar}."""
    out = {}
    with np.load(GOLDEN/f'{name}.npz', allow_pickle=False) as z:
        for k in z.files:
            a = z[k]
            if a.dtype.kind in 'US': out[k] = str(a)
            elif k.startswith('meta_') and a.ndim == 0: out[k] = a.item()
            else: out[k] = torch.from_numpy(a.copy())
    if out.get('meta_compact'): expand_compact(name, out)
    return out


def expand_compact(name: str, g: dict) -> None:
    """A compact `train_*` fixture stores no images / disparities / noise: regenerate them from the seed (bit-exact generator,
    tests/golden/exact_inputs.py), REFUSE the fixture if their bit checksums are not the ones recorded when the reference ran on
    them, and unpack the bit-packed boolean maps."""
    from exact_inputs import bit_checksum, make_inputs_exact
    scales = [int(s) for s in g['meta_scales']]
    inp = make_inputs_exact(int(g['meta_seed']), g['meta_b'], g['meta_h'], g['meta_w'], g['meta_n'], scales, learn_K=bool(g['meta_learn_K']))
    for k in ('imgs', 'supp_imgs', 'noise', 'K'):
        assert bit_checksum(inp[k]) == int(g[f'chk_{k}']), f'{name}: regenerated {k} differs from what the reference was run on'
        g[f'in_{k}'] = inp[k]
    for s in scales:
        assert bit_checksum(inp['disp'][s]) == int(g[f'chk_disp_{s}']), f'{name}: regenerated disp[{s}] differs from what the reference was run on'
        g[f'in_disp_{s}'] = inp['disp'][s]
    for k in [k for k in g if k.startswith('bits_')]:
        shape = tuple(int(v) for v in g['shape_' + k[5:]])
        n = int(np.prod(shape))
        g['out_' + k[5:]] = torch.from_numpy(np.unpackbits(g[k].numpy())[:n].astype(bool).reshape(shape))


def ref_map(g: dict, key: str, actual: torch.Tensor):
    """-> (actual', expected) for a float map of a `train_*` fixture: the whole map in the small layout (`out_x` / `mid_x`), every
    `meta_stride`-th pixel in both directions in the compact one (`outs_x` / `mids_x`)."""
    if key in g: return actual, g[key]
    pre, rest = key.split('_', 1)
    st = int(g['meta_stride'])
    return actual[..., ::st, ::st], g[f'{pre}s_{rest}']


def rel_to_max(a, b):
    """max |a - b| relative to max |b| (gradient comparisons: element-wise rtol is meaningless near zero crossings)."""
    return ((a - b).abs().max()/b.abs().max().clamp(min=1e-20)).item()


def case_inputs(g: dict, device='cpu', dtype=torch.float32, requires_grad=True):
    """Rebuild the leaves of a `train_*` fixture.  Returns (leaves, static) dicts of tensors on `device`."""
    scales = [int(s) for s in g['meta_scales']]
    leaves = {f'disp_{s}': g[f'in_disp_{s}'].to(device, dtype).clone() for s in scales}
    leaves['aa'] = g['in_aa'].to(device, dtype).clone()
    leaves['t'] = g['in_t'].to(device, dtype).clone()
    if g['meta_learn_K']:
        leaves['fs'] = g['in_fs'].to(device, dtype).clone()
        leaves['cs'] = g['in_cs'].to(device, dtype).clone()
    if requires_grad:
        for v in leaves.values(): v.requires_grad_(True)
    static = {'imgs': g['in_imgs'].to(device, dtype), 'supp_imgs': g['in_supp_imgs'].to(device, dtype),
              'K': g['in_K'].to(device, dtype), 'noise': g['in_noise'].to(device, dtype) if 'in_noise' in g else None,
              'scales': scales, 'supp_idxs': [int(i) for i in g['meta_supp_idxs']]}
    return leaves, static


@pytest.fixture
def knobs():
    """Pin launch-shape knobs of the library for one test (`smd_set_knob`), restored afterwards.  `knobs(name, value)` -> False if this
    build lacks the knob (experiments-only)."""
    from slowtv_monodepth_amd import _lib
    yield _lib.set_knob
    _lib.reset_knobs()


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def get(name):
        if name not in cache: cache[name] = load_golden(name)
        return cache[name]
    return get


# Observed parity numbers (selection flips, error-map and gradient differences) are part of the test log even when the tests
# pass and output capture is on: a regression from 3 flips to 0.29 % must be visible, not just "still under the limit".
PARITY_REPORT: list[str] = []


def parity_note(line: str) -> None:
    PARITY_REPORT.append(line)
    print(line)


def pytest_terminal_summary(terminalreporter):
    if PARITY_REPORT:
        terminalreporter.section('observed parity numbers')
        for line in PARITY_REPORT: terminalreporter.write_line(line)
