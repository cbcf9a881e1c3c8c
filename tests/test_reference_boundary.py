"""The drop-in boundary, executed: INTEGRATION.md §1 applied VERBATIM to the stub-imported reference.

Runs only where the reference checkout exists (the build container; `/root/reference` is absent on the GPU box) and in a
subprocess, so that the import shim and the reference's `src` package never leak into the test session.  No GPU is needed:
the fused operators are replaced by recorders, because the point is the plumbing — registry keys, constructor and call
signatures, and that the reference's own `MonoDepthModule.forward_loss` lands in this package's handlers."""
import json
import re
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
REF = Path('/root/reference')

DRIVER = r'''
import importlib.abc, importlib.machinery, inspect, json, sys, types
REF, ROOT, SNIPPET = sys.argv[1], sys.argv[2], sys.argv[3]
ABSENT = ('cv2', 'skimage', 'kornia', 'timm', 'torchmetrics', 'pytorch_lightning', 'wandb', 'lmdb', 'h5py', 'torchvision', 'lightning',
          'tensorboard', 'albumentations', 'mmcv')
class Placeholder(types.ModuleType):
    __path__ = []
    def __getattr__(self, name):
        if name.startswith('__'): raise AttributeError(name)
        cls = type(name, (), {'__init__': lambda self, *a, **k: None, '__class_getitem__': classmethod(lambda c, i: c)})
        setattr(self, name, cls)
        return cls
class Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split('.')[0] in ABSENT: return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
    def create_module(self, spec): return Placeholder(spec.name)
    def exec_module(self, module): pass
sys.meta_path.insert(0, Finder())
sys.path.insert(0, REF); sys.path.insert(0, ROOT)
import torch
import src                                               # the reference
import src.core.handlers as ref_h
from src import LOSS_REG
from src.core.trainer import MonoDepthModule
from src.losses import ReconstructionLoss as RefRecon
from src.regularizers import SmoothReg as RefSmooth
from src.tools import ViewSynth as RefSynth, parsers, to_scaled
ref_sigs = {k: inspect.signature(getattr(ref_h, k)) for k in ('image_recon', 'disp_smooth', 'feat_recon', 'stereo_const', 'depth_regr', 'autoenc_recon')}
ref_sigs.update({'ReconstructionLoss.__init__': inspect.signature(RefRecon.__init__), 'ReconstructionLoss.forward': inspect.signature(RefRecon.forward),
                 'SmoothReg.__init__': inspect.signature(RefSmooth.__init__), 'SmoothReg.forward': inspect.signature(RefSmooth.forward),
                 'ViewSynth.__init__': inspect.signature(RefSynth.__init__), 'ViewSynth.forward': inspect.signature(RefSynth.forward)})

exec(compile(open(SNIPPET).read(), 'INTEGRATION.md#1', 'exec'))      # <- the integration snippet, verbatim

import slowtv_monodepth_amd as amd
from slowtv_monodepth_amd import functional as F, handlers as amd_h
from slowtv_monodepth_amd.geometry import ViewSynth as AmdSynth
out = {'registry': {k: LOSS_REG[k].__module__ for k in ('img_recon', 'disp_smooth')}, 'handlers': {k: getattr(ref_h, k).__module__ for k in ('image_recon', 'disp_smooth')}}

def compare(name, ref_sig, ours):
    """every parameter of the reference, in order, with the same name, kind and default; ours may only ADD keyword-only options"""
    rp, op = list(ref_sig.parameters.values()), list(inspect.signature(ours).parameters.values())
    problems = []
    for i, r in enumerate(rp):
        if i >= len(op) or op[i].name != r.name or op[i].kind != r.kind: problems.append(f'{name}: parameter {i} is {op[i].name if i < len(op) else None!r}, reference has {r.name!r}')
        elif (r.default is inspect.Parameter.empty) != (op[i].default is inspect.Parameter.empty) or (r.default is not inspect.Parameter.empty and r.default != op[i].default):
            problems.append(f'{name}: default of {r.name!r} is {op[i].default!r}, reference has {r.default!r}')
    for extra in op[len(rp):]:
        if extra.kind is not inspect.Parameter.KEYWORD_ONLY and extra.default is inspect.Parameter.empty: problems.append(f'{name}: extra required parameter {extra.name!r}')
    return problems
problems = []
for k in ('image_recon', 'disp_smooth', 'feat_recon', 'stereo_const', 'depth_regr', 'autoenc_recon'): problems += compare(k, ref_sigs[k], getattr(amd_h, k))
problems += compare('ReconstructionLoss.__init__', ref_sigs['ReconstructionLoss.__init__'], amd.losses.ReconstructionLoss.__init__)
problems += compare('ReconstructionLoss.forward', ref_sigs['ReconstructionLoss.forward'], amd.losses.ReconstructionLoss.forward)
problems += compare('SmoothReg.__init__', ref_sigs['SmoothReg.__init__'], amd.regularizers.SmoothReg.__init__)
problems += compare('SmoothReg.forward', ref_sigs['SmoothReg.forward'], amd.regularizers.SmoothReg.forward)
problems += compare('ViewSynth.__init__', ref_sigs['ViewSynth.__init__'], AmdSynth.__init__)
problems += compare('ViewSynth.forward', ref_sigs['ViewSynth.forward'], AmdSynth.forward)
out['signature_problems'] = problems

# ---- dispatch: the reference's own forward_loss, unbound, with the losses built by the reference's parser from a reference cfg
calls = []
def fake_recon(stacked, imgs, supp, Ts, Ks, K_inv=None, *, flags, noise=None, seed=0, want_warp=False, want_err=True, prepared=None):
    calls.append(('image_recon_fused', tuple(stacked.shape), flags))
    S, b = stacked.shape[:2]; h, w = imgs.shape[-2:]
    return stacked.mean(), torch.zeros(S, b, 1, h, w), torch.zeros(S, b, 1, h, w, dtype=torch.uint8), torch.zeros_like(supp)
def fake_smooth(disps, img, use_edges=False, want_aux=True, use_laplacian=False, prepared=None):
    calls.append(('disp_smooth_fused', len(disps), bool(use_edges)))
    d0 = disps[min(disps)]
    return sum(d.mean() for d in disps.values()), torch.zeros_like(d0), torch.zeros_like(d0)
F.image_recon_fused, F.disp_smooth_fused = fake_recon, fake_smooth
F.inv_intrinsics = lambda K: torch.linalg.inv(K)
cfg = {'img_recon': {'weight': 1, 'loss_name': 'ssim', 'use_min': True, 'use_automask': True}, 'disp_smooth': {'weight': 0.001, 'use_edges': True}}
losses, weights = parsers.get_loss(dict(cfg))
out['loss_classes'] = {k: type(v).__module__ for k, v in losses.items()}
b, h, w, S = 2, 16, 24, 4
from src.utils import MultiLevelTimer
ns = types.SimpleNamespace(losses=losses, weights=weights, synth=RefSynth((h, w)), timer=MultiLevelTimer(name='x'), to_depth=lambda d: to_scaled(d, 0.1, 100)[1])
fwd = {'disp': {s: torch.rand(b, 1, h >> s, w >> s) for s in range(S)}, 'T_-1': torch.eye(4).repeat(b, 1, 1), 'T_1': torch.eye(4).repeat(b, 1, 1)}
x = {'imgs': torch.rand(b, 3, h, w), 'supp_idxs': torch.tensor([-1, 1])}
y = {'imgs': torch.rand(b, 3, h, w), 'supp_imgs': torch.rand(2, b, 3, h, w), 'K': torch.eye(4).repeat(b, 1, 1)}
fwd = MonoDepthModule.forward_postprocess(ns, fwd, x, y)
loss, ld = MonoDepthModule.forward_loss(ns, fwd, x, y)
out['calls'] = calls
out['loss_dict_keys'] = sorted(ld)

# ---- a reference MonodepthDecoder state_dict loads into this package's decoder through the key bridge, same outputs
from src.networks.decoders.monodepth import MonodepthDecoder as RefDec
from slowtv_monodepth_amd.networks.decoders import MonodepthDecoder as AmdDec
from slowtv_monodepth_amd.networks import checkpoint as ck
torch.manual_seed(0)
kw = dict(num_ch_enc=[64, 64, 128, 256, 512], enc_sc=[2, 4, 8, 16, 32], out_sc=[0, 2, 3], out_ch=1, out_act='sigmoid')
ref_dec, amd_dec = RefDec(**kw), AmdDec(**kw)
wrap_ref, wrap_amd = torch.nn.ModuleDict({'disp': ref_dec}), torch.nn.ModuleDict({'disp': amd_dec})
holder_ref, holder_amd = torch.nn.Module(), torch.nn.Module()
holder_ref.decoders, holder_amd.decoders = wrap_ref, wrap_amd
missing = ck.load_reference_state_dict(holder_amd, holder_ref.state_dict(), strict=True)
feats = [torch.rand(1, c, 64 >> i, 96 >> i) for i, c in enumerate(kw['num_ch_enc'], start=1)]
with torch.no_grad(): ro, ao = ref_dec(feats), amd_dec(feats)
out['decoder_max_diff'] = max((ro[k] - ao[k]).abs().max().item() for k in ro)
out['decoder_keys_back'] = sorted(ck.to_reference_state_dict(holder_amd)) == sorted(holder_ref.state_dict())
print('RESULT ' + json.dumps(out))
'''


@pytest.mark.skipif(not REF.is_dir(), reason='needs the reference checkout (build container only)')
def test_integration_snippet_plugs_into_the_reference(tmp_path):
    text = (ROOT/'INTEGRATION.md').read_text()
    blocks = re.findall(r'```python\n(.*?)```', text, flags=re.S)
    assert blocks and 'register(' in blocks[0], 'INTEGRATION.md must open with the registry-level snippet'
    snippet = tmp_path/'integration_snippet.py'
    snippet.write_text(blocks[0])
    driver = tmp_path/'driver.py'
    driver.write_text(DRIVER)
    proc = subprocess.run([sys.executable, str(driver), str(REF), str(ROOT), str(snippet)], capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    out = json.loads(next(l for l in proc.stdout.splitlines() if l.startswith('RESULT '))[7:])
    # registry keys and handler names now resolve to this package
    assert all(m.startswith('slowtv_monodepth_amd') for m in out['registry'].values()), out['registry']
    assert all(m.startswith('slowtv_monodepth_amd') for m in out['handlers'].values()), out['handlers']
    assert all(m.startswith('slowtv_monodepth_amd') for m in out['loss_classes'].values()), out['loss_classes']
    # same call surface as the reference
    assert not out['signature_problems'], '\n'.join(out['signature_problems'])
    # the reference's forward_loss reached the fused operators, with the flags of the reference cfg (min | automask = 3)
    names = [c[0] for c in out['calls']]
    assert names == ['image_recon_fused', 'disp_smooth_fused'], out['calls']
    assert out['calls'][0][1][0] == 4 and out['calls'][0][2] == 3 and out['calls'][1][1:] == [4, True], out['calls']
    for k in ('loss_img_recon', 'loss_disp_smooth', 'automask', 'supp_imgs_warp', 'disp_grad', 'image_grad'): assert k in out['loss_dict_keys'], out['loss_dict_keys']
    # a reference decoder checkpoint loads through networks/checkpoint.py and computes the same thing
    assert out['decoder_max_diff'] < 1e-6 and out['decoder_keys_back'], (out['decoder_max_diff'], out['decoder_keys_back'])
