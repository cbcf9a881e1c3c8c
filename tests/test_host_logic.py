"""CPU tests of the host side: registry / cfg surface, pose + intrinsics prologue, networks' output contract, the
training step driven through the oracle backend, and the C-ABI library's symbol table (no compute calls without a GPU)."""
import copy
import os
import re
from pathlib import Path

import pytest
import torch

from conftest import ROOT, case_inputs
from oracle import view_synth_oracle as O
from oracle.backend import OracleBackend

import slowtv_monodepth_amd as pkg  # noqa: F401
from slowtv_monodepth_amd import _lib, geometry, io, ops, parsers, registry
from slowtv_monodepth_amd.handlers import ScaleDict
from slowtv_monodepth_amd.losses import ReconstructionLoss
from slowtv_monodepth_amd.networks import DepthNet, MonodepthDecoder, PoseNet, create_encoder
from slowtv_monodepth_amd.regularizers import SmoothReg
from slowtv_monodepth_amd.synthetic import make_batch
from slowtv_monodepth_amd.trainer import MonoDepthModule


# ------------------------------------------------------------------------------------------------- C ABI
def test_library_exports_every_declared_symbol():
    header = (ROOT/'include'/'smd_hotpath.h').read_text()
    declared = set(re.findall(r'\b(smd_[a-z0-9_]+)\s*\(', header))
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(_lib.lib, name), f'{name} declared in include/smd_hotpath.h but not exported by {_lib.lib_path}'
        assert name in _lib.PROTOTYPES, f'{name} has no ctypes prototype'
    assert set(_lib.PROTOTYPES) <= declared
    # every ctypes prototype has exactly the header's parameter count, and pointer / integer / float kinds in the same places (ctypes accepts MORE
    # arguments than `argtypes` lists and passes the surplus as 32-bit ints: a prototype one pointer short once truncated the stream handle)
    import ctypes as C
    code = re.sub(r'/\*.*?\*/', ' ', header, flags=re.S)
    for name, params in re.findall(r'\b(smd_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;', code, flags=re.S):
        params = [q.strip() for q in params.replace('\n', ' ').split(',')]
        if params == ['void']: params = []
        kinds = ['p' if '*' in q else ('f' if re.match(r'(const\s+)?float\b', q) else 'i') for q in params]
        proto = _lib.PROTOTYPES[name][1]
        got = ['p' if t in (C.c_void_p, C.c_char_p) else ('f' if t is C.c_float else 'i') for t in proto]
        assert got == kinds, f'{name}: ctypes prototype {got} does not match the header {kinds}'
    assert _lib.lib.smd_abi_version() == 8
    # launch-shape knobs (the parity tests' pins): known name, unknown name, experiments-only name in the product build; and nothing reads the environment
    assert _lib.set_knob('fwd_rh', 12) is True and _lib.set_knob('bwd_pair', 1) is False
    with pytest.raises(ValueError): _lib.set_knob('no_such_knob', 1)
    _lib.reset_knobs()
    import subprocess
    syms = subprocess.run(['nm', '-D', '--undefined-only', str(_lib.lib_path)], capture_output=True, text=True).stdout
    assert 'getenv' not in syms, 'the product library must not read the environment (knobs: smd_set_knob; experiments: make EXPERIMENTS=1)'
    assert _lib.lib.smd_image_recon_workspace_bytes(12, 2, 4, 192, 640) > 2*12*4*12*4   # at least the pose partials
    assert _lib.lib.smd_image_recon_workspace_bytes(0, 2, 4, 192, 640) == 0
    # image part (texels + target pixels + two window-term planes) + the tail: K0 row table for SMD_MAX_SCALES pyramid levels + 1 + b arrival counters (padded to 16 B),
    # padded to 256 B, + the liveness table the forward leaves for the backward: b ints (padded to 256 B) + SMD_MAX_SCALES x b x (strips of >= 4 rows) x 4 supports x 8 B
    head = 2*12*193*641*12 + 12*192*640*(12 + 32) + 8*(192 + 4)*16 + 16*4
    assert _lib.lib.smd_packed_supports_bytes(12, 2, 192, 640) == (head + 255)//256*256 + 256 + 8*12*(11*48)*4*8


def test_abi_rejects_bad_arguments_without_touching_the_gpu():
    # argument validation happens before any launch, so these are safe on a machine without a GPU
    rc = _lib.lib.smd_image_recon_fwd(*([None]*7), 0, *([None]*6), 0, 1, 1, 1, 1, 1, 0, None)
    assert rc == -1 and b'invalid sizes' in _lib.lib.smd_last_error()
    rc = _lib.lib.smd_image_recon_fwd(*([None]*7), 0, *([None]*6), 0, 2, 2, 4, 8, 8, 0, None)
    assert rc == -1 and b'null pointer' in _lib.lib.smd_last_error()
    with pytest.raises(ValueError): _lib.call('smd_debug_lane_shift', None, None, None)


def test_hip_path_refuses_cpu_tensors():
    from slowtv_monodepth_amd import functional as F
    with pytest.raises(RuntimeError, match='no CPU implementation'):
        F.disp_to_depth([torch.rand(1, 1, 4, 4)], (4, 4), 0.1, 100)
    with pytest.raises(RuntimeError, match='no CPU implementation'):
        SmoothReg(use_edges=True)(torch.rand(1, 1, 4, 4), torch.rand(1, 3, 4, 4))


# ------------------------------------------------------------------------------------------------- registry / cfg
def test_registry_semantics():
    assert {'img_recon', 'disp_smooth'} <= (registry.trigger_losses() or set(registry.LOSS_REG))
    registry.trigger_nets(); registry.trigger_decoders()
    assert {'depth', 'pose'} <= set(registry.NET_REG) and 'monodepth' in registry.DEC_REG
    assert registry.LOSS_REG['img_recon'] is ReconstructionLoss and registry.LOSS_REG['disp_smooth'] is SmoothReg

    @registry.register(('tmp_a', 'tmp_b'))
    class TmpLoss(torch.nn.Module): pass
    assert registry.LOSS_REG['tmp_a'] is TmpLoss and registry.LOSS_REG['tmp_b'] is TmpLoss
    with pytest.raises(ValueError, match='already in'):
        @registry.register('tmp_a')
        class OtherLoss(torch.nn.Module): pass

    @registry.register('tmp_a', overwrite=True)
    class ThirdLoss(torch.nn.Module): pass
    assert registry.LOSS_REG['tmp_a'] is ThirdLoss
    with pytest.raises(ValueError, match='no known patterns'):
        @registry.register('x')
        class Nameless(torch.nn.Module): pass
    with pytest.raises(TypeError):
        @registry.register('x', type='bogus')
        class SomeNet(torch.nn.Module): pass
    for k in ('tmp_a', 'tmp_b'): registry.LOSS_REG.pop(k)


def test_yaml_merge_rule(tmp_path):
    (tmp_path/'a.yaml').write_text('net:\n  depth: {enc_name: resnet18, out_scales: [0, 1, 2, 3]}\n  pose: {enc_name: resnet18}\nloss:\n  img_recon: {weight: 1}\n')
    (tmp_path/'b.yaml').write_text('net:\n  depth: {out_scales: [0]}\n  pose: ~\nloss:\n  disp_smooth: {weight: 0.001, use_edges: True}\n')
    cfg = io.load_merge_yaml(tmp_path/'a.yaml', tmp_path/'b.yaml')
    assert cfg['net']['depth'] == {'enc_name': 'resnet18', 'out_scales': [0]}     # dicts merge, lists replace
    assert cfg['net']['pose'] is None                                              # None disables an entry
    assert set(cfg['loss']) == {'img_recon', 'disp_smooth'}


def test_parsers_build_losses_nets_optimizer():
    cfg = {'img_recon': {'weight': 1, 'use_min': True, 'use_automask': True}, 'disp_smooth': {'weight': 0.001, 'use_edges': True}, 'feat_peaky': None}
    losses, weights = parsers.get_loss(cfg)
    assert list(losses) == ['img_recon', 'disp_smooth'] and 'weight' not in cfg['img_recon']   # `weight` is popped (reference quirk)
    assert weights['disp_smooth'].item() == pytest.approx(0.001) and not weights['disp_smooth'].requires_grad
    assert losses['img_recon'].use_min and losses['img_recon'].use_automask and losses['disp_smooth'].use_edges
    with pytest.raises(KeyError): parsers.get_loss({'nope': {}})
    with pytest.raises(ValueError): ReconstructionLoss(mask_name='bogus')
    assert ReconstructionLoss(mask_name='explainability').mask_name == 'explainability'   # predictive masks run on the un-fused operators
    assert SmoothReg(use_laplacian=True).use_laplacian
    with pytest.raises(NotImplementedError): SmoothReg(use_blur=True, use_laplacian=True)   # a blur between the two differences: not built
    assert SmoothReg(use_blur=True).use_blur                                               # first-order form: kornia's 3x3 Gaussian restated (parity unpinned)
    with pytest.raises(ValueError, match="original 'source'"):
        ReconstructionLoss(use_automask=True)(torch.rand(2, 1, 3, 4, 4), torch.rand(1, 3, 4, 4))

    nets = parsers.get_net({'depth': {'enc_name': 'resnet18', 'pretrained': False}, 'pose': {'enc_name': 'resnet18', 'learn_K': True}, 'autoencoder': None})
    assert list(nets) == ['depth', 'pose']
    opt = parsers.get_opt(nets, {'type': 'adamw', 'lr': 1e-4, 'weight_decay': 1e-3})
    decays = {g['weight_decay'] for g in opt.param_groups}
    assert decays == {0.0, 1e-3}                                                   # biases / norm parameters are exempt
    assert sum(len(g['params']) for g in opt.param_groups) == sum(1 for p in nets.parameters() if p.requires_grad)
    sch = parsers.get_sched(opt, {'steplr': {'step_size': 40, 'gamma': 0.1}, 'linear': {'start_factor': 0.1, 'total_iters': 4}})
    assert set(sch) == {'steplr', 'linear'}
    with pytest.raises(KeyError): parsers.get_opt(nets, {'lr': 1e-4})


# ------------------------------------------------------------------------------------------------- prologue
def test_pose_and_intrinsics_prologue_match_reference(golden):
    g = golden('op_T_from_AAt')
    aa = g['in_aa'].clone().requires_grad_(True); t = g['in_t'].clone().requires_grad_(True)
    T = geometry.T_from_AAt(aa, t)
    torch.testing.assert_close(T, g['out_T'], rtol=1e-5, atol=1e-6)
    (T*g['in_gT']).sum().backward()
    torch.testing.assert_close(aa.grad, g['grad_aa'], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(t.grad, g['grad_t'], rtol=1e-6, atol=1e-7)
    with pytest.raises(ValueError): geometry.T_from_AAt(torch.rand(2, 4), torch.rand(2, 3))

    g = golden('train_learnK_n4_40x56')
    K = geometry.resize_K(geometry.build_K(g['in_fs'], g['in_cs']), (g['meta_h'], g['meta_w']))
    torch.testing.assert_close(K, g['out_K'], rtol=1e-6, atol=1e-6)

    g = golden('op_to_depth')
    sd, dep = geometry.to_scaled(g['in_disp'], 0.1, 100)
    torch.testing.assert_close(sd, g['out_scaled_disp']); torch.testing.assert_close(dep, g['out_depth'])
    torch.testing.assert_close(geometry.to_inv(g['in_disp']), g['out_inv'])
    with pytest.raises(ValueError): geometry.to_scaled(g['in_disp'], 0.0)

    x = torch.rand(2, 3, 4, 5)
    assert ops.expand_dim(x, 7, dim=1, insert=True).shape == (2, 7, 3, 4, 5)
    assert ops.expand_dim(torch.rand(1, 1, 1), num=(5, 3), dim=(0, 1), insert=True).shape == (5, 3, 1, 1, 1)
    torch.testing.assert_close(ops.mean_normalize(x).mean(dim=(2, 3)), torch.ones(2, 3))
    torch.testing.assert_close(ops.unstandardize(ops.standardize(x)), x, rtol=1e-5, atol=1e-6)
    assert ops.eps() == pytest.approx(1.1920929e-07)


# ------------------------------------------------------------------------------------------------- networks
@pytest.mark.parametrize('enc,chs,red', [('resnet18', [64, 64, 128, 256, 512], [2, 4, 8, 16, 32]),
                                        ('convnext_tiny', [96, 192, 384, 768], [4, 8, 16, 32])])
def test_network_output_contract(enc, chs, red):
    e = create_encoder(enc, in_chans=6)
    assert e.feature_info.channels() == chs and e.feature_info.reduction() == red
    feats = e(torch.rand(1, 6, 64, 96))
    assert [f.shape[1] for f in feats] == chs and [64//f.shape[2] for f in feats] == red
    d = DepthNet(enc, pretrained=False, out_scales=[0, 1, 2, 3])
    out = d(torch.rand(2, 3, 64, 96))
    assert list(out['disp']) == [0, 1, 2, 3]
    for s, v in out['disp'].items():
        assert v.shape == (2, 1, 64 >> s, 96 >> s) and v.min() > 0 and v.max() < 1
    p = PoseNet(enc, learn_K=True)(torch.rand(3, 6, 64, 96))
    assert p['R'].shape == (3, 2, 3) and p['t'].shape == (3, 2, 3) and p['fs'].shape == (3, 2) and (p['fs'] > 0).all() and (p['cs'] < 1).all()
    with pytest.raises(KeyError): DepthNet(dec_name='nope')
    with pytest.raises(NotImplementedError): DepthNet(use_virtual_stereo=True)
    with pytest.raises(KeyError): MonodepthDecoder([64], [2], out_act='tanh')


# ------------------------------------------------------------------------------------------------- training step (oracle backend)
def _cfg(learn_K=False):
    return {'net': {'depth': {'enc_name': 'resnet18', 'pretrained': False, 'out_scales': [0, 1, 2, 3]},
                    'pose': {'enc_name': 'resnet18', 'learn_K': learn_K}},
            'loss': {'img_recon': {'weight': 1, 'use_min': True, 'use_automask': True}, 'disp_smooth': {'weight': 0.001, 'use_edges': True}},
            'optimizer': {'type': 'adamw', 'lr': 1e-4, 'weight_decay': 1e-3},
            'scheduler': {'steplr': {'step_size': 40, 'gamma': 0.1}, 'linear': {'start_factor': 0.1, 'total_iters': 4}},
            'trainer': {'min_depth': 0.1, 'max_depth': 100, 'log_images': True}}


@pytest.mark.parametrize('learn_K', [False, True])
def test_training_step_on_cpu_with_oracle_backend(learn_K):
    torch.manual_seed(0)
    cfg = _cfg(learn_K)
    m = MonoDepthModule(copy.deepcopy(cfg), loss_backend=OracleBackend())
    batch = make_batch(2, 64, 96, (-1, 1), seed=1)
    loss, ld, fwd = m.step(batch)
    assert torch.isfinite(loss) and {'loss_img_recon', 'loss_disp_smooth', 'automask', 'supp_imgs_warp', 'disp_grad', 'image_grad'} <= set(ld)
    assert ld['supp_imgs_warp'].shape == (2, 2, 3, 64, 96) and ld['automask'].dtype == torch.bool
    assert fwd['Ts'].shape == (2, 2, 4, 4) and set(fwd['depth_up']) == {0, 1, 2, 3}
    assert ('K' in fwd) == learn_K
    torch.testing.assert_close(loss, ld['loss_img_recon'] + 0.001*ld['loss_disp_smooth'])
    conf = m.configure_optimizers()
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.nets['depth'].parameters())
    pose_grads = [p.grad for n, p in m.nets['pose'].named_parameters() if ('focal' in n or 'offset' in n)]
    if learn_K: assert pose_grads and all(g is not None for g in pose_grads)      # intrinsics heads receive gradient through K
    conf['optimizer'].step(); conf['lr_scheduler'].step()
    with pytest.raises(ValueError, match='Missing loss key'):
        bad = MonoDepthModule(copy.deepcopy(cfg), loss_backend=OracleBackend())
        bad.losses['nope'] = SmoothReg(); bad.weights['nope'] = torch.nn.Parameter(torch.tensor(1.), requires_grad=False)
        bad.step(batch)


class _RecordingBackend(OracleBackend):
    """The oracle backend plus a `loss_path` that only records what the trainer hands the single-node operator (and declines, so the handlers run)."""
    def __init__(self): super().__init__(); self.calls = []
    def intrinsics(self, fs, cs, size):
        K, _ = super().intrinsics(fs, cs, size)
        return K, torch.linalg.inv(K)                             # (the HIP backend returns the pair; the single-node path is only offered leaves when K_inv exists)
    def loss_path(self, crit, reg, depths, disps, imgs, supp_imgs, Ts, Ks, K_inv, w_recon, w_smooth, pose=None, intrinsics=None, prepared=None):
        self.calls.append({'w': (w_recon, w_smooth), 'pose': pose, 'intrinsics': intrinsics, 'K_inv': K_inv})
        return None


def test_single_node_loss_path_gets_intrinsics_leaves_only_with_pose_leaves():
    """a learned-K pose network with a STEREO support (index 0: its pose comes with the batch) has no pose leaves for the loss path; handing it
    `intrinsics=(fs, cs)` anyway let the forward succeed and the backward die (`the intrinsics' chain rule needs the pose chain`).  The trainer must then pass
    K / K_inv as tensors (autograd carries their gradients) — and the operator itself must refuse the combination before launching anything."""
    torch.manual_seed(0)
    cfg = _cfg(True)
    for idxs, want_leaves in (((-1, 1), True), ((-1, 0), False)):
        be = _RecordingBackend()
        m = MonoDepthModule(copy.deepcopy(cfg), loss_backend=be)
        m.want_aux = False                                       # (the single-node path is the no-image-logging path)
        x, y, meta = make_batch(2, 64, 96, idxs, seed=1)
        if 0 in idxs: y['T_stereo'] = torch.eye(4).expand(2, 4, 4).clone()
        be.inv_intrinsics = lambda K: torch.linalg.inv(K)
        loss, ld, fwd = m.step((x, y, meta))
        assert len(be.calls) == 1
        c = be.calls[0]
        assert (c['pose'] is not None) == want_leaves and (c['intrinsics'] is not None) == want_leaves, (idxs, c['pose'] is not None, c['intrinsics'] is not None)
        assert '_pose_leaves' in fwd and not hasattr(m, '_pose_leaves')          # they travel with `fwd`, nothing graph-attached stays on the module
        loss.backward()                                                          # the handlers' path that ran instead: gradients reach the intrinsics heads
        assert all(p.grad is not None for n, p in m.nets['pose'].named_parameters() if ('focal' in n or 'offset' in n))


def test_single_node_loss_path_follows_the_state_dict_weights():
    """`weights` is a ParameterDict in the state dict; a checkpoint / --resume overwrites it in place and the handlers' path (and the reference)
    then weigh the losses with the CHECKPOINT's values.  The single-node path takes the weights as host scalars: they must follow."""
    torch.manual_seed(0)
    be = _RecordingBackend()
    m = MonoDepthModule(copy.deepcopy(_cfg(False)), loss_backend=be)
    m.want_aux = False
    batch = make_batch(2, 64, 96, (-1, 1), seed=1)
    m.step(batch)
    assert be.calls[-1]['w'] == (1.0, pytest.approx(0.001))
    sd = m.state_dict()
    sd['weights.disp_smooth'] = torch.tensor(0.25); sd['weights.img_recon'] = torch.tensor(2.0)
    m.load_state_dict(sd)
    loss, ld, _ = m.step(batch)
    assert be.calls[-1]['w'] == (2.0, 0.25)
    torch.testing.assert_close(loss, 2.0*ld['loss_img_recon'] + 0.25*ld['loss_disp_smooth'])   # (the handlers' path that ran: same weights)


def test_loss_phases_reproduce_the_reference_numbers(golden):
    """forward_postprocess + forward_loss of the module (oracle backend) on the reference's own fixture."""
    g = golden('train_kbr_24x32')
    leaves, static = case_inputs(g, requires_grad=False)
    m = MonoDepthModule(_cfg(), loss_backend=OracleBackend(aten=True))
    n, b = leaves['aa'].shape[:2]
    Ts = geometry.T_from_AAt(leaves['aa'].flatten(0, 1), leaves['t'].flatten(0, 1)).unflatten(0, (n, b))
    fwd = {'disp': {s: leaves[f'disp_{s}'] for s in static['scales']}}
    for i, T in zip(static['supp_idxs'], Ts): fwd[f'T_{i}'] = torch.linalg.inv(T) if i < 0 else T   # always_fwd_pose
    x = {'imgs': static['imgs'], 'supp_idxs': torch.tensor(static['supp_idxs'])}
    y = {'imgs': static['imgs'], 'supp_imgs': static['supp_imgs'], 'K': static['K']}
    torch.manual_seed(42 + 7)   # the seed under which the fixture drew its tie-break noise
    fwd = m.forward_postprocess(fwd, x, y)
    loss, ld = m.forward_loss(fwd, x, y)
    torch.testing.assert_close(fwd['Ts'], g['out_Ts'], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(loss, g['out_loss'], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(ld['loss_disp_smooth'], g['out_loss_disp_smooth'], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(ld['supp_imgs_warp'], g['out_supp_imgs_warp'], rtol=0, atol=2e-5)


def test_scale_dict_is_a_dict_of_views():
    st = torch.rand(3, 2, 1, 4, 5)
    sd = ScaleDict.from_stack([0, 1, 3], st)
    assert list(sd) == [0, 1, 3] and sd.stacked is st and sd[3].data_ptr() == st[2].data_ptr()


def test_miopen_db_is_installed_to_a_private_versioned_copy(tmp_path, monkeypatch):
    """The shipped find-db must never be written to in place: MIOPEN_USER_DB_PATH points at a per-user copy keyed by content."""
    import importlib
    from slowtv_monodepth_amd import miopen_tuning
    monkeypatch.delenv('MIOPEN_USER_DB_PATH', raising=False)
    monkeypatch.delenv('SMD_NO_MIOPEN_DB', raising=False)
    monkeypatch.setattr(miopen_tuning.tempfile, 'gettempdir', lambda: str(tmp_path))
    dst = miopen_tuning.install()
    assert dst is not None and dst.startswith(str(tmp_path)) and os.environ['MIOPEN_USER_DB_PATH'] == dst
    shipped = sorted(p.name for p in (ROOT/'slowtv_monodepth_amd'/'miopen_db').glob('*.txt'))
    assert shipped and sorted(os.listdir(dst)) == shipped
    monkeypatch.setenv('MIOPEN_USER_DB_PATH', '/somewhere/else')
    assert miopen_tuning.install() is None and os.environ['MIOPEN_USER_DB_PATH'] == '/somewhere/else'   # the user's choice wins
    importlib.reload(miopen_tuning)


def test_checkpoint_key_bridge_round_trips_and_names_match_timm():
    """networks/checkpoint.py: our names -> reference/timm names -> our names is the identity for every supported trunk, the
    timm-side names are the published ones, and a reference-layout checkpoint of the whole trainer loads back bit-exactly."""
    import torch
    from slowtv_monodepth_amd.networks import DepthNet, PoseNet, checkpoint as ck
    for enc in ('resnet18', 'resnet50', 'convnext_tiny'):
        net = DepthNet(enc_name=enc, pretrained=False)
        ref = ck.to_reference_state_dict(net)
        assert {ck.from_reference_key(k, net.out_scales) for k in ref} == set(net.state_dict())
    r18 = ck.to_reference_state_dict(DepthNet(enc_name='resnet18', pretrained=False))
    for k in ('encoder.conv1.weight', 'encoder.bn1.running_var', 'encoder.layer1.0.conv1.weight', 'encoder.layer2.0.downsample.0.weight',
              'encoder.layer4.1.bn2.bias', 'decoders.disp.decoder.0.conv.weight', 'decoders.disp.decoder.9.conv.bias', 'decoders.disp.decoder.13.weight'):
        assert k in r18, k
    cnx = ck.to_reference_state_dict(DepthNet(enc_name='convnext_tiny', pretrained=False))
    for k in ('encoder.stem_0.weight', 'encoder.stem_1.bias', 'encoder.stages_0.blocks.2.conv_dw.weight', 'encoder.stages_1.downsample.1.weight',
              'encoder.stages_3.blocks.0.mlp.fc2.bias', 'encoder.stages_2.blocks.8.gamma'):
        assert k in cnx, k
    trainer = torch.nn.Module()
    trainer.nets = torch.nn.ModuleDict({'depth': DepthNet(enc_name='resnet18', pretrained=False), 'pose': PoseNet(enc_name='resnet18', learn_K=True)})
    ckpt = ck.reference_checkpoint(trainer, epoch=3, global_step=99)
    assert all(k.startswith('nets.') for k in ckpt['state_dict']) and 'nets.depth.encoder.layer1.0.conv1.weight' in ckpt['state_dict']
    other = torch.nn.Module()
    other.nets = torch.nn.ModuleDict({'depth': DepthNet(enc_name='resnet18', pretrained=False), 'pose': PoseNet(enc_name='resnet18', learn_K=True)})
    ck.load_reference_checkpoint(other, ckpt)
    for (ka, va), (kb, vb) in zip(trainer.state_dict().items(), other.state_dict().items()): assert ka == kb and torch.equal(va, vb)


def test_pretrained_request_is_refused_loudly():
    import pytest, warnings
    from slowtv_monodepth_amd.networks.encoders import create_encoder
    with pytest.warns(UserWarning, match='pretrained'): create_encoder('resnet18', pretrained=True)


# ------------------------------------------------------------------------------------------------- train.py guards
def test_cfg_in_the_reference_dataset_layout_is_refused_without_synthetic_data(tmp_path):
    """The reference keys `dataset` BY TYPE (cfg/default.yaml:86-111, parsers.get_ds iterates `for t, kw in cfg.items()`): such a
    cfg names real datasets and must not silently train on synthetic triplets."""
    from slowtv_monodepth_amd import train as T
    cfg = {'dataset': {'kitti_lmdb': {'split': 'eigen_benchmark', 'supp_idxs': [-1, 1], 'train': {'mode': 'train', 'shape': [376, 1242]}},
                       'mannequin_lmdb': {'datum': 'image support K', 'supp_idxs': [-2, 1]}, 'slow_tv_lmdb': None}}
    assert T.dataset_types(cfg) == ['kitti_lmdb', 'mannequin_lmdb']           # `key: null` drops an inherited dataset
    assert T.dataset_types({'dataset': {'main': {'type': 'kitti_lmdb'}}}) == ['kitti_lmdb'] and T.dataset_types({}) == []
    assert T.dataset_supp_idxs(cfg) == [-1, 1] and T.dataset_supp_idxs({}) == [-1, 1]
    import yaml
    full = {'net': {'depth': {'enc_name': 'resnet18', 'pretrained': False, 'dec_name': 'monodepth', 'out_scales': [0, 1, 2, 3]},
                    'pose': {'enc_name': 'resnet18', 'pretrained': False}},
            'loss': {'img_recon': {'weight': 1, 'loss_name': 'ssim', 'use_min': True, 'use_automask': True}},
            'optimizer': {'type': 'adamw', 'lr': 1e-4}, 'loader': {'batch_size': 1}, **cfg}
    f = tmp_path/'cfg.yaml'; f.write_text(yaml.safe_dump(full))
    with pytest.raises(SystemExit) as e: T.main(['-c', str(f), '-n', 'x', '-o', str(tmp_path), '--steps', '1', '--shape', '32', '64'])
    assert 'kitti_lmdb' in str(e.value) and '--synthetic-data' in str(e.value)


def test_losses_without_a_producing_network_are_refused_at_construction():
    cfg = {'net': {'depth': {'enc_name': 'resnet18', 'pretrained': False, 'dec_name': 'monodepth', 'out_scales': [0]},
                   'pose': {'enc_name': 'resnet18', 'pretrained': False}},
           'loss': {'img_recon': {'weight': 1}, 'stereo_const': {'weight': 1}}}
    with pytest.raises(NotImplementedError, match='stereo_const'): MonoDepthModule(cfg, loss_backend=OracleBackend())
    cfg['loss'] = {'img_recon': {'weight': 1}, 'feat_recon': {'weight': 1, 'loss_name': 'l2'}}
    with pytest.raises(NotImplementedError, match='autoencoder'): MonoDepthModule(cfg, loss_backend=OracleBackend())


def test_checkpoint_has_the_fields_lightning_restores_and_resume_round_trips(tmp_path):
    from slowtv_monodepth_amd.networks.checkpoint import load_reference_checkpoint, reference_checkpoint
    cfg = {'net': {'depth': {'enc_name': 'resnet18', 'pretrained': False, 'dec_name': 'monodepth', 'out_scales': [0, 1]},
                   'pose': {'enc_name': 'resnet18', 'pretrained': False}},
           'loss': {'img_recon': {'weight': 1}}, 'optimizer': {'type': 'adamw', 'lr': 1e-3},
           'scheduler': {'steplr': {'step_size': 1, 'gamma': 0.5}}}
    torch.manual_seed(0)
    m = MonoDepthModule(cfg, loss_backend=OracleBackend())
    conf = m.configure_optimizers(); opt, sched = conf['optimizer'], conf['lr_scheduler']
    sched.step()
    ck = reference_checkpoint(m, epoch=3, global_step=77, optimizer=opt, scheduler=sched)
    for k in ('state_dict', 'epoch', 'global_step', 'optimizer_states', 'lr_schedulers', 'loops', 'callbacks', 'pytorch-lightning_version'): assert k in ck
    assert ck['lr_schedulers'][0] == sched.state_dict() and ck['epoch'] == 3
    torch.save(ck, tmp_path/'last.ckpt')
    torch.manual_seed(1)
    m2 = MonoDepthModule(cfg, loss_backend=OracleBackend())
    load_reference_checkpoint(m2, str(tmp_path/'last.ckpt'))
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()): assert k1 == k2 and torch.equal(v1, v2)


def test_resume_restores_optimizer_and_scheduler_together_or_fast_forwards_the_schedule():
    """`train.restore_training_state`: a checkpoint whose optimizer state does not fit (a reference checkpoint: timm's parameter groups)
    must not leave the epoch counter at N with a scheduler at epoch 0 — the fresh scheduler is stepped to N; a fitting checkpoint restores both."""
    import warnings
    from slowtv_monodepth_amd.train import restore_training_state
    def fresh():
        p = [torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2))]
        opt = torch.optim.AdamW([{'params': [p[0]]}, {'params': [p[1]]}], lr=1e-3)
        return opt, torch.optim.lr_scheduler.StepLR(opt, step_size=2, gamma=0.1)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        opt, sched = fresh()
        for _ in range(5): sched.step()
        good = {'epoch': 4, 'optimizer_states': [opt.state_dict()], 'lr_schedulers': [sched.state_dict()]}
        o2, s2 = fresh()
        assert restore_training_state(good, o2, s2, verbose=False) == 5 and s2.last_epoch == 5 and abs(o2.param_groups[0]['lr'] - 1e-5) < 1e-12
        other = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=1.0)      # one parameter group instead of two: load_state_dict raises ValueError
        bad = {'epoch': 4, 'optimizer_states': [other.state_dict()], 'lr_schedulers': [sched.state_dict()]}
        o3, s3 = fresh()
        assert restore_training_state(bad, o3, s3, verbose=False) == 5
        assert s3.last_epoch == 5 and abs(o3.param_groups[0]['lr'] - 1e-5) < 1e-12 and len(o3.state) == 0     # schedule at epoch 5, moments fresh
        o4, s4 = fresh()
        assert restore_training_state({'epoch': 1}, o4, s4, verbose=False) == 2 and abs(o4.param_groups[1]['lr'] - 1e-4) < 1e-12


def test_trainer_applies_the_aspect_ratio_augmentation_in_training_mode_only():
    """src/core/trainer.py:54-60, 106: `training_step` augments the batch, `validation_step` does not (host logic, CPU operator)."""
    import random
    cfg = {'net': {'depth': {'enc_name': 'resnet18', 'pretrained': False, 'dec_name': 'monodepth', 'out_scales': [0, 1]},
                   'pose': {'enc_name': 'resnet18', 'pretrained': False}},
           'loss': {'img_recon': {'weight': 1, 'use_min': True, 'use_automask': True}},
           'trainer': {'min_depth': 0.1, 'max_depth': 100, 'aspect_ratio_aug_prob': 1.0, 'aspect_ratio_ref_shape': [128, 192]}}
    torch.manual_seed(0)
    m = MonoDepthModule(cfg, loss_backend=OracleBackend())
    random.seed(5); torch.manual_seed(5)                  # samples a 16/9 crop (81, 145) -> resized to (96, 192)
    batch = make_batch(1, 128, 192, (-1, 1), seed=1)
    loss, _, fwd = m.step(batch, mode='train')
    assert torch.isfinite(loss) and len(batch[2]['augs']) == 2
    h, w = batch[0]['imgs'].shape[-2:]
    assert (h, w) == (96, 192) and h*w <= 0.8*128*192
    assert batch[1]['supp_imgs'].shape[-2:] == (h, w) and fwd['depth_up'][0].shape[-2:] == (h, w)
    batch2 = make_batch(1, 128, 192, (-1, 1), seed=1)
    m.step(batch2, mode='val')
    assert tuple(batch2[0]['imgs'].shape[-2:]) == (128, 192) and 'augs' not in batch2[2]


def test_dead_tile_shares_count_what_the_backward_could_skip():
    """`functional.dead_tile_shares`: per support and scale, the share of (row, 60-column tile) units where no pixel routes gradient to
    the support (a reporting statistic of `profiles/r03_skip_regimes.txt`).  Pure torch, so it runs on the host."""
    import torch
    from slowtv_monodepth_amd import functional as F
    S, b, h, w = 2, 3, 8, 130                       # 130 columns = tiles of 60, 60 and 10
    sel = torch.full((S, b, 1, h, w), 255, dtype=torch.uint8)
    assert torch.equal(F.dead_tile_shares(sel, True, 2), torch.ones(2, S)) and torch.equal(F.dead_tile_shares(sel, False, 2), torch.ones(2, S))
    sel[0] = 1                                      # scale 0: support 1 wins everywhere (its waves are busy in every row, support 0's never)
    assert F.dead_tile_shares(sel, True, 2).tolist() == [[1.0, 1.0], [0.0, 1.0]] and F.dead_tile_shares(sel, False, 2).tolist() == [[0.0, 1.0], [0.0, 1.0]]
    sel[0] = 255; sel[0, :, :, :4, 125] = 0; sel[0, :, :, :, 3] = 1     # support 0: one pixel of the last (10-wide) tile in half of the rows; support 1: the first tile of every row
    sh = F.dead_tile_shares(sel, True, 2)
    assert abs(float(sh[0, 0]) - (1 - 0.5/3)) < 1e-6 and abs(float(sh[1, 0]) - (1 - 1/3)) < 1e-6 and sh[:, 1].tolist() == [1.0, 1.0]


def test_row_skip_tuner_schedule(monkeypatch):
    """`functional._RowSkipTuner`: the first 2*trials backward calls of a period alternate between the two row loops, afterwards the
    chosen one is used; `SMD_BWD_SKIP` in the environment switches the tuner off.  (Events are only created on a GPU: the schedule
    itself is checked here with the event calls stubbed out.)"""
    import torch
    from slowtv_monodepth_amd import functional as F
    from slowtv_monodepth_amd._lib import FLAGS

    class FakeEvent:
        def __init__(self, enable_timing=False): pass
        def record(self, stream=None): pass
        def query(self): return True
        def elapsed_time(self, other): return FakeEvent.times.pop(0)
    monkeypatch.setattr(torch.cuda, 'Event', FakeEvent); monkeypatch.setattr(torch.cuda, 'current_stream', lambda dev=None: None)
    monkeypatch.delenv('SMD_BWD_SKIP', raising=False)
    t = F._RowSkipTuner(); t.period = t.period_min = t.period_max = 10; t.settle, t.trials = 1, 2
    FakeEvent.times = [0.100, 0.120, 0.101, 0.119,    # period 1: skipping 0.100 / 0.101, plain 0.120 / 0.119 -> skipping
                       0.130, 0.120, 0.131, 0.121]    # period 2: the other way round -> plain
    seen = []
    for _ in range(20):
        flag, token = t.begin('cuda:0'); t.end(token)
        seen.append(flag != 0)
    assert seen[:5] == [False, True, False, True, False] and all(seen[5:10]) and t.last is not None
    assert seen[10:15] == [True, True, False, True, False] and not any(seen[15:20])
    assert t.last == {'skipping_ms': 0.13, 'plain_ms': 0.12, 'next_period': 10}
    # the period adapts: it doubles while the timings confirm the choice and falls back to the minimum when the choice flips
    a = F._RowSkipTuner(); a.period = a.period_min = 8; a.period_max = 32; a.settle, a.trials = 0, 1
    FakeEvent.times = [0.1, 0.2]*3 + [0.2, 0.1] + [0.2, 0.1]    # skipping x3, then plain x2
    periods = []
    for _ in range(8 + 8 + 16 + 32 + 8 + 4):
        flag, token = a.begin('cuda:0'); a.end(token)
        periods.append(a.period)
    assert periods[2] == 8 and periods[8 + 2] == 16 and periods[8 + 16 + 2] == 32 and periods[8 + 16 + 32 + 2] == 8 and periods[-1] == 16, periods[::4]
    monkeypatch.setenv('SMD_BWD_SKIP', '2')      # pinned from the environment: no timing, the gated loop's flag on every call
    assert t.begin('cuda:0') == (FLAGS['bwd_skip_rows'], None)
    monkeypatch.setenv('SMD_BWD_SKIP', '0')
    assert t.begin('cuda:0') == (0, None)
