"""§8f rank 4 pinned on the reference: `net_decoder_64x96.npz` holds what the REFERENCE's `MonodepthDecoder` (src/networks/decoders/monodepth.py) produced —
its four sigmoid disparities, the gradients w.r.t. every encoder feature and every parameter — on named, seeded weights / features / output gradients
(tests/golden/exact_inputs.py; written by `make_golden.py --decoder-only`, which imports the reference in the build container).

CPU: this package's decoder evaluated op by op (the ATen composition, as the reference) reproduces it.
GPU: so does the path the trainer runs — glue kernels between the convolutions (`smd_elu_pad_*`, `smd_elu_up_cat_pad_*`), the one-channel heads as stencils
(`smd_conv3x3_head_*`), the thin last stage on fp32 MFMA (`smd_conv3x3_thin_*`), the wide stages on the bf16 matrix cores with three-way split operands
(`smd_conv3x3_mfma_*`) or MIOpen."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, rel_to_max
from exact_inputs import DECODER_KW, bit_checksum, decoder_feats, decoder_out_grads, decoder_state


def build(device):
    from slowtv_monodepth_amd.networks import checkpoint as ck
    from slowtv_monodepth_amd.networks.decoders import MonodepthDecoder
    dec = MonodepthDecoder(**DECODER_KW)
    holder = torch.nn.Module(); holder.decoders = torch.nn.ModuleDict({'disp': dec})
    shapes = {k: tuple(v.shape) for k, v in ck.to_reference_state_dict(holder).items()}
    state = decoder_state(shapes)
    ck.load_reference_state_dict(holder, state, strict=True)
    holder.to(device)
    return dec, holder, shapes, state


def run_and_compare(device, out_tol, grad_tol):
    from slowtv_monodepth_amd.networks import checkpoint as ck
    g = load_golden('net_decoder_64x96')
    with np.load(GOLDEN/'net_decoder_64x96.npz') as z: keys = [str(k) for k in z['meta_keys']]
    dec, holder, shapes, state = build(device)
    assert sorted(shapes) == keys, 'the key bridge no longer yields the reference decoder\'s parameter names'
    feats, gouts = decoder_feats(), decoder_out_grads()
    assert sum(bit_checksum(v) for v in state.values()) == int(g['chk_state']) and sum(bit_checksum(f) for f in feats) == int(g['chk_feats']) \
        and sum(bit_checksum(v) for v in gouts.values()) == int(g['chk_gouts']), 'the seeded inputs are not the ones the fixture was made from'
    feats = [f.to(device).requires_grad_(True) for f in feats]
    out = dec(feats)
    sum((out[i]*gouts[i].to(device)).sum() for i in out).backward()
    for i in DECODER_KW['out_sc']:
        d = (out[i].detach().cpu() - g[f'out_{i}']).abs().max().item()
        assert d <= out_tol, f'disparity at scale {i}: {d:.2e}'
    for j, f in enumerate(feats):
        r = rel_to_max(f.grad.cpu(), g[f'gfeat_{j}'])
        assert r <= grad_tol, f'gradient w.r.t. encoder feature {j}: {r:.2e}'
    # parameters by REFERENCE name: the small ones element by element, all of them through their sum and sum of magnitudes
    grads = {k: v for k, v in zip(ck.to_reference_state_dict(holder).keys(), (p.grad for p in holder.state_dict(keep_vars=True).values()))}
    stats = g['gparam_stats']
    for n, k in enumerate(keys):
        gk = grads[k].detach().double().cpu()
        if f'gparam_{k}' in g:
            r = rel_to_max(gk, g[f'gparam_{k}'].double())
            assert r <= grad_tol, f'gradient of {k}: {r:.2e}'
        assert abs(gk.abs().sum().item() - stats[n, 1].item()) <= 10*grad_tol*stats[n, 1].item(), f'sum of |gradient| of {k}'
        assert abs(gk.sum().item() - stats[n, 0].item()) <= 10*grad_tol*stats[n, 1].item(), f'sum of the gradient of {k}'
    return out


def test_decoder_matches_the_reference_decoder_on_the_cpu():
    run_and_compare('cpu', 1e-6, 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('route', ['mfma', 'auto', 'miopen'])
def test_decoder_kernels_match_the_reference_decoder(route):
    """`route` = who serves the wide convolutions: 'mfma' pins every one of them (forward, data and weight gradients) on the split-bf16 MFMA kernels
    (`smd_conv3x3_mfma_*`, round 6) — the reference's outputs and parameter gradients are then the yardstick for those kernels too; 'auto' is what the
    trainer runs (this box's A/B per operator and shape); 'miopen' the round-5 composition."""
    if not torch.cuda.is_available(): pytest.skip('needs a GPU')
    from slowtv_monodepth_amd import _lib, functional as HF
    _lib.lib.smd_last_error()
    HF.set_conv_route(route)
    try:
        out = run_and_compare('cuda', 2e-5, 2e-4)
    finally:
        HF.set_conv_route('auto')
    assert all(o.is_cuda for o in out.values())


@pytest.mark.gpu
@pytest.mark.parametrize('route', ['mfma', 'auto'])
def test_decoder_under_bf16_autocast_stays_close_to_the_fp32_reference(route):
    """BASELINE cfg 5 (the reference's `cfg/kbr/default.yaml`) trains the networks in bf16-mixed precision: under `torch.autocast(bfloat16)` the decoder's glue
    kernels write bf16, its thin stage and (route 'mfma': every) wide stage run the one-piece bf16 form of `smd_conv3x3_mfma_*`, its heads the bf16-input
    stencils.  Yardstick: the REFERENCE decoder's fp32 outputs and gradients (the fixture) at bf16's resolution — disparities to 1e-2,
    every feature / weight gradient by its sum of magnitudes to 5e-2 (ten bf16 roundings deep) — i.e. the autocast path computes the same network."""
    if not torch.cuda.is_available(): pytest.skip('needs a GPU')
    from slowtv_monodepth_amd import functional as HF
    from slowtv_monodepth_amd.networks import checkpoint as ck
    g = load_golden('net_decoder_64x96')
    dec, holder, shapes, state = build('cuda')
    feats, gouts = decoder_feats(), decoder_out_grads()
    feats = [f.cuda().requires_grad_(True) for f in feats]
    HF.set_conv_route(route)
    try:
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out = dec(feats)
        sum((out[i].float()*gouts[i].cuda()).sum() for i in out).backward()
    finally:
        HF.set_conv_route('auto')
    for i in DECODER_KW['out_sc']:
        d = (out[i].detach().float().cpu() - g[f'out_{i}']).abs().max().item()
        assert d <= 1e-2, f'disparity at scale {i}: {d:.2e}'
    for j, f in enumerate(feats):
        ref = g[f'gfeat_{j}']
        e = (f.grad.float().cpu() - ref).abs().sum().item()/ref.abs().sum().item()
        assert e <= 5e-2, f'gradient w.r.t. encoder feature {j}: {e:.2e} of its sum of magnitudes'
    grads = {k: v for k, v in zip(ck.to_reference_state_dict(holder).keys(), (p.grad for p in holder.state_dict(keep_vars=True).values()))}
    with np.load(GOLDEN/'net_decoder_64x96.npz') as z: keys = [str(k) for k in z['meta_keys']]
    stats = g['gparam_stats']
    for n, k in enumerate(keys):      # the weights (a bias gradient is ONE sum per channel that may cancel: no yardstick for it at bf16's resolution in the fixture)
        if not k.endswith('.weight'): continue
        gk = grads[k].detach().double().cpu()
        assert abs(gk.abs().sum().item() - stats[n, 1].item()) <= 5e-2*stats[n, 1].item(), f'sum of |gradient| of {k}'
