"""Seeded random shapes and options through the whole loss path the trainer runs — K0-fused reconstruction + smoothness, forward and
backward — against the CPU oracle.  The hand-picked sweeps of test_gpu_parity.py sit on the boundaries someone thought of (strip and
wave widths, epochs of the target ring, streaming K0 ratios); this one draws batch, image size, number of supports, pyramid sizes
(exact halvings or unrelated ones), depth range, loss and mask options at random so that an unlucky combination is met before a user
meets it.  Same tolerances as test_gpu_parity.py; a pixel whose selection differs (a tie decided by rounding) moves its gradient
between supports, so cases with flips get the coarse gradient bound.

What the first run of this file found (round 4): the oracle evaluated the bilinear source index with two roundings where ATen (and the
kernels) fuse it into one — lambda off by up to 1e-5 beyond column 128 at pyramid ratios that are not exact halvings (fixed in
oracle/view_synth_oracle.py::_src_index, checked against F.interpolate); and one L1 sign knife edge (below).

Reading a wider hunt (SMD_FUZZ_SEEDS=300-400): 4 % of the one-to-four-support cases and 9 % of the five-to-eight-support cases are flagged, always
through one to four gradient elements or through the pose gradient of ONE support.  Dissected with tests/fuzz_case.py / fuzz_case_many.py against
the fp64 oracle under the same routing, every one is a discontinuity of the loss itself that fp32 rounding decides: the sign of an L1 term with
|pred - target| ~ 1e-6, a min-reprojection tie, or a sampling coordinate within rounding of an integer (the bilinear value is continuous there,
its derivative is not: the two sides read different texel pairs).  The fp32 oracle shows the same events, of the same size, under a 1e-6
perturbation of its own inputs, and in four of the ten dissected cases it is the kernel that agrees with fp64 and the fp32 oracle that does not.
(Seeds 23 / 141 / 118 of the many-supports family: the worst pixel samples the flagged support at 12.000000, 20.999998, 27.999995.)"""
import os
import random

import pytest
import torch

from conftest import rel_to_max
from oracle import view_synth_oracle as O

pytestmark = pytest.mark.gpu

_N = int(os.environ.get('SMD_FUZZ_SEEDS', 0))                     # a wider net for a one-off hunt: SMD_FUZZ_SEEDS=400
SEEDS = list(range(_N or 24))
_more = lambda base: list(range(max(base, _N//2)))


def draw(seed):
    r = random.Random(9000 + seed)
    b = r.choice([1, 1, 2, 3])
    h = r.choice([r.randint(2, 12), r.randint(13, 40), r.randint(41, 90)])
    w = r.choice([r.randint(2, 20), r.randint(55, 70), r.randint(110, 135), r.randint(21, 200)])
    n = r.choice([1, 2, 2, 3, 4])
    S = r.choice([1, 2, 3, 4, 4])
    if r.random() < 0.5: lows = [(max(h >> s, 1), max(w >> s, 1)) for s in range(S)]
    else: lows = [(r.randint(1, h), r.randint(1, w)) for _ in range(S)]
    opts = dict(loss_name=r.choice(['ssim', 'ssim', 'ssim', 'l1']), use_min=r.random() < 0.7, use_automask=r.random() < 0.7, use_edges=r.random() < 0.7)
    lo = r.choice([0.1, 0.01, 0.5]); hi = r.choice([100.0, 10.0, 80.0])
    return b, h, w, n, lows, opts, lo, hi


@pytest.mark.parametrize('seed', SEEDS)
def test_random_shapes_and_options_match_the_oracle(seed):
    if not torch.cuda.is_available(): pytest.skip('needs a GPU')
    from slowtv_monodepth_amd import functional as F
    b, h, w, n, lows, opts, lo, hi = draw(seed)
    S = len(lows)
    gen = torch.Generator().manual_seed(seed)
    imgs = torch.rand(b, 3, h, w, generator=gen)
    # supports = a blend of the target and an unrelated image: the identity error of the automask and the warped errors are of the same
    # order, so both the automask and the supports win somewhere (a support that is the target + noise is masked everywhere)
    mix = 0.6*torch.rand(1, generator=gen).item()
    supp = mix*imgs[None] + (1 - mix)*torch.rand(n, b, 3, h, w, generator=gen)
    disps = {s: 0.05 + 0.9*torch.rand(b, 1, hs, ws, generator=gen) for s, (hs, ws) in enumerate(lows)}
    aa = 0.02*torch.randn(n*b, 3, generator=gen); t = 0.1*torch.randn(n*b, 3, generator=gen)
    K = torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]])[None].repeat(b, 1, 1)
    noise = torch.randn(S*b, 1, h, w, generator=gen) if opts['use_automask'] else None
    what = f'seed {seed}: b={b} {h}x{w} n={n} pyramid {lows} {opts} depth [{lo}, {hi}]'

    # oracle (fp32, CPU)
    dc = {s: d.clone().requires_grad_(True) for s, d in disps.items()}
    Tc = O.T_from_AAt(aa, t).unflatten(0, (n, b)).clone().requires_grad_(True)
    loss_c, out_c = O.loss_path(dc, imgs, supp, Tc, K, min_depth=lo, max_depth=hi, loss_name=opts['loss_name'], use_min=opts['use_min'],
                                use_automask=opts['use_automask'], use_edges=opts['use_edges'], w_smooth=0.1, noise=noise)
    loss_c.backward()

    # HIP
    dg = [d.cuda().requires_grad_(True) for d in disps.values()]
    Tg = Tc.detach().cuda().requires_grad_(True)
    l_rec, err, sel, _, dep = F.image_recon_fused_disp(dg, imgs.cuda(), supp.cuda(), Tg, K.cuda(), flags=F.recon_flags(opts['loss_name'], opts['use_min'], opts['use_automask']),
                                                       min_depth=lo, max_depth=hi, noise=None if noise is None else noise.cuda(), want_err=True)
    l_sm, _, _ = F.disp_smooth_fused(dict(enumerate(dg)), imgs.cuda(), use_edges=opts['use_edges'], want_aux=False)
    (l_rec + 0.1*l_sm).backward()

    for s in range(S): torch.testing.assert_close(dep[s].cpu(), out_c['depth_up'][s].detach(), rtol=2e-5, atol=1e-5, msg=lambda m: f'{what}: depth_up[{s}] {m}')
    torch.testing.assert_close(l_sm.detach().cpu(), out_c['loss_disp_smooth'].detach(), rtol=2e-5, atol=1e-7, msg=lambda m: f'{what}: smoothness {m}')
    full = out_c['full']
    flips = (sel.cpu() != full['sel']).flatten()
    share = flips.float().mean().item()
    assert share <= 0.01, f'{what}: selection differs on {share:.2%} of the pixels'
    torch.testing.assert_close(err.cpu().flatten()[~flips], full['err'].detach().flatten()[~flips], rtol=0, atol=3e-4, msg=lambda m: f'{what}: error map {m}')
    torch.testing.assert_close(l_rec.detach().cpu(), out_c['loss_img_recon'].detach(), rtol=1e-4 if not flips.any() else 2e-3, atol=1e-6, msg=lambda m: f'{what}: loss {m}')
    tol = 5e-2 if flips.any() else (1e-2 if opts['loss_name'] == 'l1' else 2e-3)
    for s in range(S):
        # per element, relative to the largest gradient.  A few elements may sit on a knife edge of the loss itself — |pred - target| of one
        # channel within rounding of zero flips the sign of its L1 term (seed 16: 1.3e-6 in fp64, one pixel of 9918; tests/fuzz_case.py shows
        # the fp32 oracle on one side and the kernel on the other) — so isolated outliers are allowed, bounded in number and in size
        e = (dg[s].grad.cpu() - dc[s].grad).abs()/dc[s].grad.abs().max().clamp(min=1e-20)
        n_out = int((e > tol).sum())
        assert n_out <= max(3, int(2e-4*e.numel())) and e.max().item() < 5e-2, \
            f'{what}: d loss / d disp_{s}: {n_out} of {e.numel()} elements off by more than {tol:.0e} of the max, worst {e.max().item():.3e}'
    e = rel_to_max(Tg.grad.cpu()[..., :3, :], Tc.grad[..., :3, :])
    assert e < tol, f'{what}: d loss / d T off by {e:.3e} (rel. to max)'


@pytest.mark.parametrize('seed', _more(32))
def test_random_shapes_single_node_loss_path_equals_the_separate_operators(seed):
    """Round 5: the trainer's hot call is the single-node loss path (`functional.loss_path_fused`: guest blocks in the drain of the forward launch and
    beside the K0 adjoint, the weighted sum formed in-launch, the pose chain rule in the epilogue's wave).  Random batch / image / pyramid sizes, one
    to four supports, inverted poses, learned intrinsics or not: bit-equal to `pose_matrices` + `image_recon_fused_disp` + `disp_smooth_fused` + the
    eager weighted sum wherever the operator serves the draw (it declines single-level pyramids, a level taller than the image and two
    full-resolution levels: then `Unsupported` must be raised and nothing else)."""
    if not torch.cuda.is_available(): pytest.skip('needs a GPU')
    from slowtv_monodepth_amd import functional as F
    from slowtv_monodepth_amd._lib import Unsupported
    r = random.Random(7000 + seed)
    b = r.choice([1, 2, 3, 5])
    h = r.choice([r.randint(2, 12), r.randint(13, 40), r.randint(41, 100)])
    w = r.choice([r.randint(2, 20), r.randint(55, 70), r.randint(110, 135), r.randint(21, 260)])
    n = r.choice([1, 2, 2, 3, 4])
    S = r.choice([1, 2, 3, 4, 4, 4])
    if r.random() < 0.6: lows = [(max(h >> s, 1), max(w >> s, 1)) for s in range(S)]
    else: lows = [(r.randint(1, h), r.randint(1, w)) for _ in range(S)]
    keys = list(range(S)) if r.random() < 0.7 else sorted(r.sample(range(6), S))
    use_min, automask, learn_k = r.random() < 0.7, r.random() < 0.7, r.random() < 0.4
    gen = torch.Generator(device='cuda').manual_seed(seed)
    imgs = torch.rand(b, 3, h, w, device='cuda', generator=gen)
    supp = 0.5*imgs[None] + 0.5*torch.rand(n, b, 3, h, w, device='cuda', generator=gen)
    d0 = [0.05 + 0.9*torch.rand(b, 1, hs, ws, device='cuda', generator=gen) for hs, ws in lows]
    aa0, t0 = 0.02*torch.randn(n*b, 3, device='cuda', generator=gen), 0.1*torch.randn(n*b, 3, device='cuda', generator=gen)
    inv = torch.tensor([r.random() < 0.5 for _ in range(n) for _ in range(b)], dtype=torch.uint8, device='cuda') if r.random() < 0.7 else None
    fs0 = torch.tensor([0.58, 1.92], device='cuda')[None].repeat(b, 1)*(1 + 0.05*torch.randn(b, 2, device='cuda', generator=gen)); cs0 = 0.5 + 0.03*torch.randn(b, 2, device='cuda', generator=gen)
    K0 = torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device='cuda')[None].repeat(b, 1, 1)
    flags = F.recon_flags('ssim', use_min, automask)
    w_sm = torch.tensor(r.choice([0.001, 0.1, 1.0]), device='cuda')
    what = f'seed {seed}: b={b} {h}x{w} n={n} pyramid {lows} keys {keys} min={use_min} automask={automask} learnK={learn_k}'
    os.environ['SMD_BWD_SKIP'] = '0'
    try:
        def run(one_node):
            L = dict(d=[v.clone().requires_grad_(True) for v in d0], aa=aa0.clone().requires_grad_(True), t=t0.clone().requires_grad_(True))
            if learn_k: L.update(fs=fs0.clone().requires_grad_(True), cs=cs0.clone().requires_grad_(True))
            Ts = F.pose_matrices(L['aa'], L['t'], inv).unflatten(0, (n, b))
            K, K_inv = F.intrinsics(L['fs'], L['cs'], (h, w)) if learn_k else (K0, None)
            if one_node:
                loss, l_rec, l_sm, sel, dep = F.loss_path_fused(dict(zip(keys, L['d'])), imgs, supp, Ts, K, K_inv, pose=(L['aa'], L['t'], inv),
                                                                intrinsics=(L['fs'], L['cs']) if learn_k else None, flags=flags, min_depth=0.1, max_depth=100, seed=seed,
                                                                w_recon=1.0, w_smooth=float(w_sm))
            else:
                l_rec, _, sel, _, dep = F.image_recon_fused_disp(L['d'], imgs, supp, Ts, K, K_inv, flags=flags, min_depth=0.1, max_depth=100, seed=seed, want_err=False)
                l_sm, _, _ = F.disp_smooth_fused(dict(zip(keys, L['d'])), imgs, use_edges=True, want_aux=False)
                loss = (0. + torch.tensor(1.0, device='cuda')*l_rec) + w_sm*l_sm
            loss.backward()
            return loss.detach(), l_rec.detach(), l_sm.detach(), sel, dep.detach(), L
        supported = S >= 2 and all(hs <= h for hs, _ in lows) and sum((hs, ws) == (h, w) for hs, ws in lows) <= 1
        if not supported:
            with pytest.raises(Unsupported): run(True)
            return
        a, bb = run(False), run(True)
        torch.cuda.synchronize()
    finally: del os.environ['SMD_BWD_SKIP']
    for k, name in enumerate(('loss', 'l_rec', 'l_sm', 'sel', 'depth_up')): assert torch.equal(a[k], bb[k]), f'{what}: {name} differs'
    for k, (x, y) in enumerate(zip(a[5]['d'], bb[5]['d'])): assert torch.equal(x.grad, y.grad), f'{what}: d loss / d disp[{k}] differs by {(x.grad - y.grad).abs().max().item():.3e}'
    for k in ('aa', 't') + (('fs', 'cs') if learn_k else ()): assert rel_to_max(bb[5][k].grad, a[5][k].grad) <= 2e-6, f'{what}: d loss / d {k}: {rel_to_max(bb[5][k].grad, a[5][k].grad):.2e}'


@pytest.mark.parametrize('seed', _more(16))
def test_random_shapes_backward_block_composition(seed, knobs):
    """Round 5: with four scales the fused backward's blocks are the SCALES of one strip (knob `bwd_scales_block`) wherever the partition leaves every
    block its strip, else strips of one scale; and a strip's supports go to one wave in turn or to a wave each (knob `bwd_wps`).  Random sizes, two to
    four supports, either row loop: whatever the composition, the same disparity gradients bit for bit (the pose gradient to 2e-6: its per-block fp32
    sums group other waves), and the launch says which composition it ran."""
    if not torch.cuda.is_available(): pytest.skip('needs a GPU')
    from slowtv_monodepth_amd import functional as F, _lib
    r = random.Random(8100 + seed)
    b = r.choice([1, 2, 3, 6])
    h = r.choice([r.randint(17, 40), r.randint(41, 130), 64, 96])
    w = r.choice([r.randint(55, 70), r.randint(110, 135), r.randint(200, 500), 240, 480])
    n = r.choice([2, 2, 3, 4])
    lows = [(max(h >> s, 1), max(w >> s, 1)) for s in range(4)]
    use_min, automask = r.random() < 0.8, r.random() < 0.7
    skip = r.choice(['0', '0', '2'])
    gen = torch.Generator(device='cuda').manual_seed(seed)
    imgs = torch.rand(b, 3, h, w, device='cuda', generator=gen)
    supp = 0.5*imgs[None] + 0.5*torch.rand(n, b, 3, h, w, device='cuda', generator=gen)
    d0 = [0.05 + 0.9*torch.rand(b, 1, hs, ws, device='cuda', generator=gen) for hs, ws in lows]
    T0 = torch.eye(4, device='cuda').repeat(n, b, 1, 1); T0[..., :3, 3] = 0.02*torch.randn(n, b, 3, device='cuda', generator=gen)
    K = torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device='cuda')[None].repeat(b, 1, 1)
    flags = F.recon_flags('ssim', use_min, automask)
    what = f'seed {seed}: b={b} {h}x{w} n={n} min={use_min} automask={automask} row loop {skip}'
    os.environ['SMD_BWD_SKIP'] = skip
    try:
        def run(wps, scales):
            knobs('bwd_wps', wps); knobs('bwd_scales_block', scales)
            d = [v.clone().requires_grad_(True) for v in d0]; T = T0.clone().requires_grad_(True)
            loss, *_ = F.image_recon_fused_disp(d, imgs, supp, T, K, flags=flags, min_depth=0.1, max_depth=100, seed=seed, want_err=False)
            loss.backward(); torch.cuda.synchronize()
            return [v.grad for v in d], T.grad, _lib.lib.smd_last_kernel_variant(1).decode()
        ref = run(min(n, 4), 0)
        assert 'scales' not in ref[2], ref[2]
        seen = []
        for wps in sorted({1, 2, min(n, 4)}):
            # one wave per strip adds the supports' shares in the order one wave per support does; two waves for three or four supports pair them
            # otherwise ((g0 + g2) + (g1 + g3)): equal to rounding there, and bit-equal between its own two block compositions
            own = ref if wps == min(n, 4) else run(wps, 0)
            assert 'scales' not in own[2], own[2]
            if wps in (1, min(n, 4)):
                for k, (x, y) in enumerate(zip(own[0], ref[0])): assert torch.equal(x, y), f'{what}: d loss / d disp[{k}] with bwd_wps={wps} differs by {(x - y).abs().max().item():.3e} ({own[2]})'
            else:
                for k, (x, y) in enumerate(zip(own[0], ref[0])): assert rel_to_max(x, y) <= 1e-6, f'{what}: d loss / d disp[{k}] with bwd_wps={wps}: {rel_to_max(x, y):.2e}'
            assert rel_to_max(own[1], ref[1]) <= 2e-6, f'{what}: d loss / d T with bwd_wps={wps}: {rel_to_max(own[1], ref[1]):.2e}'
            g, gT, label = run(wps, 1)
            seen.append('scales' in label)
            for k, (x, y) in enumerate(zip(g, own[0])): assert torch.equal(x, y), f'{what}: d loss / d disp[{k}] with bwd_wps={wps}: blocks of scales differ by {(x - y).abs().max().item():.3e} ({label})'
            assert rel_to_max(gT, own[1]) <= 2e-6, f'{what}: d loss / d T with bwd_wps={wps}, blocks of scales: {rel_to_max(gT, own[1]):.2e}'
    finally: del os.environ['SMD_BWD_SKIP']
    print(f'{what}: blocks of scales ran in {sum(seen)} of {len(seen)} compositions')


@pytest.mark.parametrize('seed', _more(24))
def test_random_shapes_decoder_convolutions(seed):
    """Round 5: the decoder's one-channel heads (`conv3x3_head`: stencil kernels, three forward instantiations chosen by size) and its thin last stage
    (`conv3x3_thin`: fp32 MFMA, tiles of 64 x 2 / 64 x 4 pixels, the 16-byte-store path for rows that allow it) on random batch / channel / image sizes
    — one-pixel-wide remainders, images smaller than a tile, channel counts off every unroll — against ATen's `conv2d` (+ sigmoid) in fp64: outputs
    and all gradients."""
    if not torch.cuda.is_available(): pytest.skip('needs a GPU')
    import torch.nn.functional as TF
    from slowtv_monodepth_amd import functional as F
    r = random.Random(8800 + seed)
    B = r.choice([1, 2, 3, 5])
    h = r.choice([2, 3, r.randint(4, 20), r.randint(21, 70), r.randint(71, 140)])
    w = r.choice([2, 3, r.randint(4, 63), 64, 65, 128, r.randint(66, 200), r.randint(201, 330)])
    gen = torch.Generator(device='cuda').manual_seed(seed)
    for kind in ('head', 'thin'):
        C = r.choice([1, 2, 3, 7, 8, 16, 33, 64, 130]) if kind == 'head' else r.choice([16, 32])
        Co = 1 if kind == 'head' else 16
        act = r.choice(['sigmoid', None]) if kind == 'head' else None
        xp = torch.randn(B, C, h + 2, w + 2, device='cuda', generator=gen)
        wt = torch.randn(Co, C, 3, 3, device='cuda', generator=gen)/(3*C**0.5)
        bs = torch.randn(1, device='cuda', generator=gen) if (kind == 'head' and r.random() < 0.7) else None
        gy = torch.randn(B, Co, h, w, device='cuda', generator=gen)
        what = f'seed {seed}: {kind} B={B} C={C} {h}x{w} act={act} bias={bs is not None}'
        L = [t.clone().requires_grad_(True) for t in (xp, wt)] + ([bs.clone().requires_grad_(True)] if bs is not None else [])
        y = F.conv3x3_head(L[0], L[1], L[2] if bs is not None else None, act) if kind == 'head' else F.conv3x3_thin(L[0], L[1])
        y.backward(gy)
        R = [t.double().clone().requires_grad_(True) for t in (xp, wt)] + ([bs.double().clone().requires_grad_(True)] if bs is not None else [])
        yr = TF.conv2d(R[0], R[1], R[2] if bs is not None else None)
        if act == 'sigmoid': yr = torch.sigmoid(yr)
        yr.backward(gy.double())
        assert rel_to_max(y.double(), yr) <= 3e-6, f'{what}: output {rel_to_max(y.double(), yr):.2e}'
        for nm, a, ref in zip(('g_xp', 'g_weight'), L, R):
            assert rel_to_max(a.grad.double(), ref.grad) <= 3e-6, f'{what}: {nm} {rel_to_max(a.grad.double(), ref.grad):.2e}'
        if bs is not None:      # one sum over every pixel, which may cancel (seed 26 of a 300-seed hunt: 0.059 from 702 terms of +-0.2): judged against the sum of magnitudes
            gp = gy.double()*(yr.detach()*(1 - yr.detach()) if act == 'sigmoid' else 1.0)
            err = (L[2].grad.double() - R[2].grad).abs().item()
            assert err <= 1e-6*gp.abs().sum().item(), f'{what}: g_bias off by {err:.2e} of {gp.abs().sum().item():.2e}'


@pytest.mark.parametrize('seed', _more(12))
def test_random_smoothness_options_match_the_oracle(seed):
    """`handlers.disp_smooth` over random image / pyramid sizes with `use_edges` and `use_laplacian` drawn at random (the first-order form is
    the streaming sweep with cached or in-launch edge weights, the second-order form the per-pixel kernels)."""
    if not torch.cuda.is_available(): pytest.skip('needs a GPU')
    import slowtv_monodepth_amd as amd
    r = random.Random(500 + seed)
    b, h, w = r.choice([1, 2, 3]), r.randint(2, 70), r.randint(2, 150)
    lows = [(h, w)] + [(r.randint(1, h), r.randint(1, w)) for _ in range(r.randint(0, 3))] if r.random() < 0.5 else [(max(h >> s, 1), max(w >> s, 1)) for s in range(r.randint(1, 4))]
    use_edges, use_lap = r.random() < 0.6, r.random() < 0.5
    gen = torch.Generator().manual_seed(seed)
    img = torch.rand(b, 3, h, w, generator=gen)
    dc = {s: (0.05 + 0.9*torch.rand(b, 1, *hw, generator=gen)).requires_grad_(True) for s, hw in enumerate(lows)}
    dg = {s: v.detach().clone().cuda().requires_grad_(True) for s, v in dc.items()}
    what = f'seed {seed}: b={b} {h}x{w} pyramid {lows} edges={use_edges} laplacian={use_lap}'
    l_ref = torch.stack([O.smooth_reg(d, O.resize_bilinear(img, d.shape[-2:]), use_edges, use_laplacian=use_lap)[0]/2**s for s, d in dc.items()]).mean()
    l_ref.backward()
    reg = amd.regularizers.SmoothReg(use_edges=use_edges, use_laplacian=use_lap)
    l_hip, _ = amd.handlers.disp_smooth(reg, dg, img.cuda(), want_aux=False)
    l_hip.backward()
    assert abs(l_hip.item() - l_ref.item()) <= 2e-5*abs(l_ref.item()) + 1e-8, f'{what}: loss {l_hip.item()} vs {l_ref.item()}'
    for s in dc:
        # |d_i - d_j| of two disparities within rounding of each other is the same kind of knife edge as the L1 term above
        e = (dg[s].grad.cpu() - dc[s].grad).abs()/dc[s].grad.abs().max().clamp(min=1e-20)
        n_out = int((e > 1e-3).sum())
        assert n_out <= 2 and e.max().item() < 0.5, f'{what}: d loss / d disp_{s}: {n_out} elements off, worst {e.max().item():.3e}'


@pytest.mark.parametrize('seed', _more(8))
def test_random_shapes_with_more_than_four_supports(seed):
    """Five to eight supports run as passes of four with a carried minimum / sum (`smd_image_recon_supports_per_pass`): the depth-input form of
    the fused operator, forward and backward, against the oracle."""
    if not torch.cuda.is_available(): pytest.skip('needs a GPU')
    from slowtv_monodepth_amd import functional as F
    r = random.Random(700 + seed)
    b, h, w, n, S = r.choice([1, 2]), r.randint(3, 60), r.randint(3, 140), r.randint(5, 8), r.choice([1, 2, 3])
    use_min, use_auto = r.random() < 0.7, r.random() < 0.6
    gen = torch.Generator().manual_seed(100 + seed)
    imgs = torch.rand(b, 3, h, w, generator=gen)
    mix = 0.5*torch.rand(1, generator=gen).item()
    supp = mix*imgs[None] + (1 - mix)*torch.rand(n, b, 3, h, w, generator=gen)
    depth = 1 + 10*torch.rand(S, b, 1, h, w, generator=gen)
    aa = 0.02*torch.randn(n*b, 3, generator=gen); t = 0.2*torch.randn(n*b, 3, generator=gen)
    K = torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]])[None].repeat(b, 1, 1)
    noise = torch.randn(S*b, 1, h, w, generator=gen)
    what = f'seed {seed}: b={b} {h}x{w} n={n} S={S} min={use_min} automask={use_auto}'
    d_c = depth.clone().requires_grad_(True); T_c = O.T_from_AAt(aa, t).unflatten(0, (n, b)).clone().requires_grad_(True)
    loss_c, _, full = O.image_recon({s: d_c[s] for s in range(S)}, imgs, supp, T_c, K, 'ssim', use_min, use_auto, noise=noise)
    loss_c.backward()
    d_g = depth.cuda().requires_grad_(True); T_g = T_c.detach().cuda().requires_grad_(True)
    loss, err, sel, _ = F.image_recon_fused(d_g, imgs.cuda(), supp.cuda(), T_g, K.cuda(), flags=F.recon_flags('ssim', use_min, use_auto), noise=noise.cuda())
    loss.backward()
    flips = (sel.cpu() != full['sel']).flatten()
    assert flips.float().mean().item() <= 0.01, f'{what}: selection differs on {flips.float().mean().item():.2%} of pixels'
    torch.testing.assert_close(err.cpu().flatten()[~flips], full['err'].detach().flatten()[~flips], rtol=0, atol=3e-4, msg=lambda m: f'{what}: error map {m}')
    torch.testing.assert_close(loss.detach().cpu(), loss_c.detach(), rtol=1e-4 if not flips.any() else 2e-3, atol=1e-6, msg=lambda m: f'{what}: loss {m}')
    tol = 5e-2 if flips.any() else 2e-3
    e = (d_g.grad.cpu() - d_c.grad).abs()/d_c.grad.abs().max().clamp(min=1e-20)
    assert int((e > tol).sum()) <= max(3, int(2e-4*e.numel())) and e.max().item() < 5e-2, f'{what}: d loss / d depth: {int((e > tol).sum())} elements off, worst {e.max().item():.3e}'
    assert rel_to_max(T_g.grad.cpu()[..., :3, :], T_c.grad[..., :3, :]) < tol, what


def test_pose_matrices_over_the_whole_angle_range():
    """`T_from_AAt` (+ the inverted rows of `always_fwd_pose`) from |aa| = 0 through the clip branch, small and ordinary angles, up to and
    beyond pi (the pose network emits 0.01 x its output, so real angles are small — the kernel must still be the reference's function
    everywhere): values and gradients against the oracle in fp64, random axes, random upstream gradients."""
    if not torch.cuda.is_available(): pytest.skip('needs a GPU')
    from slowtv_monodepth_amd import functional as F
    import math
    mags = [0.0, 1e-12, 1e-9, 1e-7, 3e-6, 1e-4, 1e-3, 0.03, 0.3, 1.0, 2.0, 3.0, math.pi - 1e-3, math.pi, math.pi + 1e-3, 3.5, 6.0, 2*math.pi, 7.0]
    gen = torch.Generator().manual_seed(77)
    N = 4*len(mags)
    ax = torch.randn(N, 3, generator=gen); ax = ax/ax.norm(dim=1, keepdim=True)
    aa = ax*torch.tensor(mags).repeat_interleave(4)[:, None]
    t = torch.randn(N, 3, generator=gen)
    inv = (torch.arange(N) % 2).to(torch.uint8)
    gT = torch.randn(N, 4, 4, generator=gen)
    aa_c, t_c = aa.double().requires_grad_(True), t.double().requires_grad_(True)
    T_c = O.T_from_AAt(aa_c, t_c)
    T_c = torch.stack([torch.linalg.inv(Ti) if f else Ti for Ti, f in zip(T_c, inv)])
    T_c.backward(gT.double())
    aa_g, t_g = aa.cuda().requires_grad_(True), t.cuda().requires_grad_(True)
    T_g = F.pose_matrices(aa_g, t_g, inv.cuda())
    T_g.backward(gT.cuda())
    torch.testing.assert_close(T_g.detach().cpu().double(), T_c.detach(), rtol=2e-5, atol=2e-6)
    # d/d aa: relative to the largest entry of the row (the gradient's scale grows with the upstream gradient, not with |aa|)
    for name, a, c in (('aa', aa_g.grad.cpu().double(), aa_c.grad), ('t', t_g.grad.cpu().double(), t_c.grad)):
        e = (a - c).abs()/c.abs().amax(dim=1, keepdim=True).clamp(min=1e-3)
        worst = e.amax(dim=1)
        bad = (worst > 2e-4).nonzero().flatten().tolist()
        assert not bad, f'd/d{name} off for rows {bad}: |aa| = {[mags[i//4] for i in bad]}, rel {[f"{worst[i].item():.2e}" for i in bad]}'


@pytest.mark.parametrize('seed', _more(16))
def test_random_generic_channel_operators(seed):
    """`ViewSynth` on C-channel inputs + `PhotoError` ('ssim' / 'l1' / 'l2') + `RegressionLoss`, the un-fused operators the other handlers
    (feat_recon, depth_regr, stereo_const, the hints tool) are built from: random batch, channels, size; values and gradients against the oracle."""
    if not torch.cuda.is_available(): pytest.skip('needs a GPU')
    from slowtv_monodepth_amd import functional as F
    r = random.Random(300 + seed)
    B, C, h, w = r.randint(1, 4), r.choice([1, 2, 3, 4, 7, 16, 33]), r.randint(2, 50), r.randint(2, 130)
    loss_name = r.choice(['ssim', 'l1', 'l2'])
    regr, invert, use_mask = r.choice(['l1', 'log_l1', 'berhu']), r.random() < 0.5, r.random() < 0.5
    gen = torch.Generator().manual_seed(seed)
    inp = torch.rand(B, C, h, w, generator=gen); tgt = torch.rand(B, C, h, w, generator=gen)
    depth = 0.5 + 8*torch.rand(B, 1, h, w, generator=gen)
    aa = 0.03*torch.randn(B, 3, generator=gen); t = 0.2*torch.randn(B, 3, generator=gen)
    K = torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]])[None].repeat(B, 1, 1)
    g_err = torch.randn(B, 1, h, w, generator=gen)
    mask = (torch.rand(B, 1, h, w, generator=gen) > 0.3).float() if use_mask else None
    what = f'seed {seed}: B={B} C={C} {h}x{w} {loss_name} regr={regr} invert={invert} mask={use_mask}'

    def run(dev, M):
        i_, d_, T_ = inp.to(dev).requires_grad_(True), depth.to(dev).requires_grad_(True), O.T_from_AAt(aa, t).to(dev).requires_grad_(True)
        warp, dwarp, valid = M.view_synth(i_, d_, T_, K.to(dev))
        err = M.photo_error(warp, tgt.to(dev), loss_name)
        m = None if mask is None else (mask.bool().to(dev) if dev == 'cuda' else mask)   # the HIP operator takes a 0/1 mask as bool
        l_regr = (M.regression_loss(dwarp, d_.detach() + 0.3, m, loss_name=regr, invert=invert)[0] if dev == 'cuda'
                  else M.regression_loss(dwarp, d_.detach() + 0.3, m, regr, invert)[0])
        ((err*g_err.to(dev)).mean() + l_regr).backward()
        return warp.detach().cpu(), dwarp.detach().cpu(), valid.cpu(), err.detach().cpu(), l_regr.detach().cpu(), i_.grad.cpu(), d_.grad.cpu(), T_.grad.cpu()[..., :3, :]
    hip, ref = run('cuda', F), run('cpu', O)
    torch.testing.assert_close(hip[0], ref[0], rtol=0, atol=1e-4, msg=lambda m: f'{what}: warp {m}')   # a few ulps of a coordinate near 100 x the slope of a random image (as test_gpu_parity.py)
    torch.testing.assert_close(hip[1], ref[1], rtol=2e-5, atol=1e-5, msg=lambda m: f'{what}: depth_warp {m}')
    assert (hip[2] != ref[2].bool()).float().mean().item() <= 2e-3, f'{what}: mask_valid'
    torch.testing.assert_close(hip[3], ref[3], rtol=0, atol=3e-4, msg=lambda m: f'{what}: error map {m}')
    torch.testing.assert_close(hip[4], ref[4], rtol=1e-4, atol=1e-6, msg=lambda m: f'{what}: regression loss {m}')
    for name, a, c in zip(('input', 'depth', 'T'), hip[5:], ref[5:]):
        e = (a - c).abs()/c.abs().max().clamp(min=1e-20)
        n_out = int((e > 2e-3).sum())    # isolated elements: sign(0) of an L1 term, berHu's |d| = delta switch, a sample on the border
        assert n_out <= max(3, int(3e-4*e.numel())) and e.max().item() < 0.2, f'{what}: d/d {name}: {n_out} of {e.numel()} elements off, worst {e.max().item():.3e}'


@pytest.mark.parametrize('seed', _more(12))
def test_random_crop_resize_windows(seed):
    """`smd_crop_resize` (aspect-ratio augmentation) at random image, window and output sizes against the oracle's restatement of
    kornia's `center_crop(align_corners=False)` + `F.interpolate` (crop half parity-unpinned: kornia is absent)."""
    if not torch.cuda.is_available(): pytest.skip('needs a GPU')
    from slowtv_monodepth_amd import functional as F
    from oracle import aspect_ratio_oracle as A
    r = random.Random(900 + seed)
    H, W = r.randint(2, 80), r.randint(2, 150)
    crop = (r.randint(2, H), r.randint(2, W)) if r.random() < 0.8 else (H, W)
    out = (r.randint(2, 96), r.randint(2, 160)) if r.random() < 0.8 else crop
    gen = torch.Generator().manual_seed(seed)
    tens = [torch.rand(2, 3, H, W, generator=gen), torch.rand(2, 2, 3, H, W, generator=gen), torch.rand(2, 1, H, W, generator=gen)]
    Kc = torch.rand(2, 4, 4, generator=gen)
    o_hip, K_hip = F.crop_resize([t.cuda() for t in tens], crop, out, Kc.cuda())
    o_ref, K_ref = A.crop_resize(tens, crop, out, Kc)
    for a, c in zip(o_hip, o_ref): torch.testing.assert_close(a.cpu(), c, rtol=1e-5, atol=3e-5, msg=lambda m: f'seed {seed}: {H}x{W} crop {crop} -> {out}: {m}')
    torch.testing.assert_close(K_hip.cpu(), K_ref, rtol=1e-6, atol=1e-6)


def test_kitti_raw_size_matches_the_oracle():
    """A full-resolution KITTI raw frame (375 x 1242, odd on both axes; the BASELINE configs train at 192 x 640 / 384 x 640): 21 strip columns,
    24 strips, a pyramid whose levels are floor halvings of odd sizes — the whole path against the oracle once at a size no other test reaches."""
    if not torch.cuda.is_available(): pytest.skip('needs a GPU')
    from slowtv_monodepth_amd import functional as F
    b, h, w, n = 1, 375, 1242, 2
    lows = [(h >> s, w >> s) for s in range(4)]
    gen = torch.Generator().manual_seed(4242)
    low = lambda c, hh, ww: torch.nn.functional.interpolate(torch.rand(b, c, 6, 20, generator=gen), size=(hh, ww), mode='bilinear', align_corners=False)
    imgs = (low(3, h, w) + 0.05*torch.rand(b, 3, h, w, generator=gen)).clamp(0, 1)
    supp = torch.stack([(imgs.roll(shifts=(1, 3*(k + 1)), dims=(-2, -1)) + 0.02*torch.rand(b, 3, h, w, generator=gen)).clamp(0, 1) for k in range(n)])
    disps = {s: (0.1 + 0.8*low(1, hs, ws)) for s, (hs, ws) in enumerate(lows)}
    aa = 0.005*torch.randn(n*b, 3, generator=gen); t = 0.05*torch.randn(n*b, 3, generator=gen)
    K = torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]])[None].repeat(b, 1, 1)
    noise = torch.randn(4*b, 1, h, w, generator=gen)
    dc = {s: d.clone().requires_grad_(True) for s, d in disps.items()}
    Tc = O.T_from_AAt(aa, t).unflatten(0, (n, b)).clone().requires_grad_(True)
    loss_c, out_c = O.loss_path(dc, imgs, supp, Tc, K, min_depth=0.1, max_depth=100, loss_name='ssim', use_min=True, use_automask=True, use_edges=True, w_smooth=0.001, noise=noise)
    loss_c.backward()
    dg = [d.cuda().requires_grad_(True) for d in disps.values()]
    Tg = Tc.detach().cuda().requires_grad_(True)
    l_rec, err, sel, _, dep = F.image_recon_fused_disp(dg, imgs.cuda(), supp.cuda(), Tg, K.cuda(), flags=F.recon_flags('ssim', True, True), min_depth=0.1, max_depth=100,
                                                       noise=noise.cuda(), want_err=True)
    l_sm, _, _ = F.disp_smooth_fused(dict(enumerate(dg)), imgs.cuda(), use_edges=True, want_aux=False)
    (l_rec + 0.001*l_sm).backward()
    full = out_c['full']
    flips = (sel.cpu() != full['sel']).flatten()
    assert flips.float().mean().item() <= 1e-3, f'selection differs on {flips.float().mean().item():.3%} of the pixels'
    assert 0.02 < (full['sel'] == 255).float().mean().item() < 0.98, 'the case should exercise both the automask and the supports'
    for s in range(4): torch.testing.assert_close(dep[s].cpu(), out_c['depth_up'][s].detach(), rtol=2e-5, atol=1e-5)
    torch.testing.assert_close(err.cpu().flatten()[~flips], full['err'].detach().flatten()[~flips], rtol=0, atol=3e-4)
    torch.testing.assert_close(l_rec.detach().cpu(), out_c['loss_img_recon'].detach(), rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(l_sm.detach().cpu(), out_c['loss_disp_smooth'].detach(), rtol=2e-5, atol=1e-7)
    tol = 5e-2 if flips.any() else 2e-3
    for s in range(4):
        e = (dg[s].grad.cpu() - dc[s].grad).abs()/dc[s].grad.abs().max().clamp(min=1e-20)
        assert int((e > tol).sum()) <= max(3, int(2e-4*e.numel())) and e.max().item() < 0.2, f'd loss / d disp_{s}: {int((e > tol).sum())} elements off, worst {e.max().item():.3e}'
    assert rel_to_max(Tg.grad.cpu()[..., :3, :], Tc.grad[..., :3, :]) < tol
