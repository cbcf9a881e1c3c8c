"""GPU value checks of the SCHEDULING `bench.py` times (reference step: src/core/trainer.py:166-190).

`MonoDepthModule.step` runs the frame-only half of the reconstruction forward (`k_recon_prep`) ahead of the loss, on the pose
network's side stream, and hands the result to the handler as `prepared=`: the handler then sets `SMD_PACKED_READY`, waits on
an event and `record_stream`s a buffer that was allocated on another stream; `k_recon_prep` also zeroes the arrival counters the
in-launch reductions of the forward and the backward rely on.  None of this changes a single arithmetic instruction, so every
comparison below is BIT-equality against the inline placement of the same launches."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def F():
    if not torch.cuda.is_available(): pytest.skip('needs a GPU')
    from slowtv_monodepth_amd import functional
    return functional


def _cfg2_inputs(b=12, h=192, w=640, supp=(-1, 1), S=4, seed=7):
    from test_gpu_parity import _baseline_inputs
    y, disps, aa, t, _ = _baseline_inputs(b, h, w, supp, S, seed=seed)
    return y, disps, aa, t


def _handler_run(F, y, disps, aa, t, *, prepared_on, skip_env=None, monkeypatch=None):
    """One forward + backward of the trainer's loss call; `prepared_on`: None (inline prep) | 'current' | 'side'."""
    import slowtv_monodepth_amd as amd
    from slowtv_monodepth_amd.handlers import LazyDepths
    dev = 'cuda'
    b, _, h, w = y['imgs'].shape
    n = y['supp_imgs'].shape[0]
    leaf = lambda v: v.detach().clone().to(dev).requires_grad_(True)
    d = {s: leaf(v) for s, v in disps.items()}
    a_, t_ = leaf(aa), leaf(t)
    imgs, sup, K = y['imgs'].to(dev), y['supp_imgs'].to(dev), y['K'].to(dev)
    crit = amd.losses.ReconstructionLoss(loss_name='ssim', use_min=True, use_automask=True)   # fresh criterion: same noise seed in every run
    reg = amd.regularizers.SmoothReg(use_edges=True)
    flags = F.recon_flags('ssim', True, True)
    prepared = None
    if prepared_on is not None:
        side = torch.cuda.Stream() if prepared_on == 'side' else None
        prepared = F.image_recon_prep(imgs, sup, flags=flags, pyramid=[tuple(v.shape[-2:]) for v in d.values()], stream=side, smooth_edges=True)
        if side is not None:
            # unrelated traffic on the main stream between prep and its consumer: the consumer must wait for the EVENT, not be lucky
            junk = torch.empty(1 << 26, device=dev); junk.add_(1.0); del junk
    Ts = F.pose_matrices(a_.flatten(0, 1), t_.flatten(0, 1)).unflatten(0, (n, b))
    depths = LazyDepths(list(d.keys()), list(d.values()), (h, w), 0.1, 100)
    seen = {}
    real = F.image_recon_fused_disp

    def spy(*a, **kw):
        out = real(*a, **kw)
        seen.update(prepared=kw.get('prepared'), sel=out[2], depth_up=out[4])
        return out
    F.image_recon_fused_disp = spy
    try:
        l_rec, ld = amd.handlers.image_recon(crit, amd.geometry.ViewSynth((h, w)), depths, None, imgs, sup, Ts, K, want_warp=False, prepared=prepared)
    finally:
        F.image_recon_fused_disp = real
    l_sm, _ = amd.handlers.disp_smooth(reg, d, imgs, want_aux=False, prepared=prepared)   # (its edge weights ride in the same prepared object)
    loss = l_rec + 0.001*l_sm
    loss.backward()
    torch.cuda.synchronize()
    assert (seen['prepared'] is not None) == (prepared_on is not None), 'the handler dropped (or invented) the prepared frames'
    out = {'loss': loss.detach().clone(), 'l_rec': l_rec.detach().clone(), 'l_sm': l_sm.detach().clone(), 'sel': seen['sel'].clone(), 'depth_up': seen['depth_up'].detach().clone(),
           'automask': ld['automask'].clone(), 'aa': a_.grad.clone(), 't': t_.grad.clone()}
    out.update({f'disp_{s}': v.grad.clone() for s, v in d.items()})
    return out


def _assert_bit_equal(a: dict, b: dict, what: str):
    for k in a:
        assert torch.equal(a[k], b[k]), f'{what}: `{k}` differs (max abs diff {(a[k].float() - b[k].float()).abs().max().item():.3e})'


@pytest.mark.parametrize('skip', [None, '0', '2'])
def test_prepared_frames_on_a_side_stream_equal_the_inline_prep(F, monkeypatch, skip):
    """(i) + (iii): `F.image_recon_prep(..., pyramid=..., stream=side)` -> `handlers.image_recon(..., prepared=...)` against the inline
    call at cfg 2 size: loss, `sel`, `depth_up`, automask and every gradient bit-equal — with the backward's row loop chosen by the
    tuner and pinned both ways."""
    if skip is not None: monkeypatch.setenv('SMD_BWD_SKIP', skip)
    y, disps, aa, t = _cfg2_inputs()
    inline = _handler_run(F, y, disps, aa, t, prepared_on=None)
    for where in ('current', 'side'):
        ahead = _handler_run(F, y, disps, aa, t, prepared_on=where)
        _assert_bit_equal(inline, ahead, f'prepared on the {where} stream (SMD_BWD_SKIP={skip})')
    assert torch.isfinite(inline['loss']) and (inline['sel'] != 255).any() and (inline['sel'] == 255).any()


class _DepthStub(torch.nn.Module):
    """A stand-in for the depth network with the same output contract ({'disp': {s: (b,1,h>>s,w>>s) sigmoid}}) built from element-wise
    ops and fixed-order reductions only.  The real networks are NOT bit-reproducible run to run on this stack (MIOpen: two identical runs
    of the same placement differ in ~2 % of the disparities' last bits, scripts/dev/dbg_sched.py), so a bit-equality test of the loss
    path's SCHEDULING has to feed it from something that is; the stream choreography of `MonoDepthModule.step` is unchanged."""
    out_scales = [0, 1, 2, 3]

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Parameter(torch.tensor([0.9, -0.7, 0.5, 0.6])); self.b = torch.nn.Parameter(torch.tensor([-1.2, -0.9, -1.5, -1.0]))

    def forward(self, x):
        g = x.mean(1, keepdim=True)
        b, _, h, w = g.shape
        out = {}
        for s in self.out_scales:
            f = 2**s
            gs = g.view(b, 1, h//f, f, w//f, f).mean((3, 5)) if s else g
            out[s] = torch.sigmoid(self.a[s]*gs + self.b[s])
        return {'disp': out}


class _PoseStub(torch.nn.Module):
    """(N,6,h,w) image pair -> {'R', 't'} (N,2,3) like `PoseNet` (0.01-scaled), from channel means and a fixed-order weighted sum."""
    def __init__(self):
        super().__init__()
        gen = torch.Generator().manual_seed(5)
        self.W = torch.nn.Parameter(torch.randn(6, 12, generator=gen))

    def forward(self, x):
        m = x.mean((2, 3))                                       # (N,6)
        v = 0.01*(m[:, :, None]*self.W[None]).sum(1).view(-1, 2, 6)
        return {'R': v[..., :3], 't': v[..., 3:]}


def _make_module(prep_ahead, seed=0):
    import bench
    from slowtv_monodepth_amd.trainer import MonoDepthModule
    torch.manual_seed(seed)
    cfg = bench.make_cfg(bench.WORKLOADS['cfg2'])
    cfg['trainer']['prep_ahead'] = prep_ahead
    m = MonoDepthModule(cfg)
    m.nets['depth'], m.nets['pose'] = _DepthStub(), _PoseStub()   # same keys, same output contracts; see _DepthStub
    return m.cuda().train()


@pytest.mark.parametrize('skip', [None, '0', '2'])
def test_two_steps_in_flight_with_prep_ahead_equal_inline_prep(F, monkeypatch, skip):
    """(ii) + (iii): two `MonoDepthModule.step` + backward iterations back to back with NO synchronisation between them — the second
    step's prep launch, packed buffer and arrival counters are in flight while the first step's backward still reads its own —
    with `prep_ahead='pose'` (what the bench runs) against the same two steps with `prep_ahead=False`: losses, every gradient the loss
    path hands back to the networks and every parameter gradient of both steps bit-equal.  (Deterministic stand-in networks: `_DepthStub`.)"""
    from slowtv_monodepth_amd.synthetic import make_batch
    if skip is not None: monkeypatch.setenv('SMD_BWD_SKIP', skip)
    batches = [make_batch(12, 192, 640, (-1, 1), seed=42 + k, device='cuda') for k in range(2)]

    from slowtv_monodepth_amd import functional as Fm
    real_prep, real_fused, real_path = Fm.image_recon_prep, Fm.image_recon_fused_disp, Fm.loss_path_fused
    calls = {}

    def spy_prep(*a, **kw):
        calls['prep_streams'] = calls.get('prep_streams', []) + [kw.get('stream')]
        return real_prep(*a, **kw)

    def spy_fused(*a, **kw):
        calls['prepared'] = calls.get('prepared', []) + [kw.get('prepared') is not None]
        return real_fused(*a, **kw)
    def spy_path(*a, **kw):      # the single-node loss path the trainer takes for this configuration (round 5)
        calls['prepared'] = calls.get('prepared', []) + [kw.get('prepared') is not None]
        return real_path(*a, **kw)
    monkeypatch.setattr(Fm, 'image_recon_prep', spy_prep); monkeypatch.setattr(Fm, 'image_recon_fused_disp', spy_fused); monkeypatch.setattr(Fm, 'loss_path_fused', spy_path)

    def run(prep_ahead):
        calls.clear()
        m = _make_module(prep_ahead)
        ref_state = copy.deepcopy(m.state_dict())
        losses, path_grads, param_grads = [], [], []
        for batch in batches:                      # no optimizer step: the second step's gradients must not depend on float order of an update
            for p in m.parameters(): p.grad = None
            loss, ld, fwd = m.step(batch)
            outs = [fwd['disp'][s] for s in sorted(fwd['disp'])]   # what the loss path hands back to the depth network (the pose side is covered by the parameter gradients:
                                                                   # the single-node loss path differentiates through to the pose network's outputs, `fwd['Ts']` gets no gradient)
            for o in outs: o.retain_grad()
            loss.backward()
            losses.append(loss.detach())           # no .item(): nothing here waits for the device
            path_grads.append([o.grad for o in outs])
            param_grads.append([p.grad for p in m.parameters() if p.requires_grad])
        torch.cuda.synchronize()
        return m, ref_state, [l.clone() for l in losses], path_grads, param_grads, dict(calls)

    m_a, st_a, l_a, pg_a, g_a, c_a = run('pose')
    main = torch.cuda.current_stream()
    assert m_a.prep_ahead == 'pose' and c_a.get('prepared') == [True, True], f'the steps did not consume prepared frames: {c_a}'
    assert len(c_a['prep_streams']) == 2 and all(s is not None and s != main for s in c_a['prep_streams']), 'prep did not run on the side stream'
    m_i, st_i, l_i, pg_i, g_i, c_i = run(False)
    assert m_i.prep_ahead is False and c_i.get('prepared') == [False, False] and 'prep_streams' not in c_i
    _, _, l_r, pg_r, g_r, _ = run(False)          # the same placement again: is the network side itself reproducible on this box?
    nets_repeatable = all(torch.equal(x, y_) for k in range(2) for x, y_ in zip(g_i[k], g_r[k]))
    for k in st_a: assert torch.equal(st_a[k], st_i[k]), f'the two modules did not start identical: {k}'
    for k in range(2):
        assert torch.isfinite(l_a[k])
        assert torch.equal(l_a[k], l_i[k]), f'step {k}: loss {l_a[k].item():.9f} (prep ahead) vs {l_i[k].item():.9f} (inline)'
        for i, (x, y_) in enumerate(zip(pg_a[k], pg_i[k])):
            assert torch.equal(x, y_), f'step {k}: gradient #{i} out of the loss path differs (max {(x - y_).abs().max().item():.3e})'
        if nets_repeatable:
            bad = [i for i, (x, y_) in enumerate(zip(g_a[k], g_i[k])) if not torch.equal(x, y_)]
            assert not bad, f'step {k}: {len(bad)} of {len(g_a[k])} parameter gradients differ between prep-ahead and inline (first: #{bad[0]})'
        else:   # MIOpen's weight-gradient kernels use atomics on this box: identical runs differ in the last bits — compare to that spread
            for x, y_, z in zip(g_a[k], g_i[k], g_r[k]):
                spread = (y_ - z).abs().max()
                assert (x - y_).abs().max() <= 4*spread + 1e-6*y_.abs().max(), f'step {k}: a parameter gradient is further from the inline run than two inline runs are from each other'
    print(f'two steps in flight (SMD_BWD_SKIP={skip}): losses {[round(v.item(), 7) for v in l_a]}, network side bit-reproducible: {nets_repeatable}')


def test_prepared_frames_for_other_inputs_are_refused_or_ignored(F):
    """A `PreparedFrames` built for other frames must never be consumed: the functional raises, the handler falls back to inline."""
    y, disps, aa, t = _cfg2_inputs(b=2)
    dev = 'cuda'
    imgs, sup = y['imgs'].to(dev), y['supp_imgs'].to(dev)
    flags = F.recon_flags('ssim', True, True)
    other = F.image_recon_prep(imgs.clone(), sup, flags=flags, pyramid=[tuple(v.shape[-2:]) for v in disps.values()])
    Ts = F.pose_matrices(aa.to(dev).flatten(0, 1), t.to(dev).flatten(0, 1)).unflatten(0, (2, 2))
    with pytest.raises(ValueError):
        F.image_recon_fused_disp([v.to(dev) for v in disps.values()], imgs, sup, Ts, y['K'].to(dev), flags=flags, min_depth=0.1, max_depth=100, prepared=other)
    ok = _handler_run(F, y, disps, aa, t, prepared_on=None)
    assert torch.isfinite(ok['loss'])
