"""Seeded inputs for the BASELINE-resolution fixtures, reproducible BIT FOR BIT on any machine.

The small fixtures store their inputs.  At 192x640 / 384x640 that would be 5-15 MB per case, so the compact fixtures
(`meta_compact = 1`, written by `make_golden.py: run_trainer_case(compact=True)`) store only what the reference PRODUCED and
the inputs are regenerated here — by the build container when the fixture is made and by the test process when it is read.
That only pins anything if both get the same bits, so nothing below depends on a math library or on a reduction order:

  * random draws are `torch.randint` / `torch.rand` on a CPU `torch.Generator` (integer generator + exact int -> float scaling),
  * images are built in int64 (nearest up-sampling of random cells, box sums by cumulative sums, integer contrast stretch,
    integer shifts between the frames); the only floating-point operations are `(u8 + rand)/256`: one IEEE addition and an exact
    scaling by a power of two,
  * disparities are `a + b*texture + c*rand`: single IEEE element-wise operations in a fixed order,
  * the tie-break "noise" the reference draws with `randn_like` (src/losses/reconstruction.py:72) is replaced — in the
    reference run and in the tests alike — by a sum of twelve uniforms minus six (Irwin-Hall), added in a fixed order.
    It only has to be A fixed tensor of roughly unit scale: it decides ties between two errors closer than ~1e-6.

No transcendental function, no `sum()` / `mean()` over floats.  The fixture carries `chk_*` = the int64 sum of every input's
bit pattern; `expand_compact` (conftest.py) refuses a fixture whose regenerated inputs do not reproduce them.
"""
from __future__ import annotations

import torch

__all__ = ['make_inputs_exact', 'frame_shifts', 'bit_checksum', 'decoder_state', 'decoder_feats', 'decoder_out_grads', 'DECODER_KW']


def _box_sum(x: torch.Tensor, r: int) -> torch.Tensor:
    """(2r+1)x(2r+1) window sums of an int64 tensor (..., H, W) with replicated borders; exact."""
    H, W = x.shape[-2:]
    iy = torch.arange(-r, H + r).clamp(0, H - 1); ix = torch.arange(-r, W + r).clamp(0, W - 1)
    p = x[..., iy, :][..., :, ix]
    c = torch.zeros(*p.shape[:-2], p.shape[-2] + 1, p.shape[-1] + 1, dtype=torch.int64)
    c[..., 1:, 1:] = p.cumsum(-2).cumsum(-1)
    k = 2*r + 1
    return c[..., k:, k:] - c[..., :-k, k:] - c[..., k:, :-k] + c[..., :-k, :-k]


def _cells(g: torch.Generator, lead: tuple, H: int, W: int, cell: int) -> torch.Tensor:
    """Random cells of `cell` pixels, blurred twice (box radius cell/2) and stretched to 0..255 per plane; int64 (..., H, W)."""
    low = torch.randint(0, 256, (*lead, H//cell + 2, W//cell + 2), generator=g, dtype=torch.int64)
    up = low.repeat_interleave(cell, -2).repeat_interleave(cell, -1)[..., :H, :W]
    r = max(cell//2, 1)
    v = _box_sum(_box_sum(up, r), r)
    lo, hi = v.amin(dim=(-2, -1), keepdim=True), v.amax(dim=(-2, -1), keepdim=True)
    return (v - lo)*255//(hi - lo).clamp(min=1)


def _scene_u8(g: torch.Generator, b: int, H: int, W: int) -> torch.Tensor:
    """A smooth three-octave RGB texture, int64 in 0..255, (b, 3, H, W)."""
    return (4*_cells(g, (b, 3), H, W, 32) + 2*_cells(g, (b, 3), H, W, 12) + _cells(g, (b, 3), H, W, 4))//7


def _irwin_hall(g: torch.Generator, shape) -> torch.Tensor:
    acc = torch.rand(shape, generator=g)
    for _ in range(11): acc = acc + torch.rand(shape, generator=g)
    return acc - 6.0


def bit_checksum(t: torch.Tensor) -> int:
    t = t.detach().contiguous()
    if t.dtype == torch.float32: t = t.view(torch.int32)
    return int(t.to(torch.int64).sum())


def frame_shifts(n: int):
    """Integer (dx, dy) by which support frame i looks at the scene: (-2,0), (+2,0), (-3,-1), (+3,+1), ..."""
    return [((2 + i//2)*(1 if i % 2 else -1), (1 if i % 2 else -1)*(i//2)) for i in range(n)]


def make_inputs_exact(seed: int, b: int, h: int, w: int, n: int, scales, learn_K: bool = False):
    """-> dict(imgs (b,3,h,w), supp_imgs (n,b,3,h,w), disp {s: (b,1,h>>s,w>>s)}, K (b,4,4), noise (S*b,1,h,w)).
    Pose / intrinsics leaves (`aa`, `t`, `fs`, `cs`: a few dozen numbers drawn with `randn`) are stored in the fixture instead."""
    g = torch.Generator().manual_seed(seed)
    pad = 8
    scene = _scene_u8(g, b, h + 2*pad, w + 2*pad)

    def frame(dx, dy):
        f = scene[..., pad + dy: pad + dy + h, pad + dx: pad + dx + w] + torch.randint(0, 8, (b, 3, h, w), generator=g, dtype=torch.int64)
        # + a uniform fraction of a grey level: no two pixels are EXACTLY equal.  (On 8-bit plateaus a bilinear blend of equal values
        # reproduces the target to an ulp, and sign(pred - target) of the L1 term — 0 in one implementation, +-1 in another — is then
        # decided by the order of the blend's roundings: a subgradient convention, not arithmetic.)
        return (f.clamp(0, 255).to(torch.float32) + torch.rand(b, 3, h, w, generator=g))/256.0
    imgs = frame(0, 0)
    supp = torch.stack([frame(dx, dy) for dx, dy in frame_shifts(n)])
    # one depth field, seen at every pyramid level (block sums of the full-resolution field) + a little per-pixel roughness
    field = _cells(g, (b, 1), h, w, 32)*3 + _cells(g, (b, 1), h, w, 8)
    disp = {}
    for s in scales:
        f = 1 << s
        hs, ws = max(h >> s, 1), max(w >> s, 1)
        blk = field[..., :hs*f, :ws*f].reshape(b, 1, hs, f, ws, f).sum(dim=(3, 5))//(f*f) if f > 1 and hs*f <= h and ws*f <= w else field[..., :hs, :ws]
        disp[s] = 0.03 + (blk.to(torch.float32)/1020.0)*0.5 + 0.02*torch.rand(b, 1, hs, ws, generator=g)
    K = torch.tensor([[0.58*w, 0, 0.5*w, 0], [0, 1.92*h, 0.5*h, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=torch.float32)[None].repeat(b, 1, 1)
    noise = _irwin_hall(g, (len(scales)*b, 1, h, w))
    return dict(imgs=imgs, supp_imgs=supp, disp=disp, K=K, noise=noise)


# --------------------------------------------------------------------------------------------------
# The decoder fixture (`net_decoder_64x96.npz`): weights, encoder features and output gradients by name and seed, so that the fixture holds only what
# the reference's `MonodepthDecoder` PRODUCED.  `torch.randn` on a CPU generator is reproducible for a given torch build (the build container and the GPU
# box run the same image); `chk_*` in the fixture are the bit checksums of what was drawn, and the test refuses to compare if they differ.
# --------------------------------------------------------------------------------------------------
DECODER_KW = dict(num_ch_enc=[64, 64, 128, 256, 512], enc_sc=[2, 4, 8, 16, 32], out_sc=[0, 1, 2, 3], out_ch=1, out_act='sigmoid')


def decoder_state(shapes: dict, seed: int = 77) -> dict:
    """{reference key: tensor} for {reference key: shape}: keys in sorted order, weights ~ N(0, 1/fan_in), biases ~ 0.1 N(0, 1)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(shapes):
        shape = tuple(shapes[k])
        fan = 1
        for d in shape[1:]: fan *= d
        v = torch.randn(shape, generator=g)
        out[k] = v/float(fan)**0.5 if len(shape) > 1 else 0.1*v
    return out


def decoder_feats(seed: int = 78, b: int = 2, h: int = 64, w: int = 96) -> list:
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(b, c, h//s, w//s, generator=g) for c, s in zip(DECODER_KW['num_ch_enc'], DECODER_KW['enc_sc'])]


def decoder_out_grads(seed: int = 79, b: int = 2, h: int = 64, w: int = 96) -> dict:
    g = torch.Generator().manual_seed(seed)
    return {i: torch.randn(b, 1, h >> i, w >> i, generator=g) for i in DECODER_KW['out_sc']}
