"""CPU oracle for the self-supervised view-synthesis loss path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32 or fp64) restatement of the algorithm that jspenmar/slowtv_monodepth
runs in `src/losses`, `src/regularizers/smooth.py`, `src/tools/geometry.py` and the two handlers that drive
them.  It exists to CHECK the HIP kernels; nothing in `slowtv_monodepth_amd/` may import it.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use it.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function below against the vectors in
`tests/golden/*.npz`, which were produced by importing the reference itself (`tests/golden/make_golden.py`).
The reference ships no tests of its own for this path, so those vectors are the pins.

Every function cites the reference lines it restates (paths relative to the reference checkout).  The math is
written out per pixel (explicit 4-tap gathers, explicit 3x3 window sums) instead of calling grid_sample /
avg_pool2d, so that the oracle documents the algorithm the kernels implement; `aten=True` switches the two
heavy primitives to the ATen ops the reference itself calls (used for the timed CPU baseline).
"""
from __future__ import annotations


import torch
import torch.nn.functional as F

EPS32 = float(torch.finfo(torch.float32).eps)  # src/tools/ops.py:63-66 (always evaluated on fp32 tensors there)
SSIM_C1 = 0.01**2  # src/losses/photometric.py:30
SSIM_C2 = 0.03**2  # src/losses/photometric.py:31
W_SSIM = 0.85      # src/losses/reconstruction.py:38 -> PhotoError(weight_ssim=0.85)


# ---------------------------------------------------------------------------------------------------
# a1/a2: upsample + disparity -> depth
# ---------------------------------------------------------------------------------------------------
def _src_index(n_out: int, n_in: int, dtype, device):
    """ATen `area_pixel_compute_source_index` for bilinear, align_corners=False (called at src/tools/ops.py:314).
    ATen computes `scale = in/out` in the tensor's dtype and evaluates `scale*(dst + 0.5) - 0.5` as ONE fused multiply-add (its CPU vector
    kernels are built with FMA, its CUDA/HIP kernels contract the expression): one rounding, not two.  With two roundings the coordinate of
    a wide image (ulp 1.5e-5 beyond column 128) moves lambda by up to 1e-5 against `F.interpolate` — found by tests/test_gpu_fuzz.py at
    pyramid ratios that are not exact halvings; the single-rounding form below agrees with `F.interpolate` to one ulp of the result."""
    dst = torch.arange(n_out, dtype=torch.float64, device=device)
    if dtype == torch.float64: src = (dst + 0.5)*(n_in/n_out) - 0.5
    else:
        scale = (torch.tensor(float(n_in), dtype=dtype)/torch.tensor(float(n_out), dtype=dtype)).double().to(device)
        src = ((dst + 0.5)*scale - 0.5).to(dtype)    # exact product and sum in fp64, rounded once = fma in `dtype`
    src = src.clamp(min=0)
    i0 = src.floor().long().clamp(max=n_in - 1)
    i1 = (i0 + 1).clamp(max=n_in - 1)
    l1 = src - i0.to(dtype)
    return i0, i1, l1


def resize_bilinear(x: torch.Tensor, size: tuple[int, int], aten: bool = False) -> torch.Tensor:
    """`ops.interpolate_like(x, other, mode='bilinear')` = F.interpolate(align_corners=False) (src/tools/ops.py:311-314)."""
    if aten: return F.interpolate(x, size=size, mode='bilinear', align_corners=False)
    h_in, w_in = x.shape[-2:]
    y0, y1, ly = _src_index(size[0], h_in, x.dtype, x.device)
    x0, x1, lx = _src_index(size[1], w_in, x.dtype, x.device)
    ly, lx = ly[:, None], lx[None, :]
    top = x[..., y0, :][..., :, x0]*(1 - lx) + x[..., y0, :][..., :, x1]*lx
    bot = x[..., y1, :][..., :, x0]*(1 - lx) + x[..., y1, :][..., :, x1]*lx
    return top*(1 - ly) + bot*ly


def to_inv(depth: torch.Tensor) -> torch.Tensor:
    """src/tools/geometry.py:86-90 — `(d > 0) / d.clamp(min=eps)`."""
    return (depth > 0).to(depth.dtype)/depth.clamp(min=EPS32)


def to_scaled(disp: torch.Tensor, min_depth: float = 0.01, max_depth: float | None = 100):
    """src/tools/geometry.py:62-76 — sigmoid disparity -> (scaled disparity, depth)."""
    if min_depth <= 0: raise ValueError(f'Min depth must be greater than 0. ({min_depth})')
    if max_depth and max_depth < min_depth: raise ValueError(f'Max depth must be greater than min. ({max_depth} vs. {min_depth})')
    i_max, i_min = 1/min_depth, (1/max_depth) if max_depth else 0
    d = (i_max - i_min)*disp + i_min
    return d, to_inv(d)


def disp_to_depth_up(disps: dict, size: tuple[int, int], min_depth=None, max_depth=None, aten=False):
    """src/core/trainer.py:316-321 — per scale: bilinear upsample to the input size, then `to_depth` (trainer.py:46-49)."""
    disp_up = {s: resize_bilinear(d, size, aten=aten) for s, d in disps.items()}
    if min_depth or max_depth: depth_up = {s: to_scaled(d, min_depth, max_depth)[1] for s, d in disp_up.items()}
    else: depth_up = {s: to_inv(d) for s, d in disp_up.items()}
    return disp_up, depth_up


# ---------------------------------------------------------------------------------------------------
# a3/a4: pose + intrinsics prologue
# ---------------------------------------------------------------------------------------------------
def T_from_AAt(aa: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """src/tools/geometry.py:181-209 (Rodrigues, with decompose_AA :135-140).  aa, t: (*, 3) -> (*, 4, 4)."""
    if aa.shape[-1] != 3 or t.shape[-1] != 3 or aa.shape != t.shape: raise ValueError('aa and t must both be (*, 3)')
    angle = aa.norm(p=2, dim=-1, keepdim=True)
    axis = aa/angle.clamp(min=EPS32)
    x, y, z = axis.unbind(-1)
    zr = torch.zeros_like(x)
    W = torch.stack([torch.stack([zr, -z, y], -1), torch.stack([z, zr, -x], -1), torch.stack([-y, x, zr], -1)], -2)  # (*,3,3)
    ang = angle.unsqueeze(-1)
    Rm = torch.eye(3, dtype=aa.dtype, device=aa.device) + W*ang.sin() + (W @ W)*(1 - ang.cos())
    T = torch.zeros(aa.shape[:-1] + (4, 4), dtype=aa.dtype, device=aa.device)
    T[..., 3, 3] = 1
    T = T.clone()
    T[..., :3, :3] = Rm
    T[..., :3, 3] = t
    return T


def build_K(fs: torch.Tensor, cs: torch.Tensor) -> torch.Tensor:
    """src/networks/pose.py:60-73 — normalised focal/principal point (b,2) -> (b,4,4)."""
    K = torch.eye(4, dtype=fs.dtype, device=fs.device).repeat(fs.shape[0], 1, 1)
    K[:, 0, 0] = fs[:, 0]; K[:, 1, 1] = fs[:, 1]; K[:, 0, 2] = cs[:, 0]; K[:, 1, 2] = cs[:, 1]
    return K


def resize_K(K: torch.Tensor, new_shape: tuple[int, int], shape: tuple[int, int] | None = None) -> torch.Tensor:
    """src/tools/geometry.py:249-263 — rows 0/1 scaled by the width/height ratio."""
    if shape is None: shape = (1, 1)
    K = K.clone()
    K[..., 0, :] = K[..., 0, :]*(new_shape[1]/shape[1])
    K[..., 1, :] = K[..., 1, :]*(new_shape[0]/shape[0])
    return K


# ---------------------------------------------------------------------------------------------------
# a6-a8: view synthesis
# ---------------------------------------------------------------------------------------------------
def _clamp_zero_grad(x: torch.Tensor, hi: float) -> torch.Tensor:
    """Border clamp of a sampling coordinate with ATen `clip_coordinates_set_grad` semantics:
    value clamped to [0, hi]; gradient 1 strictly inside (0, hi), 0 otherwise (borders count as outside)."""
    inside = (x > 0) & (x < hi)
    return torch.where(inside, x, x.detach().clamp(0, hi))


def sample_coords(depth, T, K, K_inv=None):
    """Backproject -> rigid transform -> project (src/tools/geometry.py:304-316, :386, :329-350) and the
    un-normalisation grid_sample applies for align_corners=False (ATen grid_sampler_unnormalize).

    depth (B,1,h,w); T, K (B,4,4).  Returns sx, sy (B,h,w) in support-pixel units BEFORE the border clamp,
    the transformed-frame depth (B,1,h,w) and the normalised grid (B,h,w,2)."""
    B, _, h, w = depth.shape
    if K_inv is None: K_inv = torch.linalg.inv(K)
    dt, dev = depth.dtype, depth.device
    v, u = torch.meshgrid(torch.arange(h, dtype=dt, device=dev), torch.arange(w, dtype=dt, device=dev), indexing='ij')
    pix = torch.stack([u, v, torch.ones_like(u)]).reshape(1, 3, h*w)                # geometry.py:299-301
    X = (K_inv[:, :3, :3] @ pix)*depth.reshape(B, 1, h*w)                           # :313-314
    Y = T[:, :3, :3] @ X + T[:, :3, 3:4]                                            # :315 (homogeneous 1) + :386
    z = Y[:, 2:3].clamp(min=EPS32)                                                  # :340
    p = (K[:, :3, :3] @ (Y/z.clamp(min=0.1)))[:, :2]                                # :341
    gx = (p[:, 0]/(w - 1) - 0.5)*2                                                  # :347,349
    gy = (p[:, 1]/(h - 1) - 0.5)*2                                                  # :348,349
    sx = ((gx + 1)*w - 1)/2                                                         # grid_sample, align_corners=False
    sy = ((gy + 1)*h - 1)/2
    grid = torch.stack([gx, gy], -1).reshape(B, h, w, 2)
    return sx.reshape(B, h, w), sy.reshape(B, h, w), z.reshape(B, 1, h, w), grid


def bilinear_border_sample(img, sx, sy):
    """4-tap bilinear gather with border clamping (F.grid_sample(mode='bilinear', padding_mode='border',
    align_corners=False), src/tools/geometry.py:364).  img (B,C,h,w); sx, sy (B,h,w) un-clamped pixel coords."""
    B, C, h, w = img.shape
    sx = _clamp_zero_grad(sx, w - 1.0)
    sy = _clamp_zero_grad(sy, h - 1.0)
    x0f, y0f = sx.detach().floor(), sy.detach().floor()
    fx, fy = (sx - x0f)[:, None], (sy - y0f)[:, None]
    x0, y0 = x0f.long(), y0f.long()
    x1, y1 = (x0 + 1).clamp(max=w - 1), (y0 + 1).clamp(max=h - 1)  # out-of-range taps carry weight 0
    flat = img.reshape(B, C, h*w)

    def tap(yy, xx): return flat.gather(2, (yy*w + xx).reshape(B, 1, -1).expand(-1, C, -1)).reshape(B, C, *sx.shape[1:])

    nw, ne, sw, se = tap(y0, x0), tap(y0, x1), tap(y1, x0), tap(y1, x1)
    return nw*(1 - fx)*(1 - fy) + ne*fx*(1 - fy) + sw*(1 - fx)*fy + se*fx*fy


def view_synth(inp, depth, T, K, K_inv=None, aten=False):
    """`ViewSynth.forward` (src/tools/geometry.py:366-391) -> (warp, depth_warp, mask_valid)."""
    sx, sy, z, grid = sample_coords(depth, T, K, K_inv)
    mask_valid = (grid.abs() < 1).all(dim=-1, keepdim=True).permute(0, 3, 1, 2)    # :388
    if aten: warp = F.grid_sample(inp, grid, mode='bilinear', padding_mode='border', align_corners=False)
    else: warp = bilinear_border_sample(inp, sx, sy)
    return warp, z, mask_valid


# ---------------------------------------------------------------------------------------------------
# a9: photometric error
# ---------------------------------------------------------------------------------------------------
def _box3_reflect(x: torch.Tensor, aten: bool) -> torch.Tensor:
    """ReflectionPad2d(1) then AvgPool2d(3, 1) (src/losses/photometric.py:27-28, :40-45)."""
    xp = F.pad(x, (1, 1, 1, 1), mode='reflect')
    if aten: return F.avg_pool2d(xp, 3, 1)
    h, w = x.shape[-2:]
    acc = 0
    for dy in range(3):
        for dx in range(3): acc = acc + xp[..., dy:dy + h, dx:dx + w]
    return acc/9


def ssim_error(pred, target, aten=False):
    """`SSIMError.forward` (src/losses/photometric.py:33-51): (b,c,h,w) per-channel error in [0,1]."""
    mu_x, mu_y = _box3_reflect(pred, aten), _box3_reflect(target, aten)
    sig_x = _box3_reflect(pred**2, aten) - mu_x**2
    sig_y = _box3_reflect(target**2, aten) - mu_y**2
    sig_xy = _box3_reflect(pred*target, aten) - mu_x*mu_y
    num = (2*mu_x*mu_y + SSIM_C1)*(2*sig_xy + SSIM_C2)
    den = (mu_x**2 + mu_y**2 + SSIM_C1)*(sig_x + sig_y + SSIM_C2)
    return ((1 - num/den)/2).clamp(min=0, max=1)


def photo_error(pred, target, loss_name='ssim', aten=False, weight_ssim=W_SSIM):
    """(N,3,h,w) x2 -> (N,1,h,w).  'ssim': PhotoError(weight_ssim) (photometric.py:65-88: each term is skipped when its weight
    is 0); 'l1': DenseL1Error (:11-14); 'l2': DenseL2Error (:17-20).  Selection table at src/losses/reconstruction.py:37-41."""
    if loss_name == 'ssim':
        out = pred.new_zeros((pred.shape[0], 1, *pred.shape[-2:]))
        if weight_ssim > 0: out = out + weight_ssim*ssim_error(pred, target, aten).mean(dim=1, keepdim=True)
        if 1 - weight_ssim > 0: out = out + (1 - weight_ssim)*(pred - target).abs().mean(dim=1, keepdim=True)
        return out
    if loss_name == 'l1': return (pred - target).abs().mean(dim=1, keepdim=True)
    if loss_name == 'l2': return (pred - target).pow(2).sum(dim=1, keepdim=True).clamp(min=EPS32).sqrt()
    raise KeyError(loss_name)


# ---------------------------------------------------------------------------------------------------
# a10: reconstruction loss (min / mean reprojection, automask)
# ---------------------------------------------------------------------------------------------------
def apply_mask(err, mask, mask_name):
    """`ReconstructionLoss.apply_mask` (src/losses/reconstruction.py:46-57): err (B,n,h,w), mask (B,n|1,h,w)."""
    if mask_name and mask is None: raise ValueError("Must provide a 'mask' when masking...")
    if mask_name == 'explainability': return err*mask
    if mask_name == 'uncertainty': return err*(-mask).exp() + mask
    return err


def compute_photo(pred, target, loss_name='ssim', use_min=False, aten=False, mask=None, mask_name=None):
    """`ReconstructionLoss.compute_photo` (src/losses/reconstruction.py:79-96).
    pred (n,B,3,h,w) or (B,3,h,w); target (B,3,h,w) -> reduced error (B,1,h,w) and per-support errors (B,n,h,w)."""
    if pred.ndim == 4: pred = pred[None]
    n, B = pred.shape[:2]
    tgt = target[None].expand_as(pred)
    err = photo_error(pred.flatten(0, 1), tgt.flatten(0, 1), loss_name, aten)             # (n*B,1,h,w)
    err = err.squeeze(1).unflatten(0, (n, B)).permute(1, 0, 2, 3)                          # (B,n,h,w)
    err = apply_mask(err, mask, mask_name)                                                 # :94
    red = err.min(dim=1, keepdim=True)[0] if use_min else err.mean(dim=1, keepdim=True)    # :43-44
    return red, err


def recon_loss(pred, target, source=None, loss_name='ssim', use_min=False, use_automask=False, noise=None, aten=False,
               force_sel=None, mask=None, mask_name=None):
    """`ReconstructionLoss.forward` (src/losses/reconstruction.py:98-126).
    `noise` replaces the `torch.randn_like` draw of :72 (must be (B,1,h,w)); pass None to draw it here.
    Returns loss, dict(automask, err (after automask), err_warp, sel) — sel: index of the winning support, or
    255 where the static (un-warped) error won.
    `force_sel` (B,1,h,w uint8; test aid, not part of the reference): take the min-reprojection / automask decisions from this
    map instead of the arg-min.  The parity tests use it to compare GRADIENTS under identical routing when a handful of
    near-ties (errors equal to ~1e-7) are decided differently by fp32 rounding; out['tie_gap'] reports how close they were."""
    err_warp, per = compute_photo(pred, target, loss_name, use_min, aten, mask, mask_name)
    sel = per.argmin(dim=1, keepdim=True) if use_min else torch.zeros_like(err_warp, dtype=torch.long)
    out = {}
    if force_sel is not None and use_min:
        pick = force_sel.long().clamp(max=per.shape[1] - 1)
        forced = per.gather(1, pick)
        warped = force_sel != 255
        out['tie_gap'] = ((forced - err_warp)*warped).detach()       # >= 0: how much worse the forced support is than the best one
        err_warp = torch.where(warped, forced, err_warp)
        sel = torch.where(warped, pick, sel)
    out['err_warp'] = err_warp
    err = err_warp
    if use_automask:
        if source is None: raise ValueError("Must provide the original 'source' images when automasking...")
        err_static, _ = compute_photo(source, target, loss_name, use_min, aten, mask, mask_name)   # :70 (the mask weights the identity error too)
        if noise is None: noise = torch.randn_like(err_static)
        err_static = err_static + EPS32*noise                                                # :72
        err, idx = torch.min(torch.cat((err_warp, err_static), dim=1), dim=1, keepdim=True)  # :74-75
        if force_sel is not None:
            masked = force_sel == 255
            gap = torch.where(masked, err_static - err, err_warp - err).detach()
            out['tie_gap'] = torch.maximum(out['tie_gap'], gap) if 'tie_gap' in out else gap
            err, idx = torch.where(masked, err_static, err_warp), masked.long()
        out['automask'] = idx == 0                                                           # :76
        sel = torch.where(out['automask'], sel, torch.full_like(sel, 255))
    out['err'] = err
    out['sel'] = sel.to(torch.uint8)
    return err.mean(), out                                                                   # :125


# ---------------------------------------------------------------------------------------------------
# a12: smoothness
# ---------------------------------------------------------------------------------------------------
def _abs_fwd_diff(x):
    """`compute_grad` (src/regularizers/smooth.py:12-30): |x - x_right|, |x - x_below| with a zero last col/row."""
    dx = torch.zeros_like(x); dy = torch.zeros_like(x)
    dx[..., :, :-1] = (x[..., :, :-1] - x[..., :, 1:]).abs()
    dy[..., :-1, :] = (x[..., :-1, :] - x[..., 1:, :]).abs()
    return dx, dy


def gaussian_blur3x3(x):
    """`kornia.filters.gaussian_blur2d(x, kernel_size=(3, 3), sigma=(1, 1))` as src/regularizers/smooth.py:21 calls it.  kornia is absent from
    the build image (PARITY UNPINNED for this function); restated from kornia 0.6.10's published source: `get_gaussian_kernel1d(3, 1.0)` =
    exp(-d^2 / 2) / sum over d in {-1, 0, 1}; `filter2d_separable(x, k[None], k[None], border_type='reflect')` = F.pad(mode='reflect') by one and a
    depth-wise `F.conv2d` with the 1x3 kernel, then the same with the 3x1 kernel (cross-correlation; the kernel is symmetric)."""
    B, C, h, w = x.shape
    d = torch.arange(3, dtype=x.dtype, device=x.device) - 1
    k = torch.exp(-d.pow(2)/2.0); k = k/k.sum()
    kx = k.view(1, 1, 1, 3).expand(C, 1, 1, 3); ky = k.view(1, 1, 3, 1).expand(C, 1, 3, 1)
    out = F.conv2d(F.pad(x, (1, 1, 0, 0), mode='reflect'), kx, groups=C)
    return F.conv2d(F.pad(out, (0, 0, 1, 1), mode='reflect'), ky, groups=C)


def _laplacian(x, blur=lambda t: t):
    """`compute_laplacian(x)[:2]` (src/regularizers/smooth.py:33-48): (|d/dx |dx||, |d/dy |dy||); with use_blur every `compute_grad` blurs its input."""
    dx, dy = _abs_fwd_diff(blur(x))
    return _abs_fwd_diff(blur(dx))[0], _abs_fwd_diff(blur(dy))[1]


def smooth_reg(disp, img, use_edges=False, use_laplacian=False, use_blur=False):
    """`SmoothReg.forward` (src/regularizers/smooth.py:71-97); `use_laplacian`: second-order differences; `use_blur`: `compute_grad` blurs its
    input first (:21; see `gaussian_blur3x3`)."""
    blur = gaussian_blur3x3 if use_blur else (lambda t: t)
    fn = (lambda t: _laplacian(t, blur)) if use_laplacian else (lambda t: _abs_fwd_diff(blur(t)))
    d = disp/disp.mean(dim=(2, 3), keepdim=True).clamp(min=EPS32)        # ops.mean_normalize, src/tools/ops.py:279-286
    ddx, ddy = fn(d)
    disp_grad = (ddx.pow(2) + ddy.pow(2)).clamp(min=EPS32).sqrt()        # :86
    idx_, idy_ = fn(img)
    idx_, idy_ = idx_.mean(dim=1, keepdim=True), idy_.mean(dim=1, keepdim=True)
    img_grad = (idx_.pow(2) + idy_.pow(2)).clamp(min=EPS32).sqrt()       # :89
    if use_edges: ddx, ddy = ddx*(-idx_).exp(), ddy*(-idy_).exp()        # :91-94
    return ddx.mean() + ddy.mean(), {'disp_grad': disp_grad, 'image_grad': img_grad}


# ---------------------------------------------------------------------------------------------------
# a5/a11: handlers, a13: combination
# ---------------------------------------------------------------------------------------------------
def image_recon(depths: dict, imgs, supp_imgs, Ts, Ks, loss_name='ssim', use_min=False, use_automask=False,
                noise=None, aten=False, force_sel=None):
    """`handlers.image_recon` (src/core/handlers.py:14-67).  depths {s: (b,1,h,w)}, imgs (b,3,h,w),
    supp_imgs (n,b,3,h,w), Ts (n,b,4,4), Ks (b,4,4).  Flattened batch order is n-major, then scale, then b."""
    n, S = supp_imgs.shape[0], len(depths)
    b = imgs.shape[0]
    dep = torch.stack(list(depths.values())).flatten(0, 1)                 # (S*b,1,h,w)
    tgt = imgs[None].expand(S, *imgs.shape).flatten(0, 1)                  # (S*b,3,h,w)
    src = supp_imgs[:, None].expand(n, S, *supp_imgs.shape[1:]).flatten(1, 2)  # (n,S*b,3,h,w)
    T = Ts[:, None].expand(n, S, b, 4, 4).flatten(0, 2)                    # (n*S*b,4,4)
    K = Ks[None, None].expand(n, S, b, 4, 4).flatten(0, 2)
    warp = view_synth(src.flatten(0, 1), dep[None].expand(n, *dep.shape).flatten(0, 1), T, K, aten=aten)[0]
    warp = warp.unflatten(0, (n, S*b))
    loss, out = recon_loss(warp, tgt, source=src, loss_name=loss_name, use_min=use_min, use_automask=use_automask,
                           noise=noise, aten=aten, force_sel=force_sel)
    ld = {'supp_imgs_warp': warp.unflatten(1, (S, b))[:, 0]}
    if use_automask: ld['automask'] = out['automask'].unflatten(0, (S, b))[0]
    full = {'warp': warp, 'err': out['err'].unflatten(0, (S, b)), 'err_warp': out['err_warp'].unflatten(0, (S, b)),
            'sel': out['sel'].unflatten(0, (S, b))}
    if 'tie_gap' in out: full['tie_gap'] = out['tie_gap'].unflatten(0, (S, b))
    return loss, ld, full


def disp_smooth(disps: dict, imgs, use_edges=False, aten=False):
    """`handlers.disp_smooth` (src/core/handlers.py:262-281): mean over scales of loss_s / 2**s; aux of scale 0."""
    ls = {s: smooth_reg(d, resize_bilinear(imgs, d.shape[-2:], aten=aten), use_edges) for s, d in disps.items()}
    loss = torch.stack([v[0]/2**s for s, v in ls.items()]).mean()
    return loss, ls[min(ls)][1] if 0 not in ls else ls[0][1]


def loss_path(disps: dict, imgs, supp_imgs, Ts, Ks, *, min_depth=0.1, max_depth=100, loss_name='ssim', use_min=True,
              use_automask=True, use_edges=True, w_recon=1.0, w_smooth=0.001, noise=None, aten=False, force_sel=None):
    """The whole hot path as one call: forward_postprocess (src/core/trainer.py:316-321) + forward_loss for the
    keys `img_recon` and `disp_smooth` (:388-392, :436-437) + weighted sum (:462-464).
    `w_smooth=None` drops the regulariser."""
    _, depth_up = disp_to_depth_up(disps, imgs.shape[-2:], min_depth, max_depth, aten=aten)
    l_rec, ld, full = image_recon(depth_up, imgs, supp_imgs, Ts, Ks, loss_name, use_min, use_automask, noise, aten, force_sel)
    loss = w_recon*l_rec
    out = {'loss_img_recon': l_rec, **ld, 'depth_up': depth_up, 'full': full}
    if w_smooth is not None:
        l_sm, ld_sm = disp_smooth(disps, imgs, use_edges, aten=aten)
        loss = loss + w_smooth*l_sm
        out.update(loss_disp_smooth=l_sm, **ld_sm)
    return loss, out


# ---------------------------------------------------------------------------------------------------
# §8f rank 3: the other ViewSynth users — regression loss, feature / autoencoder reconstruction,
# virtual-stereo consistency, proxy-depth regression with the Depth-Hints automask
# ---------------------------------------------------------------------------------------------------
def regression_error(pred, target, loss_name='berhu'):
    """Dense regression errors (src/losses/regression.py:11-37).  'berhu' uses the dynamic threshold
    delta = 0.2*max|pred - target| over the WHOLE tensor (:32-33), and autograd differentiates through that max."""
    diff = (pred - target).abs()
    if loss_name == 'l1': return diff
    if loss_name == 'log_l1': return (1 + diff).log()
    if loss_name == 'berhu':
        delta = 0.2*diff.max()
        return torch.where(diff <= delta, diff, (diff.pow(2) + delta.pow(2))/(2*delta + EPS32))
    raise KeyError(loss_name)


def regression_loss(pred, target, mask=None, loss_name='berhu', invert=False):
    """`RegressionLoss.forward` (src/losses/regression.py:69-75): masked mean of the dense error; `invert` maps both
    inputs through `to_inv` first.  Returns loss, dict(err_regr, mask_regr)."""
    if invert: pred, target = to_inv(pred), to_inv(target)
    if mask is None: mask = torch.ones_like(target)
    err = mask*regression_error(pred, target, loss_name)
    return err.sum()/mask.sum(), {'err_regr': err, 'mask_regr': mask}


def feat_recon(depths: dict, feats, supp_feats, Ts, Ks, loss_name='l2', use_min=True, use_automask=True, noise=None, aten=False):
    """`handlers.feat_recon` (src/core/handlers.py:70-119) for tensor inputs: features are detached, bilinearly resized
    to the depth map, and scale 0 alone goes through `image_recon`.  feats (b,c,hf,wf), supp_feats (n,b,c,hf,wf)."""
    size = depths[0].shape[-2:]
    feats, supp_feats = feats.detach(), supp_feats.detach()
    n = supp_feats.shape[0]
    feats = resize_bilinear(feats, size, aten=aten)
    supp_feats = resize_bilinear(supp_feats.flatten(0, 1), size, aten=aten).unflatten(0, (n, -1))
    loss, ld, full = image_recon({0: depths[0]}, feats, supp_feats, Ts, Ks, loss_name, use_min, use_automask, noise, aten)
    return loss, {'supp_feats_warp': ld['supp_imgs_warp']}, full


def autoenc_recon(preds: dict, targets, supp_preds: dict, supp_targets, loss_name='ssim', use_min=False, aten=False):
    """`handlers.autoenc_recon` (src/core/handlers.py:122-149): every scale's autoencoder output against the input image,
    target and support frames in one batch.  preds {s: (b,3,h,w)}, supp_preds {s: (n,b,3,h,w)}, supp_targets (n,b,3,h,w)."""
    S = len(preds)
    p = torch.stack(list(preds.values())).flatten(0, 1)
    sp = torch.stack(list(supp_preds.values())).flatten(0, 2)
    t = targets[None].expand(S, *targets.shape).flatten(0, 1)
    st = supp_targets[None].expand(S, *supp_targets.shape).flatten(0, 2)
    loss, _ = recon_loss(torch.cat((p, sp)), torch.cat((t, st)), loss_name=loss_name, use_min=use_min, aten=aten)
    return loss


def stereo_const(disps: dict, depths: dict, disps_stereo: dict, depths_stereo: dict, T_stereo, K, loss_name='l1',
                 invert=False, aten=False):
    """`handlers.stereo_const` (src/core/handlers.py:152-198): warp the virtual-stereo disparity into the target view with
    the target depth (and vice versa with the inverse transform) and regress it on the un-warped disparity."""
    S = len(disps)
    d = torch.stack(list(disps.values())).flatten(0, 1); dep = torch.stack(list(depths.values())).flatten(0, 1)
    ds = torch.stack(list(disps_stereo.values())).flatten(0, 1); deps = torch.stack(list(depths_stereo.values())).flatten(0, 1)
    T = T_stereo[None].expand(S, *T_stereo.shape).flatten(0, 1)
    Kx = K[None, None].expand(2, S, *K.shape).flatten(0, 2)
    all_disps = torch.cat((ds, d))
    warp = view_synth(all_disps, torch.cat((dep, deps)), torch.cat((T, torch.linalg.inv(T))), Kx, aten=aten)[0]
    loss, _ = regression_loss(all_disps, warp, None, loss_name, invert)
    sw, dw = warp.chunk(2)
    return loss, {'disps_warp': dw.unflatten(0, (S, -1))[0], 'stereo_disps_warp': sw.unflatten(0, (S, -1))[0]}


def depth_regr(depths: dict, targets, imgs, supp_imgs, Ts, Ks, loss_name='berhu', invert=False, use_automask=False,
               photo_loss_name='ssim', photo_use_min=True, aten=False):
    """`handlers.depth_regr` (src/core/handlers.py:201-259): regress every scale's depth on the proxy depth where it is
    valid; with `use_automask` only where the proxy depth reconstructs the target better than the prediction does
    (`photo` is `ReconstructionLoss.compute_photo` of the img_recon criterion, src/core/trainer.py:430)."""
    S, n = len(depths), supp_imgs.shape[0]
    b = imgs.shape[0]
    im = imgs[None].expand(S, *imgs.shape).flatten(0, 1)
    dep = torch.stack(list(depths.values())).flatten(0, 1)
    tg = targets[None].expand(S, *targets.shape).flatten(0, 1)
    masks = tg > 0
    ld = {}
    if use_automask:
        src = supp_imgs[:, None].expand(n, S, *supp_imgs.shape[1:]).flatten(1, 2)                 # (n,S*b,3,h,w)
        T = Ts[:, None].expand(n, S, b, 4, 4).flatten(0, 2); K = Ks[None, None].expand(n, S, b, 4, 4).flatten(0, 2)
        ex = lambda z: z[None].expand(n, *z.shape).flatten(0, 1)
        hints_warp = view_synth(src.flatten(0, 1), ex(tg), T, K, aten=aten)[0].unflatten(0, (n, -1))
        pred_warp = view_synth(src.flatten(0, 1), ex(dep), T, K, aten=aten)[0].unflatten(0, (n, -1))
        e_pred = compute_photo(pred_warp, im, photo_loss_name, photo_use_min, aten)[0]
        e_hint = compute_photo(hints_warp, im, photo_loss_name, photo_use_min, aten)[0]
        automask = e_pred > e_hint
        ld['automask_hints'] = automask.unflatten(0, (S, -1))[0]
        masks = masks & automask
    loss, out = regression_loss(dep, tg, masks, loss_name, invert)
    ld['mask_regr'] = out['mask_regr'].unflatten(0, (S, -1))[0]
    return loss, ld
