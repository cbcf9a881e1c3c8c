import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch
from conftest import load_golden, case_inputs
from oracle import view_synth_oracle as O
from slowtv_monodepth_amd import functional as F
g = load_golden('train_l1_automask_24x32')
leaves, static = case_inputs(g, requires_grad=False)
scales = static['scales']; h,w = static['imgs'].shape[-2:]
_, depth_up = O.disp_to_depth_up({s: leaves[f'disp_{s}'] for s in scales}, (h,w), 0.1, 100)
dep_c = {s: d.clone().requires_grad_(True) for s,d in depth_up.items()}
Ts = g['out_Ts']; K = g['in_K']
l, ld, full = O.image_recon(dep_c, static['imgs'], static['supp_imgs'], Ts, K, 'l1', True, True, static['noise'])
l.backward()
gd_c = torch.stack([dep_c[s].grad for s in scales])  # S,b,1,h,w
dg = torch.stack([depth_up[s] for s in scales]).cuda().requires_grad_(True)
flags = F.recon_flags('l1', True, True)
lg, err, sel, _ = F.image_recon_fused(dg, static['imgs'].cuda(), static['supp_imgs'].cuda(), Ts.cuda(), K.cuda(), flags=flags, noise=static['noise'].cuda())
lg.backward()
gd_g = dg.grad.cpu()
diff = (gd_g - gd_c).abs()
print('max diff', diff.max().item(), 'max ref', gd_c.abs().max().item())
idx = diff.flatten().topk(8).indices
S,b = len(scales), static['imgs'].shape[0]
sx, sy, z, grid = None, None, None, None
for i in idx.tolist():
    s_, r = divmod(i, b*h*w); b_, r = divmod(r, h*w); v,u = divmod(r, w)
    print(f's={s_} b={b_} v={v} u={u} hip={gd_g.flatten()[i].item():.6e} ref={gd_c.flatten()[i].item():.6e} sel={sel[s_,b_,0,v,u].item()}')
    for i_s in range(2):
        sxx, syy, zz, _ = O.sample_coords(depth_up[scales[s_]], Ts[i_s], K)
        print(f'   supp{i_s}: sx={sxx[b_,v,u].item():.6f} sy={syy[b_,v,u].item():.6f} z={zz[b_,0,v,u].item():.4f}')
