import os, sys
sys.path.insert(0, '/root/repo')
import torch
from slowtv_monodepth_amd import functional as F
from slowtv_monodepth_amd.synthetic import make_batch
b, h, w, supp = 12, 192, 640, (-1, 1)
_, y, _ = make_batch(b, h, w, supp, seed=42, device='cuda')
g = torch.Generator(device='cuda').manual_seed(0)
import torch.nn.functional as Fn
for mode in ('smooth', 'rough'):
    def mk(s):
        hs, ws = h >> s, w >> s
        if mode == 'rough': return 0.05 + 0.9*torch.rand(b, 1, hs, ws, device='cuda', generator=g)
        low = 0.2 + 0.6*torch.rand(b, 1, 4, 10, device='cuda', generator=g)
        return Fn.interpolate(low, size=(hs, ws), mode='bilinear', align_corners=False) + 0.01*torch.rand(b, 1, hs, ws, device='cuda', generator=g)
    d0 = [mk(s) for s in range(4)]
    T0 = torch.eye(4, device='cuda').repeat(2, b, 1, 1); T0[..., :3, 3] = 0.05*torch.randn(2, b, 3, device='cuda', generator=g)
    flags = F.recon_flags('ssim', True, True)
    res = []
    for skip in ('2', '0'):
        os.environ['SMD_BWD_SKIP'] = skip
        d = [v.clone().requires_grad_(True) for v in d0]; T = T0.clone().requires_grad_(True)
        loss, err, sel, _, dep = F.image_recon_fused_disp(d, y['imgs'], y['supp_imgs'], T, y['K'], flags=flags, min_depth=0.1, max_depth=100, seed=1, want_err=False)
        loss.backward(); torch.cuda.synchronize()
        res.append([v.grad for v in d] + [T.grad])
    sel0 = sel.reshape(4, b, h, w)[0]
    live = (sel0 == 0)
    pad = (-w) % 60
    lv = Fn.pad(live, (0, pad)).view(b, h, -1, 60).any(-1)
    print(mode, 'dead tile fraction (support 0, scale 0):', 1 - lv.float().mean().item(), '| grads bit-identical:', [bool(torch.equal(x, z)) for x, z in zip(*res)],
          'max rel diff', max(float((x - z).abs().max()/z.abs().max()) for x, z in zip(*res)))
