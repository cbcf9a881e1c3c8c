import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, 'tests')
import torch
from conftest import load_golden
import slowtv_monodepth_amd as amd
from slowtv_monodepth_amd import functional as F
g = load_golden('op_recon_mask_uncer_min1_auto1_c3')
crit = amd.losses.ReconstructionLoss('ssim', True, True, 'uncertainty')
pred, mask = g['in_pred'].cuda().requires_grad_(True), g['in_mask'].cuda().requires_grad_(True)
loss, ld = crit(pred, g['in_target'].cuda(), source=g['in_source'].cuda(), mask=mask, noise=g['in_noise'].cuda())
loss.backward()
d = (mask.grad.cpu() - g['grad_mask']).abs()
bad = (d > 1e-5).nonzero()
print('bad entries', len(bad))
n, b = g['in_pred'].shape[:2]
tg = g['in_target'].cuda()[None].expand(n, *g['in_target'].shape).flatten(0, 1)
ew = F.photo_error(pred.detach().flatten(0, 1), tg).view(n, b, *pred.shape[-2:])
es = F.photo_error(g['in_source'].cuda().flatten(0, 1), tg).view(n, b, *pred.shape[-2:])
_, _, sel = F.recon_reduce(ew, es, use_min=True, noise=g['in_noise'].cuda(), mask=mask.detach(), mask_name='uncertainty')
for bi, ch, v, u in bad[:12].tolist():
    m = mask[bi, :, v, u].detach().cpu(); e_w = ew[:, bi, v, u].cpu(); e_s = es[:, bi, v, u].cpu()
    print((bi, ch, v, u), 'sel', sel[bi, v, u].item(), 'hip', mask.grad[bi, :, v, u].cpu().tolist(), 'ref', g['grad_mask'][bi, :, v, u].tolist(),
          'mw', (e_w*torch.exp(-m) + m).tolist(), 'ms', (e_s*torch.exp(-m) + m).tolist())
