import os, sys, torch, torch.nn.functional as TF
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from slowtv_monodepth_amd import functional as F
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
for (B, C, h, w) in [(12, 16, 384, 640), (12, 32, 192, 320), (12, 64, 96, 160), (12, 128, 48, 80), (12, 16, 192, 640)]:
    xp = torch.randn(B, C, h + 2, w + 2, device='cuda'); wt = torch.randn(1, C, 3, 3, device='cuda')/12; bs = torch.zeros(1, device='cuda'); gy = torch.randn(B, 1, h, w, device='cuda')
    res = {}
    for nm, x in (('fp32', xp), ('bf16', xp.bfloat16())):
        L = [x.clone().requires_grad_(True), wt.clone().requires_grad_(True), bs.clone().requires_grad_(True)]
        f = timeit(lambda: F.conv3x3_head(L[0], L[1], L[2], 'sigmoid'))
        def fb():
            for t in L: t.grad = None
            F.conv3x3_head(L[0], L[1], L[2], 'sigmoid').backward(gy)
        res[nm] = (f, timeit(fb) - f)
    xb = xp.bfloat16().requires_grad_(True); wb = wt.clone().requires_grad_(True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        f = timeit(lambda: torch.sigmoid(TF.conv2d(xb, wb, bs)))
        def fb():
            xb.grad = None; wb.grad = None
            torch.sigmoid(TF.conv2d(xb, wb, bs)).float().backward(gy)
        res['miopen bf16'] = (f, timeit(fb) - f)
    print(f'{C}->1 {h}x{w}: ' + '; '.join(f'{k}: fwd {v[0]:.1f} bwd {v[1]:.1f} us' for k, v in res.items()), flush=True)
