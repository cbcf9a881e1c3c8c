import torch, torch.nn.functional as F, time
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
for dt in (torch.float32, torch.bfloat16):
    for (N, C, H, W) in ((12, 128, 96, 160), (12, 256, 48, 80), (12, 512, 24, 40), (12, 1024, 12, 20), (12, 96, 48, 160)):
        x_cl = torch.randn(N, H, W, C, device='cuda', dtype=dt, requires_grad=True)
        x_cf = torch.randn(N, C, H, W, device='cuda', dtype=dt, requires_grad=True)
        w1 = torch.randn(4*C, C, device='cuda', dtype=dt, requires_grad=True); w2 = torch.randn(C, 4*C, device='cuda', dtype=dt, requires_grad=True)
        def lin():
            y = F.linear(F.gelu(F.linear(x_cl, w1)), w2); y.sum().backward()
        def cv():
            y = F.conv2d(F.gelu(F.conv2d(x_cf, w1[:, :, None, None])), w2[:, :, None, None]); y.sum().backward()
        print(f'{str(dt)[6:]:9s} C={C:4d} {H}x{W}: linear mlp fwd+bwd {timeit(lin):7.3f} ms | conv1x1 mlp {timeit(cv):7.3f} ms')
