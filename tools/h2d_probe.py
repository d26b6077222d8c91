import torch, time
x = torch.empty(32, 6, 3, 128, 128).pin_memory()
d = torch.empty_like(x, device='cuda')
for n in (1, 4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10 * n):
        d.copy_(x, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (10 * n)
    print(f'H2D {x.numel() * 4 / 1e6:.1f} MB: {1e3 * dt:.2f} ms = {x.numel() * 4 / dt / 1e9:.1f} GB/s')
y = torch.empty(128, 56, 7, 128).pin_memory()
e = torch.empty_like(y, device='cuda')
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): y.copy_(e, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f'D2H {y.numel() * 4 / 1e6:.1f} MB: {1e3 * dt:.2f} ms = {y.numel() * 4 / dt / 1e9:.1f} GB/s')
