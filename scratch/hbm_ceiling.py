"""Achievable HBM bandwidth on this box for the access patterns of the elementwise passes (torch's own kernels as the yard-stick)."""
import torch, time
def bench(fn, bytes_, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    return bytes_ / dt / 1e12, dt * 1e6
for mb in (64, 256, 1024):
    n = mb * 1024 * 1024 // 4
    a = torch.randn(n, device="cuda"); b = torch.randn(n, device="cuda"); c = torch.empty_like(a)
    print("size %5d MB  copy  %.2f TB/s (%.0f us)" % ((mb,) + bench(lambda: c.copy_(a), 2 * n * 4)))
    print("size %5d MB  triad %.2f TB/s (%.0f us)" % ((mb,) + bench(lambda: torch.add(a, b, out=c), 3 * n * 4)))
    print("size %5d MB  read  %.2f TB/s (%.0f us)" % ((mb,) + bench(lambda: a.sum(), n * 4)))
    print("size %5d MB  fill  %.2f TB/s (%.0f us)" % ((mb,) + bench(lambda: c.zero_(), n * 4)))
