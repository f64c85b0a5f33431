import torch, time
x = torch.randn(32, 3, 384, 1280).pin_memory()
y = torch.empty_like(x, device="cuda")
for _ in range(3): y.copy_(x, non_blocking=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): y.copy_(x, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("H2D pinned %.1f MB in %.2f ms = %.1f GB/s" % (x.numel() * 4 / 1e6, dt * 1e3, x.numel() * 4 / dt / 1e9))
u8 = (torch.rand(32, 384, 1280, 3) * 255).to(torch.uint8).pin_memory()
z = torch.empty_like(u8, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): z.copy_(u8, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("H2D pinned uint8 HWC %.1f MB in %.2f ms" % (u8.numel() / 1e6, dt * 1e3))
