"""H2D bandwidth of a 189 MB float32 batch out of: pageable memory, torch's pinned allocator (hipHostMalloc), a hipHostRegister'ed
private allocation, a hipHostRegister'ed shared-memory tensor (what hipmonocon.feed.RingLoader uploads from)."""
import time
import torch

n = 32 * 3 * 384 * 1280
dev = torch.empty(n, device="cuda")
rt = torch.cuda.cudart()


def bw(src, tag, flags=None):
    for _ in range(2):
        dev.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        dev.copy_(src, non_blocking=True)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("%-44s %6.1f ms per copy = %5.1f GB/s (issue %.1f ms each, is_pinned %s)"
          % (tag, dt * 1e3, n * 4 / dt / 1e9, t_issue / 5 * 1e3, src.is_pinned()), flush=True)


bw(torch.randn(n), "pageable")
bw(torch.randn(n).pin_memory(), "torch pinned (hipHostMalloc)")
for flags in (0, 1, 2, 3):
    a = torch.randn(n)
    rc = rt.cudaHostRegister(a.data_ptr(), n * 4, flags)
    bw(a, "malloc + hipHostRegister(flags=%d) rc=%s" % (flags, int(rc)))
    rt.cudaHostUnregister(a.data_ptr())
for flags in (0, 1, 2, 3):
    s = torch.randn(n).share_memory_()
    rc = rt.cudaHostRegister(s.data_ptr(), n * 4, flags)
    bw(s, "shared memory + hipHostRegister(flags=%d) rc=%s" % (flags, int(rc)))
    rt.cudaHostUnregister(s.data_ptr())
big = torch.empty(31, n).share_memory_()
big.zero_()
t0 = time.perf_counter()
rc = rt.cudaHostRegister(big.data_ptr(), big.numel() * 4, 0)
print("register 5.9 GB: %.2f s rc=%s" % (time.perf_counter() - t0, int(rc)))
bw(big[7], "slot of a 5.9 GB registered shared ring")
