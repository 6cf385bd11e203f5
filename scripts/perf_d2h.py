"""Download rate of a 4 GB result plane: pageable .cpu() against a pinned destination."""
import time, torch
x = torch.randn(50000, 10000, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter(); a = x.cpu(); t1 = time.perf_counter()
    print(f"pageable .cpu(): {t1 - t0:.3f} s ({x.numel() * 8 / (t1 - t0) / 1e9:.1f} GB/s)", flush=True)
    del a
    t0 = time.perf_counter()
    h = torch.empty(x.shape, dtype=x.dtype, pin_memory=True)
    t1 = time.perf_counter()
    h.copy_(x); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"pinned: alloc {t1 - t0:.3f} s, copy {t2 - t1:.3f} s ({x.numel() * 8 / (t2 - t1) / 1e9:.1f} GB/s)", flush=True)
    del h
