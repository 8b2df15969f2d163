"""Does the HOST of this box stall periodically while it launches kernels?  Plain torch launches (no code of ours), host time per
launch; prints every launch that took > 2 ms and when (measurement tool)."""
import time, torch
x = torch.zeros(1024, device="cuda:0")
torch.cuda.synchronize()
t_start = time.perf_counter(); last = t_start; stalls = []; n = 0
while time.perf_counter() - t_start < 3.0:
    x.add_(1.0)
    n += 1
    if n % 64 == 0:
        torch.cuda.current_stream().synchronize() if n % 4096 == 0 else None
    now = time.perf_counter()
    if now - last > 2e-3:
        stalls.append((round((now - t_start) * 1e3, 1), round((now - last) * 1e3, 1)))
    last = now
print("launches", n, "avg us per launch", 3.0 / n * 1e6)
print("stalls > 2 ms (at ms, lasted ms):", stalls[:40])
