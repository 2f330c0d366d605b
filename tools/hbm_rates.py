"""Achievable HBM rates on the box (development helper): write-only (fill), read-only (sum), copy."""
import torch
n = 1 << 30  # 4 GiB of float32
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
gb = n * 4 / 1e9
t = timed(lambda: x.fill_(1.0)); print(f"fill  (write only): {gb / t / 1e3:.2f} TB/s")
t = timed(lambda: x.sum());      print(f"sum   (read only) : {gb / t / 1e3:.2f} TB/s")
t = timed(lambda: y.copy_(x));   print(f"copy  (read+write): {2 * gb / t / 1e3:.2f} TB/s total")
t = timed(lambda: torch.add(x, 1.0, out=y)); print(f"add   (read+write): {2 * gb / t / 1e3:.2f} TB/s total")
