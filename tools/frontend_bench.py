"""Quick front-end timing (development helper; bench.py is the contract benchmark)."""
import sys, time
sys.path.insert(0, "audiodeepfake-detection_amd")
import torch
from audiofakedetect.wavelet_math import Packets

def run(name, level, B, iters=20):
    x = (0.1 * torch.randn(B, 22050, device="cuda")).clamp_(-1, 1)
    p = Packets(name, max_lev=level, log_scale=True)
    for _ in range(3):
        out, _ = p(x)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out, _ = p(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    by = 4 * (22050 + out.numel() // B) * B
    print(f"{name} L{level} B={B}: {ms*1e3:.1f} us  {B/ms*1e3:.0f} frames/s  {by/ms/1e6:.1f} GB/s ({by/ms/1e6/8000*100:.1f}% of 8 TB/s)", flush=True)

for cfg in [("haar", 14, 4096), ("haar", 8, 4096), ("sym5", 8, 128), ("sym5", 14, 128), ("coif4", 8, 128), ("coif4", 14, 128), ("coif4", 14, 1024)]:
    run(*cfg)
