"""Front-end timing (development): ms per transform of Packets(name, level) at batch B, HIP events over 20 launches.
    python3 tools/fe_time.py sym5:14:4096 sym5:8:4096 coif4:14:128 ..."""
import sys, torch
sys.path.insert(0, "audiodeepfake-detection_amd")
from audiofakedetect.wavelet_math import Packets
from audiofakedetect import wavelets
for spec in sys.argv[1:]:
    name, level, B = spec.split(":")
    level, B = int(level), int(B)
    x = (0.1 * torch.randn(B, 22050, device="cuda")).clamp_(-1, 1)
    p = Packets(name, max_lev=level, log_scale=True)
    for _ in range(5): out, _ = p(x)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): p(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    nbytes = 4.0 * B * (22050 + out.numel() / B)
    print("%-8s level %2d B=%5d: %.4f ms  %.0f GB/s  %.3f of 8 TB/s" % (name, level, B, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 8000), flush=True)
