"""Front-end time by decomposition level (development helper)."""
import sys, torch
sys.path.insert(0, "audiodeepfake-detection_amd")
from audiofakedetect.wavelet_math import Packets
def run(name, level, B, iters=30, log_scale=True):
    x = (0.1 * torch.randn(B, 22050, device="cuda")).clamp_(-1, 1)
    p = Packets(name, max_lev=level, log_scale=log_scale)
    for _ in range(5): out, _ = p(x)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): out, _ = p(x)
    e1.record(); torch.cuda.synchronize()
    print(f"{name} L{level} B={B}: {e0.elapsed_time(e1)/iters*1e3:.1f} us", flush=True)
levels = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (3, 6, 8, 11, 12, 13, 14)
for lv in levels:
    run(sys.argv[1] if len(sys.argv) > 1 else "coif4", lv, int(sys.argv[3]) if len(sys.argv) > 3 else 128)
