"""Ablation of the lattice deep kernel (AFD_WPT4_DBG switches): per-kernel time from HIP events via the library's class timing."""
import os, sys, torch
sys.path.insert(0, "audiodeepfake-detection_amd")
from audiofakedetect.wavelet_math import Packets
name = sys.argv[1] if len(sys.argv) > 1 else "coif4"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
x = (0.1 * torch.randn(B, 22050, device="cuda")).clamp_(-1, 1)
for dbg in (0, 1, 16, 17, 2, 4, 6, 8, 14, 15):
    os.environ["AFD_WPT4_DBG"] = str(dbg)
    p = Packets(name, max_lev=14, log_scale=True)
    for _ in range(3):
        out, _ = p(x)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out, _ = p(x)
    e1.record(); torch.cuda.synchronize()
    print(f"{name} B={B} dbg={dbg:2d}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us (top + deep)", flush=True)
