"""Front-end time (top + deep, log epilogue) of the lattice deep kernel by workgroup size (AFD_D4_GROUP), and of the
matrix-core composite (AFD_WPT_DEEP_MFMA=1)."""
import os, sys, torch
sys.path.insert(0, "audiodeepfake-detection_amd")
from audiofakedetect.wavelet_math import Packets
def t(name, B, iters=20):
    x = (0.1 * torch.randn(B, 22050, device="cuda")).clamp_(-1, 1)
    p = Packets(name, max_lev=14, log_scale=True)
    for _ in range(3): out, _ = p(x)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): out, _ = p(x)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("coif4", "sym5")):
    for B in (128, 4096):
        row = []
        for g in ("4", "8", "16"):
            os.environ["AFD_D4_GROUP"] = g
            row.append(f"grp{g} {t(name, B):.1f}")
        os.environ["AFD_WPT_DEEP_MFMA"] = "1"
        row.append(f"mfma {t(name, B):.1f}")
        os.environ.pop("AFD_WPT_DEEP_MFMA")
        print(f"{name} B={B}: " + "  ".join(row) + " us", flush=True)
