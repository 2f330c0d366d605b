"""Front-end A/B: level-14 transform time with the lattice deep kernel (wpt4.hip) and with the matrix-core composite
(AFD_WPT_DEEP_MFMA=1), same process, same inputs; also the largest difference between the two outputs."""
import os, sys, torch
sys.path.insert(0, "audiodeepfake-detection_amd")
from audiofakedetect.wavelet_math import Packets

def run(name, B, iters=20):
    x = (0.1 * torch.randn(B, 22050, device="cuda")).clamp_(-1, 1)
    res = {}
    outs = {}
    for mode in ("lattice", "mfma"):
        if mode == "mfma":
            os.environ["AFD_WPT_DEEP_MFMA"] = "1"
        else:
            os.environ.pop("AFD_WPT_DEEP_MFMA", None)
        p = Packets(name, max_lev=14, log_scale=False)
        for _ in range(3):
            out, _ = p(x)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            out, _ = p(x)
        e1.record(); torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / iters * 1e3
        outs[mode] = out[: min(B, 16)].clone()
    os.environ.pop("AFD_WPT_DEEP_MFMA", None)
    d = (outs["lattice"] - outs["mfma"]).abs().max().item() / outs["mfma"].abs().max().item()
    print(f"{name} B={B}: lattice {res['lattice']:.1f} us, mfma {res['mfma']:.1f} us, max diff {d:.2e} of max", flush=True)

for name in sys.argv[1].split(",") if len(sys.argv) > 1 else ("coif4", "sym5", "db8"):
    for B in (128, 4096):
        run(name, B)
