import sys, torch
sys.path.insert(0, "audiodeepfake-detection_amd")
from audiofakedetect.wavelet_math import Packets
def run(name, level, B, iters=20):
    x = (0.1 * torch.randn(B, 22050, device="cuda")).clamp_(-1, 1)
    p = Packets(name, max_lev=level, log_scale=True)
    for _ in range(3): out, _ = p(x)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): out, _ = p(x)
    e1.record(); torch.cuda.synchronize()
    print(f"{name} L{level} B={B}: {e0.elapsed_time(e1)/iters*1e3:.1f} us", flush=True)
for lv in (3, 6, 8, 11, 12, 13, 14):
    run("coif4", lv, 128)
run("sym5", 8, 128); run("sym5", 14, 128); run("haar", 8, 4096); run("haar", 14, 4096)
