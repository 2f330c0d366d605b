"""Where the level-1..8 kernel's time goes (development helper): builds of csrc/wpt3.hip with -DAFD_TOP_STOP=k return after
the frame load (k = 0) or after level k; each runs the level-8 transform of B frames in its own process (AFD_LIB).

    python3 tools/top_stages.py [B]        (libraries audiodeepfake-detection_amd/lib_stop{0,1,3,5,7} and lib)
"""
import os, subprocess, sys
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import sys, torch
sys.path.insert(0, "audiodeepfake-detection_amd")
from audiofakedetect.wavelet_math import Packets
B = int(sys.argv[1])
for name in ("coif4", "sym5"):
    x = (0.1 * torch.randn(B, 22050, device="cuda")).clamp_(-1, 1)
    p = Packets(name, max_lev=8, log_scale=True)
    for _ in range(5): p(x)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): p(x)
    e1.record(); torch.cuda.synchronize()
    print(name, "%.1f" % (e0.elapsed_time(e1) / 20 * 1e3), end="  ")
print()
'''
for tag in ("lib_stop0", "lib_stop1", "lib_stop3", "lib_stop5", "lib_stop7", "lib"):
    env = dict(os.environ, AFD_LIB=os.path.join(root, "audiodeepfake-detection_amd", tag, "libafd_hip.so"))
    out = subprocess.run([sys.executable, "-c", child, str(B)], env=env, cwd=root, capture_output=True, text=True)
    print(f"{tag:10s} us per transform (B = {B}): {out.stdout.strip() or out.stderr[-300:]}", flush=True)
