import sys, time, torch
sys.path.insert(0,'audiodeepfake-detection_amd')
from audiofakedetect.wavelet_math import STFTLayer
from audiofakedetect import _native
_native.load()
for B in (128, 1024):
    x = (0.1*torch.randn(B,1,22050)).cuda()
    st = STFTLayer(511, 220, log_scale=True)
    for _ in range(5): st(x)
    torch.cuda.synchronize(); 
    ev=[torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(50): st(x)
    ev[1].record(); torch.cuda.synchronize()
    print("stft B=%d: %.4f ms" % (B, ev[0].elapsed_time(ev[1])/50))
