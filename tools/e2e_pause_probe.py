"""The prefetch thread's pause after a hand-over (NativeFrameLoader.WORKER_PAUSE) and the in-thread loader's reader count,
on the headline step, interleaved with the resident batch (development).
    python3 tools/e2e_pause_probe.py [steps] [rounds]"""
import os, sys, time, tempfile, shutil, wave
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "audiodeepfake-detection_amd"))
import bench
from audiofakedetect.data_loader import NativeFrameLoader, get_costum_dataset

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
torch.manual_seed(0)
args, trainer, transforms = bench.build("coif4-l14", 128, False, dev)
trainer.model.train()
root = tempfile.mkdtemp(prefix="afd_probe_")
rng = np.random.default_rng(1234)
for name in ("A_real", "B_fake"):
    os.makedirs(os.path.join(root, name))
    for i in range(40):
        pcm = np.clip(rng.standard_normal(22050 * 40) * 3276.8, -32768, 32767).astype(np.int16)
        with wave.open(os.path.join(root, name, f"{i:04d}.wav"), "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(22050); f.writeframes(pcm.tobytes())
ds = get_costum_dataset(data_path=root, save_path=os.path.join(root, "index"), ds_type="train", seconds=1, resample_rate=22050, limit=-1)

def run(batches, n, warm):
    it = iter(batches)
    for _ in range(warm):
        trainer._run_batch(0, next(it))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        trainer._run_batch(0, next(it))
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

def cycle(loader):
    e = 0
    while True:
        loader.set_epoch(e); e += 1
        for b in loader:
            yield b

res = bench.synthetic_batch(128, 0, dev)
def auto(frac, cap):
    l = NativeFrameLoader(ds, 128, dev, shuffle=True, seed=0, threads=16)
    l.WORKER_PAUSE, l.WORKER_PAUSE_CAP_MS = frac, cap
    return l
for r in range(rounds):
    print("resident batch                       %.3f ms/step" % run(iter(lambda: res, None), steps, 3), flush=True)
    for frac, cap in ((0.0, 0.0), (0.2, 10.0), (0.5, 30.0), (0.8, 40.0)):
        print("auto, pause %.1f of the step (cap %2.0f ms) %.3f ms/step" % (frac, cap, run(cycle(auto(frac, cap)), steps, 8)), flush=True)
    for th in (2, 4, 8):
        print("caller's thread, %2d reader threads     %.3f ms/step" % (th, run(cycle(NativeFrameLoader(ds, 128, dev, shuffle=True, seed=0, prefetch=0, threads=th)), steps, 3)), flush=True)
print("resident batch                       %.3f ms/step" % run(iter(lambda: res, None), steps, 3))
shutil.rmtree(root, ignore_errors=True)
