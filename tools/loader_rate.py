"""Input-pipeline throughput: frames/s of the windowed WAV reader behind a DataLoader.

Writes two folders of 16-bit PCM WAV files (A_real / B_fake, 22 050 Hz and 44 100 Hz: the second one goes through
`sinc_resample`), builds the reference-format index with `get_costum_dataset` and times one pass over the train
split for several worker counts.  Host only (no GPU): `python tools/loader_rate.py [files per folder] [seconds per file]`.
"""
import os, sys, tempfile, time, wave

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "audiodeepfake-detection_amd"))
from audiofakedetect.data_loader import NativeFrameLoader, get_costum_dataset  # noqa: E402

files = int(sys.argv[1]) if len(sys.argv) > 1 else 24
secs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
root = tempfile.mkdtemp(prefix="afd_loader_")
rng = np.random.default_rng(0)
for name, rate in (("A_real", 22050), ("B_fake", int(os.environ.get("AFD_LOADER_RATE2", "44100")))):
    os.makedirs(os.path.join(root, name))
    for i in range(files):
        pcm = (rng.standard_normal(rate * secs) * 3000).astype(np.int16)
        with wave.open(os.path.join(root, name, f"{i:04d}.wav"), "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(rate); f.writeframes(pcm.tobytes())
ds = get_costum_dataset(data_path=root, save_path=os.path.join(root, "index"), ds_type="train", seconds=1,
                        resample_rate=22050, limit=-1)
print(f"{len(ds)} one-second frames in the train split ({files} files x {secs} s per folder; second folder at "
      f"{os.environ.get('AFD_LOADER_RATE2', '44100')} Hz)")
for workers in (0, 2, 4, 8):
    dl = torch.utils.data.DataLoader(ds, batch_size=128, shuffle=True, num_workers=workers, drop_last=True,
                                     persistent_workers=False)
    t0 = time.perf_counter(); n = 0
    for batch in dl:
        n += batch["audio"].shape[0]
    dt = time.perf_counter() - t0
    print(f"workers {workers}: {n / dt:9.0f} frames/s ({n} frames in {dt:.2f} s)")
if torch.cuda.is_available():
    for threads in (1, 4, 8, 16):
        nl = NativeFrameLoader(ds, 128, "cuda:0", shuffle=True, seed=0, drop_last=True, threads=threads)
        for _ in nl:  # warm-up: kernel bank, pinned allocations
            break
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 0
        for batch in nl:
            n += batch["audio"].shape[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"native loader, {threads:2d} reader threads: {n / dt:9.0f} frames/s ({n} frames in {dt:.2f} s)")
