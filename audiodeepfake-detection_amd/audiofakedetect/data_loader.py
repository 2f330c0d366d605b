"""Data-side pieces the hot path touches.

``WelfordEstimator`` mirrors reference ``src/audiofakedetect/data_loader.py:27-71``.
``CustomDataset`` / ``get_costum_dataset`` keep the reference's folder
convention, balanced 70/10/20 split and cached ``.npy`` index format (:74-507) with a standard-library
WAV reader in place of torchaudio; ``SyntheticFrames`` produces the same item format
(``{"audio": f32[1, N], "label": int64}``, :351-353,392) without a dataset on disk.
"""

from __future__ import annotations

import glob
import math
import os
import wave
from typing import Optional, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset


class WelfordEstimator:
    """Running mean / std over all axes but the last (the channel axis)."""

    def __init__(self) -> None:
        self.collapsed_axis: Optional[Tuple[int, ...]] = None

    def update(self, batch_vals: torch.Tensor) -> None:
        if not self.collapsed_axis:
            self.collapsed_axis = tuple(range(batch_vals.dim() - 1))
            dev = batch_vals.device
            nch = batch_vals.shape[-1]
            self.count = torch.zeros(1, device=dev, dtype=torch.float32)
            self.mean = torch.zeros(nch, device=dev, dtype=torch.float32)
            self.std = torch.zeros(nch, device=dev, dtype=torch.float32)
            self.m2 = torch.zeros(nch, device=dev, dtype=torch.float32)
        self.count += int(np.prod(batch_vals.shape[:-1]))
        delta = batch_vals - self.mean
        self.mean += torch.sum(delta / self.count, self.collapsed_axis)
        delta2 = batch_vals - self.mean
        self.m2 += torch.sum(delta * delta2, self.collapsed_axis)

    def finalize(self) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.mean, torch.sqrt(self.m2 / self.count)


class SyntheticFrames(Dataset):
    """Seeded synthetic 1 s frames: 0.1 * randn clipped to [-1, 1]."""

    key = "audio"

    def __init__(self, length: int = 1024, num_samples: int = 22050, seed: int = 1234,
                 num_labels: int = 2) -> None:
        self.length = length
        self.num_samples = num_samples
        self.seed = seed
        self.num_labels = num_labels

    def __len__(self) -> int:
        return self.length

    def __getitem__(self, idx: int) -> dict:
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + idx)
        audio = (0.1 * torch.randn(1, self.num_samples, generator=g)).clamp_(-1.0, 1.0)
        label = torch.randint(0, self.num_labels, (1,), generator=g)[0].to(torch.int64)
        return {"audio": audio, "label": label, "index": idx}


def _wav_info(path: str) -> Tuple[int, int]:
    """(frames, sample rate) of a PCM WAV file."""
    with wave.open(path, "rb") as fh:
        return fh.getnframes(), fh.getframerate()


def read_wav_window(path: str, frame_offset: int, num_frames: int) -> Tuple[torch.Tensor, int]:
    """[1, num_frames] float32 in [-1, 1) (mono: first channel) and the file's sample rate.

    Stands in for ``torchaudio.load(path, frame_offset, num_frames)`` of the reference
    (data_loader.py:323-327) for 8/16/32-bit PCM WAV files, read with the standard library.
    """
    with wave.open(path, "rb") as fh:
        rate, width, channels = fh.getframerate(), fh.getsampwidth(), fh.getnchannels()
        fh.setpos(int(frame_offset))
        raw = fh.readframes(int(num_frames))
    if width == 2:
        data = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        data = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        data = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise RuntimeError(f"{path}: unsupported PCM sample width {width}")
    data = data.reshape(-1, channels)[:, 0]
    if data.shape[0] < num_frames:
        data = np.pad(data, (0, int(num_frames) - data.shape[0]))
    return torch.from_numpy(np.ascontiguousarray(data)).unsqueeze(0), rate


_sinc_kernels: dict = {}


def _sinc_resample_kernel(orig: int, new: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """Polyphase bank [new, 1, 2 * width + orig] of the Hann-windowed sinc interpolator (orig, new already
    divided by their gcd), evaluated in float32 like the waveform it filters."""
    key = (orig, new, lowpass_filter_width, rolloff)
    hit = _sinc_kernels.get(key)
    if hit is not None:
        return hit
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float32)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float32)[:, None, None] / new + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / orig)
    _sinc_kernels[key] = (kernels, width)
    return kernels, width


def sinc_resample(waveform: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """Band-limited sinc resampling with the defaults of ``torchaudio.functional.resample`` (the call at
    reference data_loader.py:343-345; torchaudio==2.0.0, ``requirements.txt:8``: Hann-windowed sinc,
    ``lowpass_filter_width=6``, ``rolloff=0.99``).  torchaudio is a third-party dependency that is not
    installed here; this restates its published algorithm: the rates are reduced by their gcd, the signal is
    zero-padded by ``width`` on the left and ``width + orig`` on the right, filtered by a bank of ``new``
    phase kernels with stride ``orig``, the phases interleaved and cut to ``ceil(new * length / orig)``."""
    if int(orig_freq) == int(new_freq):
        return waveform
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    kernels, width = _sinc_resample_kernel(orig, new)
    shape = waveform.shape
    flat = waveform.reshape(-1, shape[-1]).to(torch.float32)
    length = flat.shape[-1]
    padded = torch.nn.functional.pad(flat, (width, width + orig))
    out = torch.nn.functional.conv1d(padded[:, None], kernels, stride=orig)
    out = out.transpose(1, 2).reshape(flat.shape[0], -1)
    target = -(-new * length // orig)
    return out[..., :target].reshape(shape[:-1] + (target,))


class CustomDataset(Dataset):
    """Balanced frame dataset over ``<letter>_<name>/`` folders of WAV files.

    Same on-disk contract as the reference's ``CustomDataset`` (data_loader.py:74-353): every file
    is cut into windows of ``seconds``; per folder the windows are split 70 / 10 / 20 into train /
    val / test in file order; every label contributes the same number of windows (the minimum over
    the folders); the split is cached as an object array ``[label][frame] = (path, frame index,
    window size in samples, label)`` in ``<save_path>/dataset_<names>_meta_<seconds>sec_<split>.npy``
    and re-used on the next run.  ``__getitem__`` reads the window (``frame index * window size``
    samples into the file), converts PCM to float32 and resamples down to ``resample_rate``.
    Folders named in ``only_test_folders`` contribute to val / test only.
    """

    def __init__(self, paths: list, labels: list, save_path: str, only_test_folders: Optional[list] = None,
                 abort_on_save: bool = False, ds_type: str = "train", seconds: float = 1,
                 resample_rate: int = 16000, train_ratio: float = 0.7, val_ratio: float = 0.1,
                 key: Optional[str] = "audio", limit: int = 555000, verbose: Optional[bool] = False,
                 filetype: str = "wav", asvspoof_name: Optional[str] = None) -> None:
        if ds_type not in ("train", "val", "test"):
            raise RuntimeError("Dataset type does not exists.")
        if ds_type == "train" and only_test_folders:
            raise ValueError("Since there are folders in only_test_folders this cannot be a train dataset.")
        names = [str(p).rstrip("/").split("/")[-1].split("_")[-1] for p in paths]
        self.label_names = {lab: name for lab, name in zip(labels, names)}
        destination = f"{save_path}/dataset_{'-'.join(names)}_meta_{seconds}sec"
        cache = f"{destination}_{ds_type}.npy"
        if os.path.exists(cache):
            result_set = np.load(cache, allow_pickle=True)
        else:
            os.makedirs(save_path, exist_ok=True)
            splits: dict = {"train": [], "val": [], "test": []}
            counts = []
            for path, label, name in zip(paths, labels, names):
                pattern = f"{asvspoof_name}*.{filetype}" if asvspoof_name else f"*.{filetype}"
                rows = []
                for file_name in sorted(glob.glob(os.path.join(str(path), pattern))):
                    frames, rate = _wav_info(file_name)
                    win = int(seconds * rate)
                    rows += [(file_name, i, win, label) for i in range(frames // win)]
                arr = np.empty((len(rows), 4), dtype=object)
                for r, row in enumerate(rows):
                    arr[r] = row
                n = len(rows)
                if only_test_folders and name in only_test_folders:
                    n_train = 0
                    if counts and n >= counts[-1][1] + counts[-1][2]:
                        n_val, n_test = counts[-1][1], counts[-1][2]
                    else:
                        n_val = int(val_ratio / (1.0 - train_ratio) * n)
                        n_test = n - n_val
                else:
                    n_train, n_val = int(train_ratio * n), int(val_ratio * n)
                    n_test = n - n_train - n_val
                splits["train"].append(arr[:n_train])
                splits["val"].append(arr[n_train:n_train + n_val])
                splits["test"].append(arr[n_train + n_val:n_train + n_val + n_test])
                if only_test_folders and name in only_test_folders:
                    n_train = counts[-1][0] if counts else (55500 if limit == -1 else limit)
                counts.append([n_train, n_val, n_test])
            mins = np.asarray(counts).min(axis=0)
            which = {"train": 0, "val": 1, "test": 2}[ds_type]
            result_set = np.stack([a[:mins[which]] for a in splits[ds_type]]) if mins[which] > 0 \
                else np.empty((len(paths), 0, 4), dtype=object)
            np.save(cache, result_set, allow_pickle=True)
            if abort_on_save:
                raise SystemExit("Aborting on dataset saving.")
        result_set = result_set[:, :limit] if limit is not None and limit >= 0 else result_set
        if result_set.shape[1] and resample_rate > int(result_set[:, :, 2].min() / seconds):
            raise RuntimeError("Sample rate is smaller than desired sample rate. No upsampling possible here.")
        self.audio_data = result_set.reshape(-1, 4)  # (number of samples, 4)
        self.ds_type = ds_type
        self.key = key
        self.resample_rate = resample_rate
        self.seconds = seconds
        if verbose:
            print(f"{ds_type}: {len(self)} frames from {names}", flush=True)

    def get_label_name(self, key) -> str:
        return self.label_names.get(key, f"John Doe Generator {key}")

    def __len__(self) -> int:
        return int(len(self.audio_data))

    def _load(self, idx: int) -> Tuple[torch.Tensor, int]:
        path, frame, win, _ = self.audio_data[idx]
        audio, rate = read_wav_window(str(path), int(frame) * int(win), int(win))
        if rate > self.resample_rate:
            audio = sinc_resample(audio, int(rate), int(self.resample_rate))
        elif rate < self.resample_rate:
            raise RuntimeError("Sample rate is smaller than desired sample rate. No upsampling possible here.")
        return audio, rate

    def __getitem__(self, idx: int) -> dict:
        audio, _ = self._load(idx)
        return {self.key: audio, "label": torch.tensor(int(self.audio_data[idx, 3]))}


class CustomDatasetDetailed(CustomDataset):
    """``__getitem__`` additionally names the file, frame and offset (reference :356-394)."""

    def __getitem__(self, idx: int) -> dict:
        audio, rate = self._load(idx)
        path, frame, win, label = self.audio_data[idx]
        return {self.key: audio, "label": torch.tensor(int(label)), "sample_rate": rate, "index": idx,
                "path": str(path), "frame": int(frame), "offset": int(frame) * int(win)}


class NativeFrameLoader:
    """Batches of a `CustomDataset` index without per-item Python work.

    The windows of a batch are read by the library's threaded WAV reader (`afd_wav_read_windows`) into one pinned
    int16 buffer, copied to the GPU, and converted / resampled there (`afd_pcm16_resample`, the interpolator of
    `sinc_resample`) -- the reference does this per item in DataLoader workers (data_loader.py:323-353).  Yields
    the reference's item format batched: ``{"audio": f32[B, 1, n] on the device, "label": int64[B]}``.
    Sharding over ranks, shuffling and ``set_epoch`` follow ``DistributedSampler(shuffle, seed, drop_last=True)``.
    A batch whose windows differ in rate or length, or a file that is not 16-bit PCM, goes through the dataset's
    own ``__getitem__`` (same values, slower).
    """

    AUTO_STEP_MS = 15.0  # prefetch pays above this consumer step time (level-14 steps: 23-45 ms), not on the 5 ms steps
    AUTO_PROBE = 3       # consumer gaps measured before the choice
    WORKER_PAUSE = 0.2   # the prefetch thread stays idle for this part of the measured step after handing a batch over ...
    WORKER_PAUSE_CAP_MS = 10.0  # ... at most this long

    def __init__(self, dataset: "CustomDataset", batch_size: int, device, shuffle: bool = True, seed: int = 0,
                 drop_last: bool = True, rank: int = 0, world: int = 1, threads: int = 8, prefetch="auto",
                 prefetch_readers: int = 2) -> None:
        self.ds = dataset
        self.dataset = dataset  # the attribute the trainer reads from a torch DataLoader
        self.batch_size = int(batch_size)
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())  # the worker thread needs an explicit index
        self.shuffle, self.seed, self.drop_last = shuffle, seed, drop_last
        self.rank, self.world, self.threads = rank, world, threads
        self.epoch = 0
        self._banks: dict = {}
        # prefetch: 1 = batches prepared one ahead of the consumer by a background thread on a side stream; 0 = in the
        # caller's thread, on its stream; "auto" (default; AFD_LOADER_PREFETCH=0/1 overrides) = start in the caller's
        # thread, measure the time the consumer spends between two batches over the first AUTO_PROBE of them, and hand
        # the rest of the epoch -- and the following epochs -- to the background thread when that is above
        # AUTO_STEP_MS.  A trainer that reads its loss back every step (the reference's does, train_classifier.py:981-989)
        # only asks for batch k + 1 when step k has finished: read, copy and resampling of the next batch then all lie
        # between two steps (+1.1-1.4 ms on the 45 ms level-14 step, +0.7-0.9 when they were done during step k); on the
        # 5 ms level-8 / STFT steps the trainer's own thread is busy issuing launches and the worker takes its core.
        env = os.environ.get("AFD_LOADER_PREFETCH")
        if env is not None:
            prefetch = int(env)
        if self.device.type != "cuda":
            prefetch = 0
        self.prefetch = prefetch if prefetch == "auto" else int(prefetch)
        self._auto_choice = None  # the measured decision of an "auto" loader (kept over epochs)
        self.consumer_ms = None   # the consumer's measured time between batches (an "auto" loader's probe)
        self._side = None
        self._stale_workers: list = []
        # reader threads of a prefetching loader: it has a whole step to read 128 windows (5.6 MB from the page cache), and
        # more threads take the cores the trainer's own thread needs to keep the GPU's queue filled -- level-14 step,
        # B = 128 (tools/e2e_probe.py): resident batch 44.5-44.9 ms; in the caller's thread 45.5 (16 readers) / 46.0 (1);
        # prefetching with 16 readers 50.3, with 4 / 2 / 1: 45.3 / 45.05 / 45.2
        self.prefetch_readers = max(1, int(prefetch_readers))
        self._readers = self.threads

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def _indices(self) -> np.ndarray:
        n = len(self.ds)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            perm = torch.randperm(n, generator=g).numpy()
        else:
            perm = np.arange(n)
        if self.drop_last:
            total = n // self.world * self.world
        else:  # DistributedSampler(drop_last=False): the shards are evened out by wrapping around
            total = -(-n // self.world) * self.world
            if total > n:
                perm = np.concatenate([perm, perm[:total - n]])
        return perm[self.rank:total:self.world]

    def __len__(self) -> int:
        n = len(self._indices())
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def _bank(self, orig: int, new: int):
        key = (orig, new)
        hit = self._banks.get(key)
        if hit is None:
            kernels, width = _sinc_resample_kernel(orig, new)
            hit = (kernels.reshape(new, -1).contiguous().to(self.device), width)
            self._banks[key] = hit
        return hit

    def _fallback(self, idxs) -> dict:
        items = [self.ds[int(i)] for i in idxs]
        return {self.ds.key: torch.stack([it[self.ds.key] for it in items]).to(self.device, non_blocking=True),
                "label": torch.stack([it["label"] for it in items]).to(self.device)}

    def _pinned(self, n: int, win: int):
        """A pinned staging buffer of this shape that no copy is reading: two per shape, used alternately,
        each with the event recorded behind its last host-to-device copy -- the file read of batch k + 1
        waits for the COPY of batch k - 1 only, never for the training step that consumes batch k."""
        key = ("pin", n, win)
        ring = self._banks.get(key)
        if ring is None:
            ring = {"bufs": [torch.empty((n, win), dtype=torch.int16).pin_memory() for _ in range(2)],
                    "events": [None, None], "next": 0}
            self._banks[key] = ring
        slot = ring["next"]
        ring["next"] = 1 - slot
        if ring["events"][slot] is not None:
            ring["events"][slot].synchronize()
        return ring, slot

    def __iter__(self):
        import time

        mode = self.prefetch
        if mode == "auto" and self._auto_choice is not None:
            mode = self._auto_choice
        if mode == "auto":
            # probe: the first batches in the caller's thread, timing what the consumer does between them
            gaps = []
            done = 0
            for item in self._batches(0, self.AUTO_PROBE + 1):
                t0 = time.perf_counter()
                yield item
                gaps.append(1e3 * (time.perf_counter() - t0))
                done += 1
            if done < self.AUTO_PROBE + 1:
                return  # a short epoch: nothing left to hand over (the choice is made on a later, longer one)
            # the median of the gaps after the first (which holds the consumer's warm-up): one stall -- a garbage
            # collection, a checkpoint -- does not decide
            self.consumer_ms = sorted(gaps[1:])[len(gaps[1:]) // 2]
            self._auto_choice = 1 if self.consumer_ms >= self.AUTO_STEP_MS else 0
            mode = self._auto_choice
            start = done
        else:
            start = 0
        if mode <= 0:
            self._readers = self.threads
            yield from self._batches(start)
            return
        yield from self._prefetched(start, int(mode))

    def _prefetched(self, start: int, depth: int):
        import queue
        import threading
        import time
        import warnings

        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        side = self._side
        self._readers = min(self.threads, self.prefetch_readers)
        q: "queue.Queue" = queue.Queue(maxsize=depth)
        stop = threading.Event()

        def put(x) -> bool:
            while not stop.is_set():
                try:
                    q.put(x, timeout=0.05)
                    return True
                except queue.Full:
                    continue
            return False

        # (tools/e2e_probe.py, level-14 step, same box: resident batch 43.96 ms, prefetching 44.64, with this pause 44.44-44.49)
        delay = min(1e-3 * self.WORKER_PAUSE_CAP_MS, 1e-3 * self.WORKER_PAUSE * self.consumer_ms) if self.consumer_ms else 0.0

        def worker() -> None:
            try:
                torch.cuda.set_device(self.device)
                with torch.cuda.stream(side):  # (thread-local: the library calls below pick it up as the current stream)
                    for item in self._batches(start):
                        ev = torch.cuda.Event()
                        ev.record(side)
                        if not put((item, ev)):
                            return
                        # the consumer has just taken a batch and is issuing its step's launches: stay off the
                        # interpreter lock for the first fifth of the step (at most 10 ms) before preparing the next one
                        if delay > 0.0:
                            time.sleep(delay)
                put(None)
            except BaseException as e:  # handed to the consumer
                put(e)

        th = threading.Thread(target=worker, name="afd-loader", daemon=True)
        th.start()
        try:
            while True:
                got = q.get()
                if got is None:
                    break
                if isinstance(got, BaseException):
                    raise got
                item, ev = got
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                for t in item.values():
                    if t.is_cuda:
                        t.record_stream(cur)  # allocated on the side stream, used (and freed) on the consumer's
                yield item
        finally:
            stop.set()
            th.join(timeout=10.0)
            if th.is_alive():
                # stuck in a file read or an event wait: keep it referenced (it still owns buffers of this loader) and say so
                self._stale_workers.append(th)
                warnings.warn("NativeFrameLoader: the prefetch thread did not stop within 10 s; it is left running")

    def _batches(self, start: int = 0, stop: Optional[int] = None):
        """Batches start .. stop-1 of this epoch's order (all of them by default)."""
        import ctypes

        from . import _native

        lib = _native.load()
        idx = self._indices()
        rows = self.ds.audio_data
        target = int(self.ds.resample_rate)
        n_target = int(round(target * float(self.ds.seconds)))
        stream = torch.cuda.current_stream(self.device)
        for b in range(start, len(self) if stop is None else min(stop, len(self))):
            sel = idx[b * self.batch_size:(b + 1) * self.batch_size]
            batch = rows[sel]
            n = len(sel)
            wins = np.asarray([int(w) for w in batch[:, 2]])
            out = None
            ok = True
            # windows of one length (= one file rate) are read and resampled together
            for win in np.unique(wins):
                pos = np.nonzero(wins == win)[0]
                m = len(pos)
                win = int(win)
                ring, slot = self._pinned(m, win)
                pcm = ring["bufs"][slot]
                rates = (ctypes.c_int * m)()
                paths = (ctypes.c_char_p * m)(*[os.fsencode(str(p)) for p in batch[pos, 0]])
                offs = (ctypes.c_longlong * m)(*[int(f) * win for f in batch[pos, 1]])
                rc = lib.afd_wav_read_windows(paths, offs, m, win, ctypes.c_void_p(pcm.data_ptr()), rates, self._readers)
                rate = int(rates[0]) if rc == 0 else 0
                if rc != 0 or any(int(r) != rate for r in rates) or rate < target:
                    ok = False  # the dataset's own path raises its own errors (e.g. rate < target)
                    break
                g = math.gcd(rate, target)
                orig, new = rate // g, target // g
                n_out = win if orig == new else -(-new * win // orig)
                if out is None:
                    out = torch.empty((n, 1, n_out), dtype=torch.float32, device=self.device)
                if n_out != out.shape[-1]:
                    ok = False
                    break
                dev_pcm = pcm.to(self.device, non_blocking=True)
                ev = ring["events"][slot] or torch.cuda.Event()
                ev.record(stream)
                ring["events"][slot] = ev
                bank, width = (None, 0) if orig == new else self._bank(orig, new)
                part = out if m == n else torch.empty((m, 1, n_out), dtype=torch.float32, device=self.device)
                _native.check(lib.afd_pcm16_resample(_native.ptr(dev_pcm), m, win, orig, new, width, _native.ptr(bank),
                                                     _native.ptr(part), n_out, _native.stream_ptr()),
                              "afd_pcm16_resample")
                if m != n:
                    out.index_copy_(0, torch.as_tensor(pos, device=self.device), part)
            if not ok:
                yield self._fallback(sel)
                continue
            labels = torch.tensor([int(v) for v in batch[:, 3]], dtype=torch.int64)
            yield {self.ds.key: out, "label": labels.to(self.device, non_blocking=True)}


def get_costum_dataset(data_path=None, save_path=None, ds_type="train", only_test_folders=None,
                       only_use=None, seconds=1, resample_rate=22050, limit=55504, abort_on_save=False,
                       asvspoof_name=None, train_ratio=0.7, val_ratio=0.1, file_type="wav",
                       get_details=False, synthetic=False, **_):
    """Dataset factory (name kept from the reference, data_loader.py:396-507).

    Folders ``<letter>_<name>`` under ``data_path`` become labels ``ord(letter) - 65`` (A = real),
    the next free integer on a clash; ``only_use`` filters by ``<name>``.  ``synthetic=True`` (or no
    ``data_path``) returns seeded synthetic frames instead.
    """
    if synthetic or data_path is None:
        length = int(limit) if limit else 1024
        seed = {"train": 1234, "val": 4321, "test": 9876}.get(ds_type, 1)
        return SyntheticFrames(length, int(resample_rate * (seconds or 1)), seed)
    paths = sorted(glob.glob(os.path.join(str(data_path), "*_*")))
    if not paths:
        raise RuntimeError("Given data_path is empty.")
    labels: list = []
    use: list = []
    for path in paths:
        base = path.rstrip("/").split("/")[-1]
        if only_use is not None and base.split("_")[-1] not in only_use:
            continue
        want = ord(base.split("_")[0][0]) - 65
        while want in labels:
            want += 1
        labels.append(want)
        use.append(path)
    if 0 not in labels and ds_type == "train":
        raise RuntimeError("No real training data. Aborting...")
    cls = CustomDatasetDetailed if get_details else CustomDataset
    return cls(paths=use, labels=labels, save_path=save_path, abort_on_save=abort_on_save, seconds=seconds,
               resample_rate=resample_rate, verbose=False, limit=limit, ds_type=ds_type,
               only_test_folders=only_test_folders, asvspoof_name=asvspoof_name, train_ratio=train_ratio,
               val_ratio=val_ratio, filetype=file_type)
