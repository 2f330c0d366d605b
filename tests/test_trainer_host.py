"""Host-side trainer helpers (no GPU): the reference's own aggregation known answers
(reference tests/test_trainer.py:14-117) and the EER against sklearn."""

import numpy as np
import pytest
import torch

from audiofakedetect.train_classifier import Trainer
from audiofakedetect.utils import DotDict, _Griderator, build_new_grid


def test_key_error_and_result_type():
    with pytest.raises(KeyError):
        Trainer.calculate_acc_label([{1: 1}, {2: 1}], [{1: [torch.tensor(False)], 2: []}], key=2)
    assert isinstance(Trainer.calculate_acc_label([{1: 1}], [{1: [], 2: []}], key=1), float)
    assert isinstance(Trainer.calculate_acc_label([{1: 1}], [{1: [True], 2: []}], key=1), float)


def _gathered(third_second_rank):
    counts = [{1: 3, 3: 2, 2: 1, 0: 1}, {1: 3, 3: 1, 2: 1, 0: 2}]
    t, f = torch.tensor(True), torch.tensor(False)
    oks = [{1: [t, f, f], 3: [t, t], 2: [t], 0: [f]},
           {1: [t, t, f], 3: [third_second_rank], 2: [t], 0: [f, f]}]
    return counts, oks


def test_accuracy_over_two_ranks():
    counts, oks = _gathered(torch.tensor(True))
    assert Trainer.calculate_acc_label(counts, oks, key=1) == pytest.approx(3 / 6)
    assert Trainer.calculate_acc_label(counts, oks, key=0) == pytest.approx(0.0)


def test_accuracy_dict_known_answers():
    counts, oks = _gathered(torch.tensor(False))

    class _DS:
        def get_label_name(self, key):
            return {0: "Zero", 1: "First", 2: "Second", 3: "Third"}.get(key, "")

    class _DL:
        dataset = _DS()

    got = Trainer.caculate_acc_dict(_DL(), {0, 1, 2, 3}, oks, counts)
    assert got == [("Zero", 0.0), ("First", 0.5), ("Second", 1.0), ("Third", 0.6666666865348816)]


def test_eer_matches_sklearn_brentq_recipe():
    from scipy.interpolate import interp1d
    from scipy.optimize import brentq
    from sklearn.metrics import roc_curve

    rng = np.random.default_rng(0)
    for hard in (True, False):
        for _ in range(5):
            y = rng.integers(0, 2, 300)
            score = np.clip(y * 0.4 + rng.normal(0.3, 0.35, 300), 0, 1)
            if hard:
                score = (score > 0.5).astype(np.int64)  # the reference scores hard predictions
            fpr, tpr, _ = roc_curve(y, score, pos_label=1)
            ref = brentq(lambda x: 1.0 - x - interp1d(fpr, tpr)(x), 0.0, 1.0)
            assert Trainer.calculate_eer(y, score) == pytest.approx(ref, abs=1e-9)


def test_dotdict_and_griderator():
    d = DotDict(a=1)
    d.b = 2
    assert d.a == 1 and d["b"] == 2 and d.missing is None
    g = _Griderator({"lr": [1, 2], "wd": [3]}, init_seeds=[0, 1])
    assert g.get_len() == 4 and list(g.get_keys()) == ["seed", "lr", "wd"]
    args, nxt = g.update_step(DotDict())
    assert (args.seed, args.lr, args.wd) == (0, 1, 3) and nxt == (0, 2, 3)
    assert build_new_grid({"x": [1]}, seeds=["5", 6]).init_config["seed"] == [5, 6]
    with pytest.raises(TypeError):
        _Griderator([1, 2])


def test_integrated_gradient_helpers():
    # straight path, trapezoidal rule, running mean (reference integrated_gradients.py:13-47,104-138)
    import torch
    from audiofakedetect.integrated_gradients import Mean, integral_approximation, interpolate_images

    base = torch.zeros(2, 3, 4)
    img = torch.arange(24, dtype=torch.float32).reshape(2, 3, 4)
    alphas = torch.tensor([0.0, 0.25, 1.0])
    path = interpolate_images(base, img, alphas)
    assert path.shape == (3, 2, 3, 4)
    assert torch.equal(path[0], base) and torch.equal(path[2], img) and torch.allclose(path[1], 0.25 * img)
    g = torch.stack([torch.full((2, 2), v) for v in (1.0, 3.0, 5.0)])
    assert torch.allclose(integral_approximation(g), torch.full((2, 2), 3.0))  # ((1+3)/2 + (3+5)/2) / 2
    m = Mean()
    for v in (1.0, 2.0, 6.0):
        m.update(torch.full((1, 2, 2), v))
    assert torch.allclose(m.finalize(), torch.full((2, 2), 3.0))


def test_custom_dataset_index_split_and_window_reader(tmp_path):
    # folder convention, balanced 70/10/20 split, cached .npy index format and windowed WAV reads
    # of the reference's CustomDataset (data_loader.py:74-353, 396-507)
    import wave

    import numpy as np
    import torch
    from audiofakedetect.data_loader import get_costum_dataset

    rate = 8000
    rng = np.random.default_rng(0)
    signals = {}
    for folder, files in (("A_real", (10, 7)), ("B_fakegan", (9, 12)), ("C_other", (30,))):
        (tmp_path / "data" / folder).mkdir(parents=True)
        for j, secs in enumerate(files):
            pcm = rng.integers(-20000, 20000, size=int(secs * rate + 123), dtype=np.int16)
            path = tmp_path / "data" / folder / f"f{j}.wav"
            with wave.open(str(path), "wb") as fh:
                fh.setnchannels(1); fh.setsampwidth(2); fh.setframerate(rate)
                fh.writeframes(pcm.tobytes())
            signals[str(path)] = pcm
    kw = dict(data_path=str(tmp_path / "data"), save_path=str(tmp_path / "meta"), only_use=["real", "fakegan"],
              seconds=1, resample_rate=rate, limit=1000)
    sets = {t: get_costum_dataset(ds_type=t, **kw) for t in ("train", "val", "test")}
    # 17 and 21 one-second windows -> 11/1/5 and 14/2/5 -> balanced to 11/1/5 per label
    assert [len(sets[t]) for t in ("train", "val", "test")] == [22, 2, 10]
    idx = np.load(tmp_path / "meta" / "dataset_real-fakegan_meta_1sec_train.npy", allow_pickle=True)
    assert idx.shape == (2, 11, 4) and idx[1, 0, 3] == 1 and idx[0, 3, 1] == 3 and idx[0, 0, 2] == rate
    item = sets["train"][14]  # label 1, its 4th training window
    path, frame, win, label = sets["train"].audio_data[14]
    assert label == 1 and item["label"].item() == 1 and item["audio"].shape == (1, rate)
    want = signals[str(path)][frame * win:(frame + 1) * win].astype(np.float32) / 32768.0
    assert torch.equal(item["audio"][0], torch.from_numpy(want))
    assert sets["train"].get_label_name(1) == "fakegan" and len(get_costum_dataset(ds_type="train", **kw)) == 22
    # resampling down (windowed sinc) keeps the one-second frame length at the new rate
    half = get_costum_dataset(ds_type="val", **{**kw, "resample_rate": rate // 2})
    assert half[0]["audio"].shape == (1, rate // 2)
    import pytest
    with pytest.raises(RuntimeError):
        get_costum_dataset(ds_type="val", **{**kw, "resample_rate": 2 * rate})


def test_graycode_keys_match_oracle_order():
    from audiofakedetect.wavelet_math import graycode_keys
    from oracle.wpt_oracle import graycode_paths

    assert graycode_keys(1) == ["a", "d"]
    assert graycode_keys(2) == ["aa", "ad", "dd", "da"]
    for level in (3, 8, 14):
        assert graycode_keys(level) == graycode_paths(level)


def test_sinc_resample_known_answers():
    """`sinc_resample` (torchaudio.functional.resample defaults, reference data_loader.py:343-345): kernel
    geometry for 44 100 -> 22 050 (gcd-reduced 2 -> 1: width ceil(6 * 2 / 0.99) = 13, 28 taps), output
    length ceil(new * n / orig), unit DC gain, a tone below the new Nyquist frequency keeps amplitude and
    phase, a tone above it is removed, and agreement with scipy's polyphase resampler on a band-limited
    signal away from the edges."""
    import math

    from scipy.signal import resample_poly

    from src.audiofakedetect.data_loader import _sinc_resample_kernel, sinc_resample

    k, width = _sinc_resample_kernel(2, 1)
    assert width == 13 and tuple(k.shape) == (1, 1, 28)
    assert abs(float(k.sum()) - 1.0) < 2e-3
    k3, w3 = _sinc_resample_kernel(160, 147)  # 48 000 -> 44 100
    assert w3 == math.ceil(6 * 160 / (147 * 0.99)) and tuple(k3.shape) == (147, 1, 2 * w3 + 160)

    n = 44100
    t = torch.arange(n, dtype=torch.float64) / 44100.0
    for orig, new in ((44100, 22050), (48000, 22050), (24000, 22050)):
        x = torch.randn(2, 1, 1000)
        assert sinc_resample(x, orig, new).shape == (2, 1, -(-new * 1000 // orig))
    assert sinc_resample(torch.ones(5), 22050, 22050).shape == (5,)

    low = torch.sin(2 * math.pi * 3000.0 * t).float()[None]
    y = sinc_resample(low, 44100, 22050)
    ref = torch.sin(2 * math.pi * 3000.0 * torch.arange(22050, dtype=torch.float64) / 22050.0).float()
    assert (y[0, 100:-100] - ref[100:-100]).abs().max() < 2e-3
    high = torch.sin(2 * math.pi * 15000.0 * t).float()[None]
    assert sinc_resample(high, 44100, 22050)[0, 100:-100].abs().max() < 5e-2
    mix = (low + 0.5 * torch.sin(2 * math.pi * 700.0 * t).float()[None])
    poly = resample_poly(mix.numpy().astype(np.float64), 1, 2, axis=-1)
    assert np.abs(sinc_resample(mix, 44100, 22050)[0, 200:-200].numpy() - poly[0, 200:-200]).max() < 5e-3


def _write_wav(path, pcm, rate, channels=1, extra_chunk=False):
    import struct

    data = pcm.astype("<i2").tobytes()
    fmt = struct.pack("<HHIIHH", 1, channels, rate, rate * channels * 2, channels * 2, 16)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if extra_chunk:
        body += b"LIST" + struct.pack("<I", 5) + b"abcde" + b"\x00"  # odd-sized chunk + pad byte
    body += b"data" + struct.pack("<I", len(data)) + data
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def test_native_wav_window_reader(tmp_path):
    """`afd_wav_read_windows` (the batch form of the reference's torchaudio.load(path, frame_offset, num_frames),
    data_loader.py:323-327) against the standard-library reader: mono and stereo files, a chunk between "fmt " and
    "data", windows that run past the end (zero-filled), several threads; a non-PCM file is refused."""
    import ctypes

    from audiofakedetect import _native
    from audiofakedetect.data_loader import read_wav_window

    lib = _native.load()
    rng = np.random.default_rng(3)
    specs = [("a.wav", 22050, 1, False, 50000), ("b.wav", 44100, 2, True, 70001), ("c.wav", 16000, 1, True, 3000)]
    for name, rate, ch, extra, frames in specs:
        _write_wav(tmp_path / name, rng.integers(-30000, 30000, size=frames * ch), rate, ch, extra)
    win = 4096
    items = [("a.wav", 0), ("a.wav", 45000), ("b.wav", 1234), ("b.wav", 69000), ("c.wav", 0), ("c.wav", 2999),
             ("a.wav", 60000)]
    n = len(items)
    paths = (ctypes.c_char_p * n)(*[str(tmp_path / p).encode() for p, _ in items])
    offs = (ctypes.c_longlong * n)(*[o for _, o in items])
    out = np.full((n, win), 77, dtype=np.int16)
    rates = (ctypes.c_int * n)()
    for threads in (1, 4):
        rc = lib.afd_wav_read_windows(paths, offs, n, win, out.ctypes.data_as(ctypes.c_void_p), rates, threads)
        assert rc == 0, lib.afd_last_error()
        for i, (p, o) in enumerate(items):
            if p == "a.wav" and o == 60000:
                assert not out[i].any()  # the window starts past the end: all zeros (wave.setpos would raise)
                continue
            ref, rate = read_wav_window(str(tmp_path / p), o, win)
            assert rates[i] == rate
            assert np.array_equal(out[i].astype(np.float32) / 32768.0, ref[0].numpy())
    with open(tmp_path / "bad.wav", "wb") as f:
        f.write(b"RIFF\x00\x00\x00\x00WAVEjunk")
    bad = (ctypes.c_char_p * 1)(str(tmp_path / "bad.wav").encode())
    assert lib.afd_wav_read_windows(bad, offs, 1, win, out.ctypes.data_as(ctypes.c_void_p), rates, 1) != 0
    # a negative frame offset is an argument error (it used to read header bytes as samples)
    one = (ctypes.c_char_p * 1)(str(tmp_path / "a.wav").encode())
    neg = (ctypes.c_longlong * 1)(-5)
    assert lib.afd_wav_read_windows(one, neg, 1, win, out.ctypes.data_as(ctypes.c_void_p), rates, 1) != 0
    assert b"negative" in lib.afd_last_error()
    # a file cut short of what its header promises, and a streamed file whose data size reads 0xFFFFFFFF:
    # what is there, then zeros
    raw = open(tmp_path / "a.wav", "rb").read()
    data_at = raw.index(b"data") + 8
    keep = 1000  # frames
    with open(tmp_path / "cut.wav", "wb") as f:
        f.write(raw[:data_at + 2 * keep])
    with open(tmp_path / "stream.wav", "wb") as f:
        f.write(raw[:data_at - 4] + b"\xff\xff\xff\xff" + raw[data_at:data_at + 2 * keep])
    full = np.frombuffer(raw[data_at:data_at + 2 * keep], dtype=np.int16)
    for name in ("cut.wav", "stream.wav"):
        one = (ctypes.c_char_p * 1)(str(tmp_path / name).encode())
        zero = (ctypes.c_longlong * 1)(0)
        out[0] = 77
        assert lib.afd_wav_read_windows(one, zero, 1, win, out.ctypes.data_as(ctypes.c_void_p), rates, 1) == 0, name
        assert np.array_equal(out[0, :keep], full) and not out[0, keep:].any(), name
    # block_align that contradicts channels * 2 is refused
    fmt_at = raw.index(b"fmt ") + 8
    with open(tmp_path / "align.wav", "wb") as f:
        f.write(raw[:fmt_at + 12] + b"\x04\x00" + raw[fmt_at + 14:])
    one = (ctypes.c_char_p * 1)(str(tmp_path / "align.wav").encode())
    assert lib.afd_wav_read_windows(one, (ctypes.c_longlong * 1)(0), 1, win, out.ctypes.data_as(ctypes.c_void_p), rates, 1) != 0


def test_snapshot_name_reproduces_the_reference_file_names():
    """The stem composed from the grid-search settings (scripts/gridsearch_config.py, scripts/train.sh) is the
    file name of the checkpoints the reference ships under models/ (train_classifier.py:1162, :1221-1267)."""
    from audiofakedetect.train_classifier import snapshot_name
    from audiofakedetect.utils import DotDict

    class M:
        def get_name(self):
            return "DCNN"

    a = DotDict(transform="packets", wavelet="sym5", features="none", hop_length=220, sample_rate=22050,
                window_size=22050, num_of_scales=256, f_min=1, f_max=11025, learning_rate=0.0004,
                weight_decay=0.001, batch_size=128, nclasses=2, epochs=10, loss_less="False", aug_contrast=False,
                aug_noise=False, power=2.0, only_use=["ljspeech", "fbmelgan"], seconds=1, seed=0,
                data_prefix="/p/data/model_22050_22050_0.7_fbmelgan", model="modules")
    assert snapshot_name(a, M()) == ("model_packetssym5_none_220_22050_22050_256_1-11025_0.7_0.0004_0.001_128_2_10e_"
                                     "DCNN_signsFalse_augcFalse_augnFalse_power2.0_fbmelgan_1secs_0")
    a.transform = "stft"
    assert snapshot_name(a, M()).startswith("model_stft_none_220_")
    # --model lcnn: the reference writes "customModel" whatever the class calls itself (train_classifier.py:1197)
    a.model = "lcnn"
    assert "_10e_customModel_signsFalse_" in snapshot_name(a, M())
    a.model, a.loss_less = "modules", "True"
    assert "_10e_DCNN_signsTrue_" in snapshot_name(a, M())
    a.loss_less = "yes"  # anything but "False" switches the sign channel on (:1167)
    assert "_signsTrue_" in snapshot_name(a, M())
    a.loss_less = "False"
    a.data_prefix, a.only_use = "../data/fake", None  # the defaults: fields the reference could not index are left out
    assert snapshot_name(a, M()).startswith("fake_stft_none_220_22050_22050_256_1-11025_0.0004_")


def test_tensorboard_scalars_use_the_reference_tags():
    """With a writer the trainer emits the reference's scalars (train_classifier.py:879-883, :936-943, :991-995);
    `--tensorboard` without the tensorboard package warns instead of being dropped silently."""
    import warnings

    from audiofakedetect import train_classifier as tc
    from audiofakedetect.utils import DotDict

    class W:
        def __init__(self):
            self.rows = []

        def add_scalar(self, tag, v, step):
            self.rows.append((tag, v, step))

    t = tc.Trainer.__new__(tc.Trainer)
    t.args = DotDict(ddp=False)
    t.writer = W()
    t.step_total = 7
    t._scalars({"loss/train": 0.5, "accuracy/train": 0.75})
    assert t.writer.rows == [("loss/train", 0.5, 7), ("accuracy/train", 0.75, 7)]
    t.val_data_loader, t.cross_loader_val, t.validation_list = object(), None, []
    t.val_test_loop = lambda loader, name="": (0.9, 0.1)
    t._run_validation(3)
    tags = [r[0] for r in t.writer.rows[2:]]
    assert tags == ["accuracy/validation", "eer/validation", "accuracy/cross_validation", "eer/cross_validation", "epochs"]
    t.writer = None
    t._scalars({"loss/train": 1.0})  # no writer: nothing to do

    a = DotDict(tensorboard=True, ddp=False, log_dir="/tmp/x", transform="stft", wavelet="sym5", features="none",
                batch_size=2, learning_rate=1e-3, weight_decay=0.0, epochs=1, f_min=1, f_max=2, num_of_scales=256,
                loss_less="False", aug_contrast=False, aug_noise=False, power=2.0, only_use=None, seed=0)
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            assert tc.make_writer(a, "DCNN") is None
        assert any("--tensorboard" in str(w.message) for w in rec)
    a.tensorboard = False
    assert tc.make_writer(a, "DCNN") is None
