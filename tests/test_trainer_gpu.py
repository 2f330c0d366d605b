"""End-to-end trainer on the GPU: the reference's entry point and plugin protocol on synthetic
frames (python -m src.audiofakedetect.train_classifier ... --config <get_config file>)."""

import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_entry_point_runs_like_train_sh(tmp_path):
    cmd = [sys.executable, "-m", "src.audiofakedetect.train_classifier", "--log-dir", str(tmp_path),
           "--transform", "packets", "--wavelet", "sym5", "--num-of-scales", "256", "--log-scale",
           "--model", "modules", "--init-seeds", "0", "--synthetic", "--ckpt-every", "1",
           "--config", os.path.join(ROOT, "tests", "synthetic_config.py")]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "seed 0: steps 2" in out.stdout, out.stdout[-2000:]
    # --ckpt-every 1 saved a snapshot after epoch 0; --only-testing evaluates it and must not train
    # (reference train_classifier.py:1313-1316)
    snaps = [f for f in os.listdir(tmp_path / "models") if f.endswith(".pt")]
    assert len(snaps) == 1 and snaps[0].startswith("fake_packetssym5_none_220_22050_") and "_DCNN_signsFalse_" in snaps[0]
    out2 = subprocess.run(cmd + ["--only-testing"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out2.returncode == 0, out2.stdout[-2000:] + out2.stderr[-2000:]
    assert "seed 0: steps 0" in out2.stdout and "results seed 0" in out2.stdout, out2.stdout[-2000:]


def test_entry_point_with_the_default_wavelet(tmp_path):
    """No --wavelet flag: the reference's default sym8 (utils.py:84-89), as `scripts/train.sh` runs when its fourth
    argument is sym8 (scripts/start_exps.sh:9); flattend_size 320 / time_dim_add 0 come from the grid config."""
    cmd = [sys.executable, "-m", "src.audiofakedetect.train_classifier", "--log-dir", str(tmp_path),
           "--transform", "packets", "--num-of-scales", "256", "--log-scale", "--model", "modules",
           "--init-seeds", "0", "--synthetic", "--config", os.path.join(ROOT, "tests", "synthetic_config.py")]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, AFD_TEST_DEFAULT_WAVELET="1"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "seed 0: steps 2" in out.stdout, out.stdout[-2000:]
    snaps = os.listdir(tmp_path / "models") if os.path.isdir(tmp_path / "models") else []
    assert all("packetssym8" in f for f in snaps)


def test_entry_point_with_block_norm_statistics(tmp_path):
    """--block-norm --calc-normalization: per-node statistics file + max-normalised features
    (reference wavelet_math.py:356-378, :436-447)."""
    cmd = [sys.executable, "-m", "src.audiofakedetect.train_classifier", "--log-dir", str(tmp_path),
           "--transform", "packets", "--wavelet", "sym5", "--num-of-scales", "256", "--log-scale",
           "--model", "modules", "--init-seeds", "0", "--synthetic", "--block-norm",
           "--calc-normalization", "--config", os.path.join(ROOT, "tests", "synthetic_config.py")]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, AFD_TEST_BLOCK_NORM="1"))  # the config's grid value wins
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "seed 0: steps 2" in out.stdout, out.stdout[-2000:]
    files = [f for f in os.listdir(tmp_path / "norms") if f.endswith("_mean_std_bn.pt")]
    assert len(files) == 1
    stats = torch.load(tmp_path / "norms" / files[0], map_location="cpu")
    assert len(stats) == 256 and set(next(iter(stats.values()))) == {"mean", "std"}
    assert all(k in stats for k in ("a" * 8, "d" + "a" * 7))


def test_trainer_train_eval_snapshot_roundtrip(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    from audiofakedetect.data_loader import SyntheticFrames
    from torch.utils.data import DataLoader

    torch.manual_seed(0)
    args, trainer, _ = bench.build("sym5-l8", 8, False, torch.device("cuda", 0))
    args.validation_interval = 1
    args.ckpt_every = 0
    loader = DataLoader(SyntheticFrames(16, 22050, num_labels=3), batch_size=8, drop_last=True)
    trainer.train_data_loader = trainer.val_data_loader = trainer.test_data_loader = loader
    trainer.snapshot_path = str(tmp_path / "snap.pt")
    trainer.train(1)
    assert trainer.step_total == 2 and len(trainer.loss_list) == 2
    assert all(torch.isfinite(torch.tensor(l[2])) for l in trainer.loss_list)
    acc, eer, cross_acc, cross_eer = trainer.test_results  # (test, test EER, cross-source, cross-source EER)
    assert 0.0 <= acc <= 1.0 and 0.0 <= eer <= 1.0 and cross_acc == 0.0 and cross_eer == 0.0
    assert len(trainer.validation_list) == 1  # validation_interval 1 validates epoch 0, as the reference does
    assert set(trainer.last_eval["per_label"]) <= {0, 1, 2}
    # class labels are bit-exact between two evaluations of the same snapshot
    before = trainer.last_eval["pred"].clone()
    trainer._save_snapshot(0)
    for p in trainer.model.parameters():
        p.data.add_(1.0)
    trainer.load_snapshot(trainer.snapshot_path)
    trainer.val_test_loop(loader, name="again")
    assert torch.equal(before, trainer.last_eval["pred"])


def test_integrated_gradients_match_cpu_restatement(tmp_path):
    """Trainer.integrated_grad (reference train_classifier.py:576-676) on the HIP model in eval
    mode against the same path integral on oracle/torch_ref.DCNNRef in float64; then the driver
    writes the three .npy files of the reference (:826-844)."""
    sys.path.insert(0, ROOT)
    import bench
    import numpy as np
    from audiofakedetect.data_loader import SyntheticFrames
    from audiofakedetect.integrated_gradients import integral_approximation, interpolate_images
    from oracle import torch_ref
    from torch.utils.data import DataLoader

    torch.manual_seed(1)
    args, trainer, _ = bench.build("sym5-l8", 4, False, torch.device("cuda", 0))
    net = trainer.model
    net.eval()
    ref = torch_ref.DCNNRef(args.input_dim, time_dim_add=1).double()
    ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in net.state_dict().items()})
    ref.eval()
    x = (0.1 * torch.randn(1, 1, 22050)).clamp_(-1, 1)
    with torch.no_grad():
        image = trainer._features(x.cuda())[0]
    baseline = torch.zeros_like(image)
    got = trainer.integrated_grad(baseline, image, 1, m_steps=12, batch_size=5)
    # reference computation on the CPU restatement
    alphas = torch.linspace(0.0, 1.0, 13, dtype=torch.float64)
    imgs = interpolate_images(baseline.cpu().double(), image.cpu().double(), alphas).requires_grad_(True)
    torch.softmax(ref(imgs), dim=-1)[:, 1].sum().backward()
    want = (image.cpu().double() - baseline.cpu().double()) * integral_approximation(imgs.grad)
    # the first path point is the all-zero baseline: every 2x2 pool window ties there and the
    # argmax convention routes that point's gradient differently per implementation, so the
    # agreement is bounded by its 1/(2 m_steps) quadrature weight, not by fp32 rounding
    err = (got.cpu().double() - want).abs().max().item()
    assert err <= 3e-2 * want.abs().max().item(), (err, want.abs().max().item())
    # completeness axiom (up to the quadrature error of 12 steps): sum of attributions ~ f(x) - f(0)
    with torch.no_grad():
        p = torch.softmax(net(torch.stack([baseline, image])), dim=-1)[:, 1]
    assert abs(got.sum().item() - (p[1] - p[0]).item()) <= 0.1 * abs((p[1] - p[0]).item()) + 1e-3

    args.log_dir = str(tmp_path)
    args.ig_times_per_target = 1
    args.target = None
    args.cross_sources = ["synthetic"]
    trainer.cross_loader_test = DataLoader(SyntheticFrames(8, 22050, num_labels=3), batch_size=4)
    trainer.integrated_gradients("ig_test", pbar=False)
    files = sorted(os.listdir(tmp_path / "plots"))
    assert [f.split("_target-01_")[1] for f in files] == ["integrated_gradients.npy", "last_image.npy", "mean_images.npy"]
    ig = np.load(tmp_path / "plots" / files[0])
    assert ig.shape == (256, image.shape[-1]) and np.isfinite(ig).all()


@pytest.mark.gpu
def test_native_loader_matches_the_dataset(tmp_path):
    """`NativeFrameLoader` (threaded WAV reader + `afd_pcm16_resample` on the GPU) against the dataset's own items
    (standard-library reader + `sinc_resample`, i.e. torchaudio.functional.resample's interpolator, reference
    data_loader.py:323-353): identical frames for files at the target rate, within 2e-6 for resampled ones
    (44.1 -> 22.05 kHz and 48 -> 22.05 kHz: 28 and 348 taps per phase); sharding and epoch shuffling as
    DistributedSampler."""
    import struct

    import numpy as np

    from audiofakedetect.data_loader import NativeFrameLoader, get_costum_dataset

    def write_wav(path, pcm, rate):
        data = pcm.astype("<i2").tobytes()
        fmt = struct.pack("<HHIIHH", 1, 1, rate, rate * 2, 2, 16)
        body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(data)) + data
        with open(path, "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", len(body)) + body)

    rng = np.random.default_rng(5)
    for rate, tol in ((22050, 0.0), (44100, 2e-6), (48000, 2e-6)):
        root = tmp_path / f"r{rate}"
        for name in ("A_real", "B_fake"):
            (root / name).mkdir(parents=True)
            for i in range(3):
                t = np.arange(rate * 6) / rate
                sig = 8000 * np.sin(2 * np.pi * (300 + 170 * i) * t) + 1500 * rng.standard_normal(t.size)
                write_wav(root / name / f"{i}.wav", sig, rate)
        ds = get_costum_dataset(data_path=str(root), save_path=str(root / "idx"), ds_type="train", seconds=1,
                                resample_rate=22050, limit=-1)
        loader = NativeFrameLoader(ds, 8, "cuda:0", shuffle=True, seed=3, drop_last=False, threads=4, prefetch=0)
        # prefetch=1: batches prepared one ahead by a background thread on a side stream ...
        ahead = NativeFrameLoader(ds, 8, "cuda", shuffle=True, seed=3, drop_last=False, threads=4, prefetch=1)
        assert loader.prefetch == 0 and ahead.prefetch == 1 and ahead.device.index is not None
        for got, want in zip(ahead, loader):  # ... are the batches of the in-thread form
            assert torch.equal(got["audio"], want["audio"]) and torch.equal(got["label"], want["label"])
        if rate == 22050:
            # the default policy: probe the consumer's time between batches in the caller's thread, then hand the rest
            # of the epoch (and later epochs) to the background thread iff that time is above AUTO_STEP_MS
            import time

            for pause, choice in ((0.0, 0), (0.03, 1)):
                auto = NativeFrameLoader(ds, 2, "cuda:0", shuffle=True, seed=3, drop_last=False, threads=4)
                plain = NativeFrameLoader(ds, 2, "cuda:0", shuffle=True, seed=3, drop_last=False, threads=4, prefetch=0)
                assert auto.prefetch == "auto" and len(auto) > auto.AUTO_PROBE + 2
                for _ in range(2):  # second epoch: the remembered choice from its first batch on
                    n = 0
                    for got, want in zip(auto, plain):
                        assert torch.equal(got["audio"], want["audio"]) and torch.equal(got["label"], want["label"])
                        time.sleep(pause)
                        n += 1
                    assert n == len(plain)
                assert auto._auto_choice == choice, (pause, auto.consumer_ms)
        it = iter(ahead)  # a consumer that stops early leaves no thread behind
        next(it)
        it.close()
        seen = 0
        order = loader._indices()
        for b, batch in enumerate(loader):
            sel = order[b * 8:(b + 1) * 8]
            assert batch["audio"].is_cuda and batch["audio"].shape[1:] == (1, 22050)
            for k, i in enumerate(sel):
                item = ds[int(i)]
                assert int(batch["label"][k]) == int(item["label"])
                err = (batch["audio"][k].cpu() - item["audio"]).abs().max().item()
                assert err <= tol, (rate, err)
                seen += 1
        assert seen == len(ds)
    # two ranks see disjoint halves; another epoch another order
    a = NativeFrameLoader(ds, 4, "cuda:0", shuffle=True, seed=1, rank=0, world=2)._indices()
    b = NativeFrameLoader(ds, 4, "cuda:0", shuffle=True, seed=1, rank=1, world=2)._indices()
    assert not set(a) & set(b) and len(a) == len(b) == len(ds) // 2
    l2 = NativeFrameLoader(ds, 4, "cuda:0", shuffle=True, seed=1)
    first = l2._indices().copy()
    l2.set_epoch(1)
    assert not np.array_equal(first, l2._indices())


def test_trainer_epoch_on_the_native_loader(tmp_path):
    """`create_data_loaders` with `native_loader` hands the trainer `NativeFrameLoader`s over a WAV dataset on disk
    (reference folder convention); one epoch of training + validation + test runs on them."""
    import struct

    import numpy as np

    sys.path.insert(0, ROOT)
    import bench
    from audiofakedetect.data_loader import NativeFrameLoader
    from audiofakedetect.train_classifier import create_data_loaders

    rng = np.random.default_rng(9)
    root = tmp_path / "data"
    for name in ("A_real", "B_fake"):
        (root / name).mkdir(parents=True)
        for i in range(2):
            pcm = (3000 * rng.standard_normal(22050 * 40)).astype("<i2")
            fmt = struct.pack("<HHIIHH", 1, 1, 22050, 44100, 2, 16)
            body = b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", pcm.nbytes) + pcm.tobytes()
            with open(root / name / f"{i}.wav", "wb") as f:
                f.write(b"RIFF" + struct.pack("<I", len(body)) + body)
    torch.manual_seed(0)
    args, trainer, _ = bench.build("sym5-l8", 8, False, torch.device("cuda", 0))
    args.update(data_path=str(root), save_path=str(tmp_path / "idx"), synthetic=False, native_loader=True,
                file_type="wav", limit_train=None, num_workers=4, seed=0, validation_interval=1, ckpt_every=0)
    train, val, test, _, _ = create_data_loaders(args)
    assert all(isinstance(l, NativeFrameLoader) for l in (train, val, test))
    assert len(train) == (2 * int(0.7 * 80)) // 8
    trainer.train_data_loader, trainer.val_data_loader, trainer.test_data_loader = train, val, test
    trainer.snapshot_path = str(tmp_path / "snap.pt")
    trainer.train(1)
    assert trainer.step_total == len(train) and all(np.isfinite(l[2]) for l in trainer.loss_list)
    assert 0.0 <= trainer.test_results[0] <= 1.0


def test_tensorboard_writer_is_closed_after_the_last_scalars(tmp_path, monkeypatch):
    """`--tensorboard`: the writer of an experiment is closed when the experiment ends (reference
    train_classifier.py:1364 closes it at the end of main), AFTER the test scalars were handed to it -- an open
    SummaryWriter keeps its last events in a queue that the interpreter's exit does not flush."""
    sys.path.insert(0, ROOT)
    from audiofakedetect import train_classifier as tc

    made = []

    class Recorder:
        def __init__(self):
            self.rows, self.closed_after = [], None

        def add_scalar(self, tag, v, step):
            assert self.closed_after is None, f"scalar {tag} after close()"
            self.rows.append(tag)

        def close(self):
            self.closed_after = len(self.rows)

    def fake_writer(args, model_name):
        made.append(Recorder())
        return made[-1]

    monkeypatch.setattr(tc, "make_writer", fake_writer)
    monkeypatch.setattr(sys, "argv", ["train_classifier", "--log-dir", str(tmp_path), "--transform", "packets",
                                      "--wavelet", "sym5", "--num-of-scales", "256", "--log-scale", "--model", "modules",
                                      "--init-seeds", "0", "1", "--synthetic", "--tensorboard",
                                      "--config", os.path.join(ROOT, "tests", "synthetic_config.py")])
    tc.main()
    assert len(made) == 2  # one writer per experiment (two seeds), none left open
    for w in made:
        assert w.closed_after is not None and w.closed_after == len(w.rows)
        assert "loss/train" in w.rows and "accuracy/test" in w.rows and "eer/test" in w.rows
