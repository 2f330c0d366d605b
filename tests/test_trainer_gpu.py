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
           "--model", "modules", "--init-seeds", "0", "--synthetic",
           "--config", os.path.join(ROOT, "tests", "synthetic_config.py")]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "seed 0: steps 2" in out.stdout, out.stdout[-2000:]


def test_trainer_train_eval_snapshot_roundtrip(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    from audiofakedetect.data_loader import SyntheticFrames
    from torch.utils.data import DataLoader

    torch.manual_seed(0)
    args, trainer = bench.build("sym5-l8", 8, False, torch.device("cuda", 0))
    args.validation_interval = 1
    args.ckpt_every = 0
    loader = DataLoader(SyntheticFrames(16, 22050, num_labels=3), batch_size=8, drop_last=True)
    trainer.train_data_loader = trainer.val_data_loader = trainer.test_data_loader = loader
    trainer.snapshot_path = str(tmp_path / "snap.pt")
    trainer.train(1)
    assert trainer.step_total == 2 and len(trainer.loss_list) == 2
    assert all(torch.isfinite(torch.tensor(l[2])) for l in trainer.loss_list)
    acc, eer = trainer.test_results
    assert 0.0 <= acc <= 1.0 and 0.0 <= eer <= 1.0
    assert set(trainer.last_eval["per_label"]) <= {0, 1, 2}
    # class labels are bit-exact between two evaluations of the same snapshot
    before = trainer.last_eval["pred"].clone()
    trainer._save_snapshot(0)
    for p in trainer.model.parameters():
        p.data.add_(1.0)
    trainer.load_snapshot(trainer.snapshot_path)
    trainer.val_test_loop(loader, name="again")
    assert torch.equal(before, trainer.last_eval["pred"])
