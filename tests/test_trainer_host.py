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
