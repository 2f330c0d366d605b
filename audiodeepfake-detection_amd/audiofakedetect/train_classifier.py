"""Trainer / entry point of the hot path (MI355X host side).

Mirrors the reference's ``src/audiofakedetect/train_classifier.py`` for the symbols on the
path: ``ddp_setup`` (:44-47), ``create_data_loaders`` samplers (:118-127), ``Trainer``
(``init_model`` :306-329, ``_run_batch`` :945-995, ``_run_epoch`` :887-912, ``train``
:1021-1046, ``_save_snapshot``/``load_snapshot`` :997-1019, a batched ``val_test_loop``
:365-497) and ``main`` (:1084-1368) including the ``--config`` ``get_config()`` protocol, so
``torchrun ... -m src.audiofakedetect.train_classifier <flags>`` (scripts/train.sh:33-68)
runs unchanged.

Data parallelism is one process per GPU over RCCL (``backend="nccl"`` is RCCL on ROCm):
parameters/optimizer state replicated, batches sharded by ``DistributedSampler``; per step
ONE all-reduce over the flat gradient arena of ``FusedAdam`` (the mean is folded into the
optimizer kernel) plus the packed BatchNorm statistics exchanges inside the model.  The
reference wraps the model in DistributedDataParallel twice; here the wrap is
``DataParallelRCCL`` and happens once.
"""

from __future__ import annotations

import argparse
import os
import sys
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from . import ops
from .data_loader import CustomDataset, NativeFrameLoader, get_costum_dataset
from .models import get_model, strip_ddp_prefix
from .utils import DotDict, add_default_parser_args, build_new_grid, get_input_dims, set_seed
from .wavelet_math import fuse_normalization, get_transforms


def ddp_setup() -> None:
    """Join the process group (RCCL) and bind this process to its GPU."""
    dist.init_process_group(backend="nccl")
    torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
    if os.environ.get("AFD_RCCL_DIRECT"):  # opt-in: the in-step collectives on the compute stream (ops.enable_direct_rccl)
        ops.enable_direct_rccl()


def is_lead(args: DotDict) -> bool:
    return (not args.ddp) or int(os.environ.get("RANK", "0")) == 0


class DataParallelRCCL(torch.nn.Module):
    """Replica wrapper: broadcasts rank 0's parameters and buffers at construction.

    Gradient averaging is done by the Trainer in one collective over the optimizer's flat
    gradient arena (``sync_gradients``), not by per-bucket autograd hooks.
    """

    def __init__(self, module: torch.nn.Module) -> None:
        super().__init__()
        self.module = module
        if dist.is_initialized() and dist.get_world_size() > 1:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, src=0)

    def forward(self, *a, **kw):
        return self.module(*a, **kw)


def sync_gradients(model: torch.nn.Module, optimizer) -> float:
    """Sum the gradients over ranks; returns the scale (1/world) still to be applied.

    With ``FusedAdam`` this is ONE all-reduce over the contiguous arena and the 1/world factor
    is applied inside the Adam kernel; otherwise gradients are reduced tensor by tensor and
    scaled here (returns 1.0).
    """
    if not ops.multi_rank():
        return 1.0
    world = dist.get_world_size()
    if isinstance(optimizer, ops.FusedAdam):
        work = getattr(optimizer, "_pending_allreduce", None)
        if work is not None:  # started by the end-of-backward hook (start_gradient_allreduce)
            optimizer._pending_allreduce = None
            work.wait()
            return 1.0 / world
        ops._end_of_backward.clear()  # a hook whose backward never reached the loss node (foreign loss function)
        ops.all_reduce_sum(optimizer.flat_grad)
        return 1.0 / world
    for p in model.parameters():
        if p.grad is not None:
            dist.all_reduce(p.grad)
            p.grad.mul_(1.0 / world)
    return 1.0


def _loader(args: DotDict, ds, train: bool):
    """Only the training split drops its ragged tail (sampler and loader); validation / test see every
    sample, the DistributedSampler padding a shard by wrapping around (reference :118-158)."""
    if args.native_loader and type(ds) is CustomDataset:  # the Detailed variant carries per-item fields the batch reader drops
        rank = dist.get_rank() if dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_initialized() else 1
        return NativeFrameLoader(ds, args.batch_size, torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0))),
                                 shuffle=train, seed=args.seed or 0, drop_last=train, rank=rank, world=world,
                                 threads=max(1, int(args.num_workers or 8)))
    sampler = None
    if args.ddp:
        sampler = DistributedSampler(ds, shuffle=train, seed=args.seed or 0, drop_last=train)
    workers = int(args.num_workers or 0)
    return DataLoader(ds, batch_size=args.batch_size, shuffle=(sampler is None and train), sampler=sampler,
                      drop_last=train, pin_memory=True, num_workers=workers,
                      persistent_workers=workers > 0)


class _Done:
    """Stand-in for a c10d work handle of a collective that was issued in stream order."""

    def wait(self) -> None:
        return None


def start_gradient_allreduce(optimizer) -> None:
    """Arrange for the arena all-reduce to be issued by the autograd engine itself when the coming backward
    pass ends (ops.at_end_of_backward): RCCL gets the collective the moment the last gradient kernel has been
    queued; `sync_gradients` picks the handle up."""
    if not (ops.multi_rank() and isinstance(optimizer, ops.FusedAdam)):
        return
    # a backward pass that raised never ran its end-of-backward callbacks: drop a hook it left queued (two
    # hooks would issue two all-reduces over the arena, of which only one is waited for and scaled) and
    # finish a collective it may have started
    ops._end_of_backward.clear()
    stale = getattr(optimizer, "_pending_allreduce", None)
    if stale is not None:
        optimizer._pending_allreduce = None
        stale.wait()

    def fire() -> None:
        if ops._direct_rccl:
            # on the compute stream, behind the last gradient kernel: nothing to wait for on the host
            ops.all_reduce_sum(optimizer.flat_grad)
            optimizer._pending_allreduce = _Done()
        else:
            ops.note_collective(optimizer.flat_grad)
            optimizer._pending_allreduce = dist.all_reduce(optimizer.flat_grad, async_op=True)

    ops.at_end_of_backward(fire)


def create_data_loaders(args: DotDict, limit: Optional[int] = None):
    """(train, val, test, cross val, cross test) loaders (reference :50-229).

    ``limit`` caps the synthetic splits only; real datasets are cut by ``limit_train`` / ``cross_limit``
    from the config, as in the reference."""
    synthetic = bool(args.synthetic) or args.data_path is None
    loaders = []
    for i, split in enumerate(("train", "val", "test")):
        lim = args.limit_train[i] if args.limit_train else (limit if synthetic else None)
        kw = dict(limit=lim) if lim is not None else {}
        ds = get_costum_dataset(data_path=args.data_path, ds_type=split, only_use=args.only_use,
                                save_path=args.save_path, file_type=args.file_type,
                                resample_rate=args.sample_rate, seconds=args.seconds,
                                synthetic=synthetic, **kw)
        loaders.append(_loader(args, ds, split == "train"))
    cross_val = cross_test = None
    if args.unknown_prefix is not None or args.cross_data_path is not None:
        if args.cross_data_path is None:
            raise NotImplementedError("--unknown-prefix needs cross_data_path in the config (reference :187-189)")
        for split in ("val", "test"):
            i = 1 if split == "val" else 2
            kw = dict(limit=args.cross_limit[i]) if args.cross_limit else {}
            ds = get_costum_dataset(data_path=args.cross_data_path, ds_type=split,
                                    only_test_folders=args.only_test_folders, only_use=args.cross_sources,
                                    save_path=args.save_path, file_type=args.file_type,
                                    resample_rate=args.sample_rate, seconds=args.seconds, **kw)
            if split == "val":
                cross_val = _loader(args, ds, False)
            else:
                cross_test = _loader(args, ds, False)
    return loaders[0], loaders[1], loaders[2], cross_val, cross_test


class Trainer:
    """Training / evaluation driver with the reference's method names."""

    def __init__(self, snapshot_path: str, args: DotDict, normalize, transforms,
                 test_data_loader, model, train_data_loader, val_data_loader,
                 cross_loader_val=None, cross_loader_test=None, optimizer=None, loss_fun=None,
                 writer=None) -> None:
        self.args = args
        if self.args.ddp:
            self.local_rank = int(os.environ["LOCAL_RANK"])
            self.global_rank = int(os.environ["RANK"])
            self.world_size = int(os.environ["WORLD_SIZE"])
        else:
            self.local_rank = self.global_rank = torch.cuda.current_device()
            self.world_size = 1
        self.model = model
        if model is not None:
            self._check_model_init()
        self.train_data_loader = train_data_loader
        self.val_data_loader = val_data_loader
        self.test_data_loader = test_data_loader
        self.cross_loader_val = cross_loader_val
        self.cross_loader_test = cross_loader_test
        self.optimizer = optimizer
        self.loss_fun = loss_fun
        self.epochs_run = 0
        self.snapshot_path = snapshot_path + ".pt"
        self.normalize = normalize
        self.transforms = transforms
        self.writer = writer
        # (x - mean) / std moves into the transform kernel's epilogue when it can
        if normalize is not None and transforms is not None:
            fuse_normalization(transforms, normalize)
        self.validation_list: list = []
        self.loss_list: list = []
        self.accuracy_list: list = []
        self.step_total = 0
        self.test_results: tuple = ()
        # device-side step statistics [loss, #correct]; fetched lazily (no per-step host sync)
        self.last_stats: Optional[torch.Tensor] = None
        self.sync_every_step = True

    # -- model placement ------------------------------------------------------------------
    def init_model(self, model) -> None:
        if model is None:
            raise RuntimeError("Model to initialize not given.")
        if isinstance(model, DataParallelRCCL):
            return
        if not isinstance(model, torch.nn.Module):
            raise RuntimeError("Given Model not of type torch.nn.Module.")
        model.to(self.local_rank, non_blocking=True)
        if self.args.ddp:
            self.model = DataParallelRCCL(model)

    def _check_model_init(self) -> None:
        if self.model is None:
            raise RuntimeError("Model not initialized.")
        self.init_model(self.model)
        if self.args.ddp and not isinstance(self.model, DataParallelRCCL):
            raise RuntimeError("Model not parallelized.")

    # -- the step -------------------------------------------------------------------------
    def _features(self, audio: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            feats, _ = self.transforms(audio)
            return self.normalize(feats)

    def _run_batch(self, e: int, batch: dict) -> None:
        """One training step (reference train_classifier.py:945-995)."""
        key = getattr(self.train_data_loader.dataset, "key", "audio") if self.train_data_loader is not None else "audio"
        audio = batch[key].to(self.local_rank, non_blocking=True)
        labels = (batch["label"].to(self.local_rank, non_blocking=True) != 0).type(torch.long)
        self.optimizer.zero_grad()
        feats = self._features(audio)
        out = self.model(feats)
        loss = self.loss_fun(out, labels)
        start_gradient_allreduce(self.optimizer)
        loss.backward()
        scale = sync_gradients(self.model, self.optimizer)
        if isinstance(self.optimizer, ops.FusedAdam):
            self.optimizer.step(grad_scale=scale)
        else:
            self.optimizer.step()
        self.step_total += 1
        if self.step_total == 2:
            # Everything alive after two steps -- modules, plan, scratch buffers, the library binding: ~270 000 tracked
            # objects -- is long-lived.  Python's first full collection over them is a 120-140 ms stall that otherwise lands
            # somewhere in the first hundred steps (tools/step_drift.py: 25 steps' worth at level 8); frozen, the
            # collector only ever walks what the steps themselves leave behind.
            import gc

            gc.collect()
            gc.freeze()
        stats = getattr(self.loss_fun, "last_stats", None)
        self.last_stats = stats
        if self.sync_every_step:
            # the reference reads loss.item() / acc.item() every step (:981-989)
            if stats is not None:
                lv, correct = stats.tolist()
                acc = correct / self.args.batch_size
            else:
                lv = loss.item()
                acc = (out.argmax(-1) == labels).sum().item() / self.args.batch_size
            self.loss_list.append([self.step_total, e, lv])
            self.accuracy_list.append([self.step_total, e, acc])
            self._scalars({"loss/train": lv, "accuracy/train": acc})

    def _scalars(self, values: dict) -> None:
        """The reference's TensorBoard scalars (:879-883, :936-943, :991-995) when a writer was given: same tags,
        the lead rank only, x = step_total.  (The reference's add_graph of the first step is not reproduced.)"""
        if self.writer is None or not is_lead(self.args):
            return
        for tag, v in values.items():
            self.writer.add_scalar(tag, float(v), self.step_total)

    def _run_epoch(self, epoch: int) -> None:
        sampler = getattr(self.train_data_loader, "sampler", None)
        if isinstance(sampler, DistributedSampler):
            sampler.set_epoch(epoch)
        elif hasattr(self.train_data_loader, "set_epoch"):
            self.train_data_loader.set_epoch(epoch)
        for batch in self.train_data_loader:
            self.model.train()
            self._run_batch(epoch, batch)

    # -- evaluation -----------------------------------------------------------------------
    @staticmethod
    def calculate_eer(y_true, y_score) -> float:
        """Equal error rate of a binary classifier output (reference :347-363).

        The reference feeds sklearn's ``roc_curve`` + ``brentq`` on a linear interpolation of
        the ROC; the same point is found here directly: sort by score, build the ROC, and
        intersect its polyline with tpr = 1 - fpr.
        """
        y_true = np.asarray(y_true).astype(bool).reshape(-1)
        y_score = np.asarray(y_score, dtype=np.float64).reshape(-1)
        pos = max(int(y_true.sum()), 1)
        neg = max(int((~y_true).sum()), 1)
        order = np.argsort(-y_score, kind="stable")
        ys, ss = y_true[order], y_score[order]
        last = np.r_[np.nonzero(np.diff(ss))[0], ss.size - 1]  # last index of each threshold
        tpr = np.r_[0.0, np.cumsum(ys)[last] / pos]
        fpr = np.r_[0.0, np.cumsum(~ys)[last] / neg]
        f = 1.0 - fpr - tpr  # decreasing along the curve; root = EER
        k = int(np.argmax(f <= 0))
        if f[k] == 0 or k == 0:
            return float(fpr[k])
        x0, x1, f0, f1 = fpr[k - 1], fpr[k], f[k - 1], f[k]
        return float(x0 + (x1 - x0) * f0 / (f0 - f1))

    @staticmethod
    def calculate_acc_label(count_dict_gathered: list, ok_dict_gathered: list, key: int) -> float:
        """Accuracy of one label over the per-rank gathered lists (reference :528-574)."""
        keys = set()
        for d in count_dict_gathered:
            keys.update(d.keys())
        for d in ok_dict_gathered:
            keys.update(d.keys())
        for d in list(count_dict_gathered) + list(ok_dict_gathered):
            keys &= set(d.keys())
        if key not in keys:
            raise KeyError(f"Key {key} does not exist in both dictionaries. Only available keys: {sorted(keys)}.")
        acc = sum(sum(d[key]) for d in ok_dict_gathered) / sum(d[key] for d in count_dict_gathered)
        if isinstance(acc, torch.Tensor):
            return acc.item()
        if isinstance(acc, float):
            return acc
        raise TypeError("Result should either be float or tensor.")

    @staticmethod
    def caculate_acc_dict(data_loader, common_keys, ok_dict_gathered: list,
                          count_dict_gathered: list) -> list:
        """[(label name, accuracy)] for every label (reference :499-526; name kept as is)."""
        return [(data_loader.dataset.get_label_name(k),
                 Trainer.calculate_acc_label(count_dict_gathered, ok_dict_gathered, k))
                for k in common_keys]

    def val_test_loop(self, data_loader, name: str = "", pbar: bool = False):
        """Batched evaluation (reference :365-497): argmax == (label != 0).

        The reference walks the batch sample by sample on the host (:420-429); here the
        per-label correct / total counts are bincounts on the device, summed over ranks with
        one all-reduce, and predictions / truths are all-gathered for the EER, which (like the
        reference, :479-481) is computed on the hard predictions.  Returns (accuracy, eer);
        ``self.last_eval`` keeps predictions, truths and the per-label accuracies.
        """
        self.model.eval()
        nlab = 64
        counts = torch.zeros((2, nlab), dtype=torch.float64, device=self.local_rank)
        preds, truth = [], []
        key = getattr(data_loader.dataset, "key", "audio")
        with torch.no_grad():
            for batch in data_loader:
                audio = batch[key].to(self.local_rank, non_blocking=True)
                raw = batch["label"].to(self.local_rank, non_blocking=True).to(torch.int64)
                labels = raw != 0
                out = self.model(self._features(audio))
                pred = torch.argmax(out, -1)
                ok = (pred == labels).to(torch.float64)
                counts[0] += torch.bincount(raw.clamp(0, nlab - 1), weights=ok, minlength=nlab)
                counts[1] += torch.bincount(raw.clamp(0, nlab - 1), minlength=nlab).to(torch.float64)
                preds.append(pred)
                truth.append(labels)
        preds_t = torch.cat(preds) if preds else torch.zeros(0, dtype=torch.int64, device=self.local_rank)
        truth_t = torch.cat(truth) if truth else torch.zeros(0, dtype=torch.bool, device=self.local_rank)
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(counts)
            world = dist.get_world_size()
            gp = [torch.empty_like(preds_t) for _ in range(world)]
            gt = [torch.empty_like(truth_t) for _ in range(world)]
            dist.all_gather(gp, preds_t)
            dist.all_gather(gt, truth_t)
            preds_t, truth_t = torch.cat(gp), torch.cat(gt)
        total = counts[1].sum().clamp(min=1)
        acc = (counts[0].sum() / total).item()
        per_label = {int(k): (counts[0, k] / counts[1, k]).item()
                     for k in torch.nonzero(counts[1]).flatten().tolist()}
        eer = 0.0
        if self.global_rank == 0 and preds_t.numel() and truth_t.any() and (~truth_t).any():
            eer = Trainer.calculate_eer(truth_t.cpu().numpy(), preds_t.cpu().numpy())
        self.last_eval = {"pred": preds_t, "truth": truth_t, "per_label": per_label, "name": name}
        return acc, eer

    # ---- integrated gradients (reference train_classifier.py:576-844) -------------------
    def integrated_grad(self, baseline: torch.Tensor, image: torch.Tensor, target_class_idx,
                        m_steps: int = 50, batch_size: int = 32) -> torch.Tensor:
        """Integrated gradients of one image along the straight path from ``baseline``:
        ``m_steps + 1`` path points in batches of ``batch_size``, trapezoidal integral, scaled
        by ``image - baseline`` (reference :576-627)."""
        from .integrated_gradients import integral_approximation

        alphas = torch.linspace(0.0, 1.0, m_steps + 1, device=image.device)
        grads = [self.one_batch(baseline, image, alphas[i:i + batch_size], target_class_idx)
                 for i in range(0, alphas.numel(), batch_size)]
        return (image - baseline) * integral_approximation(torch.cat(grads, dim=0))

    def one_batch(self, baseline, image, alpha_batch, target_class_idx):
        from .integrated_gradients import interpolate_images

        return self.compute_gradients(interpolate_images(baseline, image, alpha_batch), target_class_idx)

    def compute_gradients(self, images: torch.Tensor, target_class_idx) -> torch.Tensor:
        """d softmax(logits)[target] / d images for a batch of images (reference :658-676)."""
        images = images.detach().requires_grad_(True)
        probs = torch.softmax(self.model(images), dim=-1)[:, target_class_idx]
        probs.backward(torch.ones_like(probs))
        return images.grad

    def integrated_gradients(self, model_file: str = "ig", pbar: bool = True) -> None:
        """Mean attribution map over test frames, written as ``.npy`` (reference :678-844; the
        figures of the reference are drawn from these arrays by tooling outside this package)."""
        from .integrated_gradients import Mean

        plot_path = os.path.join(self.args.log_dir, "plots")
        os.makedirs(plot_path, exist_ok=True)
        loader = self.cross_loader_test if self.cross_loader_test is not None else self.test_data_loader
        times = self.args.ig_times_per_target if self.args.ig_times_per_target is not None else 2500
        target = self.args.target
        both = target is None
        try:
            target_value = 1 if both else int(target)
        except ValueError:
            target_value = 1
        counts = {0: 0, 1: 0}
        mean_ig, mean_img = Mean(), Mean()
        image = None
        self.model.zero_grad()

        def wanted(lbl: int) -> bool:
            if both:
                return counts[lbl] < times
            return lbl == target_value and counts[lbl] < times

        def done() -> bool:
            return all(c >= times for c in counts.values()) if both else counts[target_value] >= times

        for batch in loader:
            labels = (batch["label"].cuda(non_blocking=True) != 0).long()
            feats = self.normalize(self.transforms(batch["audio"].cuda(non_blocking=True))[0])
            baseline = torch.zeros_like(feats[0])
            for i in range(feats.shape[0]):
                lbl = int(labels[i])
                if not wanted(lbl):
                    continue
                image = feats[i]
                attributions = self.integrated_grad(baseline, image, lbl, m_steps=200)
                mean_ig.update(attributions.sum(dim=0).unsqueeze(0))
                mean_img.update(image)
                counts[lbl] += 1
                if done():
                    break
            if done():
                break
        if image is None:
            raise RuntimeError("integrated_gradients: the loader held no frame of the requested target")
        if is_lead(self.args):
            target_str = "01" if both else str(target_value)
            sources = "-".join(self.args.cross_sources or [])
            path = os.path.join(plot_path, f"{model_file.replace('/', '_')}_{sources}x{times}_target-{target_str}")
            np.save(path + "_integrated_gradients.npy", mean_ig.finalize().cpu().numpy())
            np.save(path + "_mean_images.npy", mean_img.finalize().squeeze().cpu().numpy())
            np.save(path + "_last_image.npy", image.squeeze().detach().cpu().numpy())

    def testing(self, only_unknown: bool = False):
        """(test acc, test EER, cross-source acc, cross-source EER) (reference :1055-1065, :846-885);
        ``only_unknown`` evaluates the cross-source loader alone."""
        self._check_model_init()
        acc = eer = cross_acc = cross_eer = 0.0
        if not only_unknown and self.test_data_loader is not None:
            acc, eer = self.val_test_loop(self.test_data_loader, name="test known")
        if self.cross_loader_test is not None:
            cross_acc, cross_eer = self.val_test_loop(self.cross_loader_test, name="test unknown")
        self.test_results = (acc, eer, cross_acc, cross_eer)
        self._scalars({"accuracy/test": acc, "eer/test": eer, "accuracy/cross_test": cross_acc, "eer/cross_test": cross_eer})
        return self.test_results

    def _run_validation(self, epoch: int) -> None:
        acc, eer = self.val_test_loop(self.val_data_loader, name="val known")
        entry = [self.step_total, epoch, acc, eer]
        if self.cross_loader_val is not None:
            entry += list(self.val_test_loop(self.cross_loader_val, name="val unknown"))
        self.validation_list.append(entry)
        cr = entry[4:6] if len(entry) >= 6 else [0, 0]
        self._scalars({"accuracy/validation": acc, "eer/validation": eer, "accuracy/cross_validation": cr[0],
                       "eer/cross_validation": cr[1], "epochs": epoch})

    # -- snapshots ------------------------------------------------------------------------
    def _save_snapshot(self, epoch: int) -> None:
        torch.save({"MODEL_STATE": self.model.state_dict(), "EPOCHS_RUN": epoch}, self.snapshot_path)
        print(f"Epoch {epoch + 1} | Training snapshot saved at {self.snapshot_path}")

    def load_snapshot(self, snapshot_path: str) -> None:
        snapshot = torch.load(snapshot_path, map_location=f"cuda:{self.local_rank}")
        target = self.model.module if isinstance(self.model, DataParallelRCCL) else self.model
        target.load_state_dict(strip_ddp_prefix(snapshot["MODEL_STATE"]))
        self.epochs_run = snapshot["EPOCHS_RUN"]

    def train(self, max_epochs: int) -> None:
        """Epoch loop with the reference's snapshot / validation cadence (:1021-1053)."""
        self._check_model_init()
        ckpt = self.args.ckpt_every or 0
        val = self.args.validation_interval or 0
        for epoch in range(self.epochs_run, max_epochs):
            self._run_epoch(epoch)
            if self.global_rank == 0 and ckpt and ((epoch > 0 and epoch % ckpt == 0) or (epoch == 0 and ckpt == 1)):
                self._save_snapshot(epoch)
            if self.val_data_loader is not None and val and \
                    ((epoch > 0 and epoch % val == 0) or (epoch == 0 and val == 1)):
                self._run_validation(epoch)
            if epoch == max_epochs - 1 and (self.test_data_loader is not None or self.cross_loader_test is not None):
                self.testing()


def loss_less_flag(args: DotDict) -> bool:
    """The sign-channel switch as the reference reads its string flag (:1167): everything but "False" is on.  One
    derivation for the model's input channels, the snapshot name and the writer directory."""
    return False if args.loss_less == "False" else True


def make_writer(args: DotDict, model_name: str):
    """`--tensorboard`: a SummaryWriter on the reference's directory layout (:1271-1291), lead rank only.  The
    tensorboard package is an optional dependency: without it the flag is refused loudly instead of being dropped."""
    if not args.tensorboard or not is_lead(args):
        return None
    try:
        from torch.utils.tensorboard import SummaryWriter
    except Exception as err:  # ImportError, or tensorboard's own import-time failures
        import warnings
        warnings.warn(f"--tensorboard given but torch.utils.tensorboard is not importable ({err}): "
                      "no scalars will be written; the loss / accuracy lists and the printed summary are unaffected",
                      RuntimeWarning, stacklevel=2)
        return None
    only_use = list(args.only_use or ["all"])
    known = only_use[1] if len(only_use) > 1 else only_use[-1]
    wav = f"{args.wavelet}/" if args.transform == "packets" else ""
    path = (f"{args.log_dir}/tensorboard/{model_name}/{args.transform}/{wav}{args.features}/"
            f"{args.batch_size}_{args.learning_rate}_{args.weight_decay}_{args.epochs}/{args.f_min}-{args.f_max}/"
            f"{args.num_of_scales}/signs{loss_less_flag(args)}/augc{args.aug_contrast}/augn{args.aug_noise}/"
            f"power{args.power}/{known}/{args.seed}")
    return SummaryWriter(path, max_queue=100)


def snapshot_name(args: DotDict, model) -> str:
    """Snapshot file stem exactly as the reference composes it (:1162, :1221-1267), so that `--only-testing` /
    `--only-ig` find files written by the reference or by earlier runs: `path_name = basename(data_prefix).split("_")`
    gives the leading field (`path_name[0]`) and the field after f_min-f_max (`path_name[3]`, the train ratio of
    the reference's prepared folders, e.g. `fake_22050_22050_0.7_fbmelgan`); the source is `only_use[1]`; the
    augmentation flags are written as given.  Where the reference would raise IndexError (a prefix with fewer
    than four fields, fewer than two sources) the field is left out / the last source is used."""
    tr = "stft" if args.transform == "stft" else "packets" + str(args.wavelet)
    # the reference asks the model for its name only for --model modules (:1197); LCNN / grid models are "customModel"
    name = model.get_name() if args.model == "modules" and hasattr(model, "get_name") else "customModel"
    path_name = str(args.data_prefix or "fake").rstrip("/").split("/")[-1].split("_")
    only_use = list(args.only_use or ["all"])
    src = only_use[1] if len(only_use) > 1 else only_use[-1]
    ratio = f"{path_name[3]}_" if len(path_name) > 3 else ""
    loss_less = loss_less_flag(args)
    return (f"{path_name[0]}_{tr}_{args.features}_{args.hop_length}_{args.sample_rate}_{args.window_size}_"
            f"{args.num_of_scales}_{int(args.f_min or 0)}-{int(args.f_max or 0)}_{ratio}{args.learning_rate}_"
            f"{args.weight_decay}_{args.batch_size}_{args.nclasses}_{args.epochs}e_{name}_"
            f"signs{loss_less}_augc{args.aug_contrast}_augn{args.aug_noise}_"
            f"power{args.power}_{src}_{args.seconds}secs_{args.seed}")


def _parse_args():
    parser = argparse.ArgumentParser(description="Train an audio deepfake classifier (MI355X)")
    return add_default_parser_args(parser).parse_args()


def main() -> None:
    """CLI / experiment loop (reference train_classifier.py:1084-1368)."""
    parsed = _parse_args()
    args = DotDict(vars(parsed))
    if args.num_workers is None:
        # the reference hard-codes 10 loader workers (:1106); here: what this process may use
        args.num_workers = 0 if args.synthetic else min(10, max(1, (os.cpu_count() or 2) - 1))
    if args.ddp:
        ddp_setup()
    if args.config:
        ns: dict = {}
        with open(args.config) as f:
            exec(compile(f.read(), args.config, "exec"), ns)  # the reference's plugin protocol
        config = ns["get_config"]()
    else:
        config = {}
    for k in ("data_path", "save_path", "only_use", "limit_train", "file_type", "seconds", "cross_data_path",
              "cross_limit", "only_test_folders"):
        args.setdefault(k, None)
    args.seconds = args.seconds or 1
    griderator = build_new_grid(config, random_seeds=args.random_seeds, seeds=args.init_seeds)
    num_exp = griderator.get_len()
    exp_results: dict = {}
    for _ in range(num_exp):
        args, _step = griderator.update_step(args)
        set_seed(args.seed)
        device = f"cuda:{int(os.environ.get('LOCAL_RANK', 0))}"
        transforms, normalize = get_transforms(args, args.features, device, args.calc_normalization,
                                               pbar=args.pbar, verbose=is_lead(args))
        args.input_dim = get_input_dims(args, transforms)
        in_channels = 2 if loss_less_flag(args) else 1
        model = get_model(args, args.model, args.nclasses, in_channels, is_lead(args))
        train_loader, val_loader, test_loader, cross_val, cross_test = create_data_loaders(
            args, limit=args.batch_size * (args.synthetic_steps or 8))
        model.to(device)
        optimizer = ops.FusedAdam(model.parameters(), lr=args.learning_rate, weight_decay=args.weight_decay)
        loss_fun = ops.CrossEntropyLoss()
        os.makedirs(os.path.join(args.log_dir, "models"), exist_ok=True)
        snap = os.path.join(args.log_dir, "models", snapshot_name(args, model))
        model_name = model.get_name() if args.model == "modules" and hasattr(model, "get_name") else "customModel"
        trainer = Trainer(snap, args, normalize, transforms, test_loader, model, train_loader,
                          val_loader, cross_val, cross_test, optimizer, loss_fun, make_writer(args, model_name))
        try:
            if args.only_testing:
                # evaluate an existing snapshot on the cross-source test set (reference :1313-1316)
                trainer._check_model_init()
                trainer.load_snapshot(trainer.snapshot_path)
                trainer.testing(only_unknown=cross_test is not None)
            elif args.only_ig:
                # attribution of an existing snapshot (reference :1317-1323)
                trainer._check_model_init()
                trainer.load_snapshot(trainer.snapshot_path)
                tag = (f"{args.transform}_{args.sample_rate}_{args.seconds}_{args.seed}_"
                       f"{(args.only_use or ['all'])[-1]}_{args.wavelet}_{args.power}_{loss_less_flag(args)}")
                trainer.integrated_gradients(tag)
            else:
                trainer.train(args.epochs)
        finally:
            # the reference closes its writer at the end of main (:1364); here one writer per experiment, closed with
            # it: the last scalars (test / cross-test accuracy and EER) sit in the writer's queue until then
            if trainer.writer is not None:
                trainer.writer.close()
                trainer.writer = None
        exp_results.setdefault(args.seed, []).append(trainer.test_results)
        if is_lead(args):
            last = trainer.loss_list[-1][2] if trainer.loss_list else float("nan")
            res = trainer.test_results
            print(f"seed {args.seed}: steps {trainer.step_total}, last loss {last:.6f}, "
                  f"test acc {res[0] if res else float('nan'):.4f}, eer {res[1] if res else float('nan'):.4f}")
    if is_lead(args) and exp_results:
        # per-seed summary (the reference's print_results, :1360): mean / std over the runs of each seed
        for seed, runs in exp_results.items():
            arr = np.asarray([r for r in runs if r], dtype=np.float64)
            if arr.size:
                print(f"results seed {seed}: mean {np.round(arr.mean(0), 4).tolist()} std {np.round(arr.std(0), 4).tolist()} "
                      f"over {arr.shape[0]} run(s) [acc, eer, cross acc, cross eer]")
    if args.ddp:
        ops.disable_direct_rccl()  # the library's own communicator (AFD_RCCL_DIRECT) goes before the process group
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
