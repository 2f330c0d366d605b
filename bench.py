"""Headline benchmark: 1 s @ 22 050 Hz frames/s of (wavelet-packet front end + DCNN train step).

Contract: ``python bench.py --gpus N --steps K --warmup W``.  For N > 1 the driver launches it as
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`` (one rank per GPU
over RCCL); run directly with ``--gpus N`` and no WORLD_SIZE in the environment it starts that same
launcher itself as a CHILD process, before anything touches the GPU, and relays the child's output
and exit code.  Rank 0 prints ONE JSON line.

A step is the reference's training step (train_classifier.py:945-995) on one synthetic batch that
is already resident in HBM: zero_grad -> WPT (+log, +normalise) -> DCNN forward -> cross entropy ->
backward -> gradient all-reduce (N > 1) -> Adam, plus the per-step loss/accuracy read-back the
reference performs.  Default workload = BASELINE.json configs[1]: packets-coif4 level 14 + DCNN,
batch 128 per GPU, fp32.  ``--workload haar-l14-frontend`` (BASELINE configs[3], B = 4096) and
``coif4-l14-frontend`` / ``sym5-l14-frontend`` time the front end alone (a step = one transform of
the batch); ``stft-lcnn-eval`` is configs[4]'s evaluation forward.

Extra objects on the JSON line:
  roofline      the kernel class with the largest share of the step, from HIP events around every launch
                (all kernels run on the caller's stream: a launch's duration is the kernel's own time).
                MFMA-bound classes: achieved = flops really issued on the matrix cores (tile padding
                included, Winograd = its 16 or 36 GEMMs) / summed launch time, frac = achieved / 157.3 TF/s;
                the layer's direct-form flops are reported beside it as algorithmic_TFLOPs.
                HBM-bound front ends: achieved = algorithmic bytes 4 (N + C P T) per frame / launch
                time over >= 20 timed launches.  traffic = HBM bytes per step of that class from the
                committed rocprofv3 --pmc summary (profiles/), algorithmic_bytes beside it.
  frontend_only the front-end-only workloads (Haar level 14 at B = 4096 = BASELINE configs[3]; coif4 / sym5 level 14 at
                B = 128 and 4096) measured in the same process after the timed region: ms per transform, GB/s,
                fraction of the HBM peak.  Default workload, single process only.
  secondary     the other BASELINE configurations (sym5 level 14, the level-8 models, STFT + DCNN, STFT + LCNN bf16
                evaluation) measured in the same process after the timed region: ms/step, frames/s, dominant kernel
                class and its fraction of the bounding roof.  Default workload, single process only.
  end_to_end    the same step fed by NativeFrameLoader from WAV files on disk (host-to-device copy included), after
                the timed region: shows that `value` survives a real input path.  Never `value` itself.
  cpu_baseline  oracle/torch_ref.py (the reference's algorithm on torch CPU: per-node packet
                recursion with Welford left on, full-width DCNN step) timed on this host's cores on a
                bounded sample of the same workload.
"""

from __future__ import annotations

import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "audiodeepfake-detection_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

METRIC = "1s@22050Hz frames/sec (WPT-coif4 + DCNN train step) at 1/2/4/8 MI355X"
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (spec)
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured float4 copy)

WORKLOADS = {
    # name: (transform, wavelet, num_of_scales, time_dim_add, kind, description)
    "coif4-l14": ("packets", "coif4", 16384, 0, "train", "packets-coif4 level-14 + DCNN train step"),
    "coif4-l8": ("packets", "coif4", 256, 0, "train", "packets-coif4 level-8 + DCNN train step"),
    "sym5-l8": ("packets", "sym5", 256, 1, "train", "packets-sym5 level-8 + DCNN train step"),
    "sym5-l14": ("packets", "sym5", 16384, 0, "train", "packets-sym5 level-14 + DCNN train step"),
    # the reference's DEFAULT wavelet (utils.py:84-89) at the level its launch scripts use (start_exps.sh:9)
    "sym8-l8": ("packets", "sym8", 256, 0, "train", "packets-sym8 level-8 + DCNN train step"),
    "stft": ("stft", "none", 256, 0, "train", "STFT(n_fft 511, hop 220) + DCNN train step"),
    # BASELINE configs[4]: evaluation (forward + argmax) of the LCNN head on STFT features
    "stft-lcnn-eval": ("stft", "none", 256, 0, "eval", "STFT(n_fft 511, hop 220) + LCNN eval forward (fp32)"),
    "stft-lcnn": ("stft", "none", 256, 0, "train", "STFT(n_fft 511, hop 220) + LCNN train step (fp32)"),
    # BASELINE configs[4] at its stated precision: matrix products of the evaluation forward on the bf16 cores
    "stft-lcnn-eval-bf16": ("stft", "none", 256, 0, "eval", "STFT(n_fft 511, hop 220) + LCNN eval forward (bf16 matrix products)"),
    # front end alone (BASELINE configs[3] is the Haar one at B = 4096)
    "haar-l14-frontend": ("packets", "haar", 16384, 0, "frontend", "packets-haar level-14 front end only"),
    "coif4-l14-frontend": ("packets", "coif4", 16384, 0, "frontend", "packets-coif4 level-14 front end only"),
    "sym5-l14-frontend": ("packets", "sym5", 16384, 0, "frontend", "packets-sym5 level-14 front end only"),
    "coif4-l8-frontend": ("packets", "coif4", 256, 0, "frontend", "packets-coif4 level-8 front end only"),
}
DEFAULT_BATCH = {"haar-l14-frontend": 4096}
MFMA_CLASSES = ("conv_winograd", "conv_wgrad", "conv_igemm", "conv1x1", "conv_wgrad_1x1", "stft", "lcnn_bf16")  # (conv_direct mixes vector-ALU and matrix kernels: never the dominant class)
# rocprof kernel names of each timing class (profiles/r*_pmc_traffic.json is keyed by kernel); every kernel the
# library launches in a step belongs to exactly one class, so that the classes add up to the step
CLASS_KERNELS = {
    "conv_igemm": ("conv3x3_kernel", "conv_igemm_kernel"),
    "conv1x1": ("conv1x1_kernel", "conv1x1_split_stats_kernel", "conv1x1_stats_reduce_kernel"),
    "conv_winograd": ("wino_conv_kernel", "wino44_conv_kernel", "wino_weights_kernel", "wino44_weights_kernel"),
    "conv_wgrad": ("wino44_wgrad_kernel", "wino44_wgrad_reduce_kernel", "wino44_wgrad_g_kernel", "wgrad3x3_kernel",
                   "wgrad3x3p_kernel", "wgrad_reduce_kernel", "conv_wgrad2_kernel", "conv_wgrad_kernel"),
    "conv_direct": ("dilconv_direct_kernel", "dilconv_wgrad_kernel", "dilconv_reduce_kernel", "dilmfma_conv_kernel",
                    "dilmfma_wgrad_kernel", "dilmfma_weights_kernel", "dilmfma_reduce1_kernel", "dilmfma_reduce2_kernel"),
    "conv_wgrad_1x1": ("conv1x1_wgrad_kernel", "conv1x1_fused_bwd_kernel", "conv1x1_fused_bwd_reduce_kernel"),
    "conv_first": ("conv1_pool_fwd_kernel", "conv1_pool_bwd_kernel", "conv1_bwd_reduce_kernel", "conv1_stats_reduce_kernel"),
    "batchnorm": ("bn_stats_kernel", "bn_apply_fwd_kernel", "bn_bwd_stats_kernel", "bn_bwd_apply_kernel",
                  "bn_finalize_kernel", "bn_bwd_means_kernel", "bn_fold_forward_kernel", "bn_fold_backward_weights_kernel",
                  "bn_fold_backward_affine_kernel", "bn_backward_coef_kernel", "wino_bnstats_reduce1_kernel",
                  "wino_bnstats_reduce2_kernel", "conv_border_sums_kernel", "conv_input_grad_sums_kernel",
                  "conv_weight_dot_kernel"),
    "elementwise": ("prelu_pool_fwd_kernel", "prelu_pool_bwd_kernel", "prelu_pool_bwd_compact_kernel",
                    "prelu_pool_bwd_compact4_kernel", "prelu_dropout_fwd_kernel", "prelu_dropout_bwd_kernel",
                    "dropout_permute_kernel", "dropout_permute4_kernel", "linear_mean_fwd_kernel",
                    "linear_mean_bwd_x_kernel", "linear_mean_bwd_w_kernel", "ce_kernel", "multi_gather_kernel",
                    "adam_kernel", "normalize_kernel", "normalize_channels_kernel", "transpose_kernel", "moments_kernel"),
    "wpt": ("wpt_fused_kernel", "wpt_haar14_kernel", "wpt3_top_kernel", "wpt4_deep_kernel"),
    "stft": ("stft_mfma_kernel",),
    "lcnn_bf16": ("lcnn_conv_nhwc_kernel", "lcnn_conv1_kernel", "lcnn_conv1_pool_kernel", "gemm_nt_bf16_kernel",
                  "gemm_nt_bf16_128_kernel", "conv_bf16_kernel", "lstm_step_bf16_kernel", "lstm_step_bf16_pair_kernel",
                  "blstm_layer_bf16_kernel"),
}


WAVELET_TAPS = {"haar": 2, "sym5": 10, "coif4": 24, "sym8": 16}


def wpt_direct_flops(wavelet: str, level: int, n: int = 22050) -> float:
    """Direct-form flops of one frame's packet transform: 2 L per filter output, outputs of
    level k = 2^k nodes of length n_k, n_k = floor((n_{k-1} + L - 2 + (n_{k-1} odd)) / 2)."""
    taps = WAVELET_TAPS[wavelet]
    outputs = 0
    for k in range(1, level + 1):
        n = (n + taps - 2 + (n & 1)) // 2
        outputs += (1 << k) * n
    return 2.0 * taps * outputs


def log(msg: str) -> None:
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def spawn_ranks(n: int, argv: list) -> int:
    """Start `n` ranks as a child `torch.distributed.run` and wait for it.

    Called before this process has made any HIP call (torch.cuda.device_count() does not
    initialise the GPU on this image), so no GPU-holding process is ever replaced; the child's
    stdout (rank 0's JSON line) and stderr go straight through."""
    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit(f"--gpus {n}: only {have} GPU(s) visible on this node")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL (see the task notes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    log("spawning: " + " ".join(cmd))
    return subprocess.call(cmd, env=env)


def make_args(workload: str, batch: int, ddp: bool):
    from audiofakedetect.utils import DotDict

    transform, wavelet, scales, add = WORKLOADS[workload][:4]
    return DotDict(
        transform=transform, wavelet=wavelet, num_of_scales=scales, features="none", log_scale=True,
        loss_less="False", power=2.0, hop_length=220, sample_rate=22050, seconds=1, mean=0.0,
        std=1.0, block_norm=False, log_dir="/tmp/afd_bench", data_path=None, only_use=None,
        batch_size=batch, ddp=ddp, ochannels1=64, ochannels2=64, ochannels3=96, ochannels4=128,
        ochannels5=32, kernel1=3, dropout_cnn=0.6, dropout_lstm=0.2, time_dim_add=add,
        learning_rate=4e-4, weight_decay=1e-3, synthetic=True,
    )


def build(workload: str, batch: int, ddp: bool, device):
    from audiofakedetect import ops
    from audiofakedetect.models import DCNN
    from audiofakedetect.train_classifier import Trainer
    from audiofakedetect.wavelet_math import fuse_normalization, get_transforms

    kind = WORKLOADS[workload][4]
    scales = WORKLOADS[workload][2]
    args = make_args(workload, batch, ddp)
    transforms, normalize = get_transforms(args, "none", str(device), False, verbose=False)
    with torch.no_grad():
        probe, _ = transforms(torch.zeros(1, 1, 22050, device=device))
    args.input_dim = [batch] + list(probe.shape[1:])
    if kind == "frontend":
        fuse_normalization(transforms, normalize)
        return args, None, transforms
    # flattened size of the dil_conv output: [time_dim, 64-24=40, P/8-24]
    p8 = args.input_dim[2] // 8
    args.flattend_size = (64 - 24) * (p8 - 24)
    if "lcnn" in workload:
        from audiofakedetect.lcnn import LCNN

        model = LCNN(classes=2, in_channels=1, lstm_channels=scales,
                     precision="bf16" if workload.endswith("bf16") else "fp32").to(device)
    else:
        model = DCNN(args).to(device)
    opt = ops.FusedAdam(model.parameters(), lr=args.learning_rate, weight_decay=args.weight_decay)
    trainer = Trainer("/tmp/afd_bench/snap", args, normalize, transforms, None, model, None, None,
                      None, None, opt, ops.CrossEntropyLoss(), None)
    return args, trainer, transforms


def synthetic_batch(batch: int, rank: int, device):
    g = torch.Generator().manual_seed(1234 + rank)
    audio = (0.1 * torch.randn(batch, 1, 22050, generator=g)).clamp_(-1.0, 1.0)
    # cross-generator style labels {0: real, 1: B_melgan, 2: C_hifigan}; the step binarises
    # them with `!= 0` as the reference does (train_classifier.py:955-957)
    labels = torch.randint(0, 3, (batch,), generator=g, dtype=torch.int64)
    return {"audio": audio.to(device), "label": labels.to(device)}


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


CPU_SHARE_PER_GPU = 16  # host cores of the GPU node per GPU (8 GPUs on 2 x 64 cores)


def host_cores() -> int:
    """Physical cores this process may run on."""
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        import psutil

        physical = psutil.cpu_count(logical=False) or allowed
    except Exception:  # noqa: BLE001 - psutil is optional
        physical = allowed
    return max(1, min(allowed, physical))


def cpu_baseline(workload: str, step_frames: int, fe_frames: int = 128, steps: int = 3, fe_batches: int = 10,
                 threads: int = 0, fe_budget_s: float = 30.0):
    """The reference's algorithm on torch CPU (oracle 'port') on a bounded sample of the workload, by the
    protocol of BASELINE.md section 3: timed front-end batches AT THE PROTOCOL'S B = 128 and 1 untimed + >= 3 timed
    train steps, medians; frames/s = 1 / (front end per frame + step per frame).

    Front end: batches of `fe_frames` (128) full frames through the per-node pad + conv1d recursion with the per-node
    Welford updates left on, as the reference runs it (wavelet_math.py:182-206), + log + normalise.  The 16 383 node
    updates of a level-14 transform are bound by per-op overhead, so a batch of 128 costs about what a batch of 1
    does (round 3 timed B = 1 and understated the host by that factor): up to `fe_batches` batches, at least 3, cut
    when `fe_budget_s` seconds are used so the default run stays within minutes; the count is on the line.
    Train workloads add the FULL-WIDTH DCNN step (forward, cross entropy, backward, Adam with coupled L2) at
    B = `step_frames`: one level-14 frame is 64 GFLOP of fp32 convolutions (~0.3 s on 16 threads, linear in the
    frames -- the convolutions are compute-bound on the host, so the per-frame time does not depend on B).
    Threads: `threads`, default one GPU's share of the node's physical cores (16 of 128 on the MI355X box -- the job
    is one of eight per node); `--cpu-threads 0` uses every physical core.
    """
    import statistics

    from oracle import torch_ref, wpt_oracle

    transform, wavelet, scales, add, kind = WORKLOADS[workload][:5]
    physical = host_cores()
    cores = physical if threads == 0 else max(1, min(threads, physical))
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(99)
    x = (0.1 * torch.randn(fe_frames, 1, 22050, generator=g)).clamp_(-1, 1)
    level = scales.bit_length() - 1

    def front_end(xx):
        if transform == "packets":
            feats, _ = torch_ref.packets_torch(xx, wpt_oracle.TAPS[wavelet], level, log_scale=True,
                                               compute_welford=True, per_node=True)
        else:
            feats = torch_ref.stft_torch(xx, 2 * scales - 1, 220, log_scale=True)
        return torch_ref.normalize_torch(feats, 0.0, 1.0)

    fe_times = []
    t_begin = time.perf_counter()
    while len(fe_times) < fe_batches:
        t0 = time.perf_counter()
        feats = front_end(x)
        fe_times.append(time.perf_counter() - t0)
        if len(fe_times) >= 3 and time.perf_counter() - t_begin > fe_budget_s:
            break
    t_fe = statistics.median(fe_times)
    log(f"cpu baseline: front end {t_fe:.2f} s per batch of {fe_frames} frames ({cores} threads, {len(fe_times)} batches)")
    sample = (f"{cpu_model()}, {cores} torch threads (one GPU's share of the host's {physical} physical cores): "
              f"front end B = {fe_frames} frames, {t_fe:.2f} s per batch (median of {len(fe_times)} timed batches; "
              f"per-node pad+conv1d recursion, Welford on)")
    per_frame = t_fe / fe_frames
    out = {"front_end_batch": fe_frames, "front_end_s_per_batch": t_fe, "front_end_batches_timed": len(fe_times)}
    if kind == "train" and "lcnn" not in workload:
        fc = feats[:step_frames].contiguous()
        del feats
        labels = torch.randint(0, 2, (step_frames,), generator=g)
        packets = fc.shape[2]
        net = torch_ref.DCNNRef(fc.shape, time_dim_add=add, flattend_size=40 * (packets // 8 - 24))
        opt = torch.optim.Adam(net.parameters(), lr=4e-4, weight_decay=1e-3)
        net.train()
        torch_ref.train_step_torch(net, opt, fc, labels)  # untimed: allocator / oneDNN primitive warm-up
        st_times = []
        for _ in range(steps):
            t1 = time.perf_counter()
            torch_ref.train_step_torch(net, opt, fc, labels)
            st_times.append(time.perf_counter() - t1)
        t_step = statistics.median(st_times)
        per_frame += t_step / step_frames
        sample += (f" + full-width DCNN forward/backward/Adam B = {step_frames} frames, {t_step:.2f} s per step "
                   f"(median of {steps} timed steps after 1 untimed); frames/s = 1 / (front end per frame + step per frame)")
        out.update({"step_batch": step_frames, "step_s": t_step})
    elif kind != "frontend":
        sample += "; model step not timed on the CPU for this workload"
    out.update({"value": 1.0 / per_frame, "unit": "frames/s", "cores": cores, "kind": "port", "sample": sample})
    return out


def end_to_end(trainer, batch_size: int, rank: int, device, steps: int, threads: int = 16):
    """The same train step fed by the input pipeline instead of one HBM-resident batch (reference
    train_classifier.py:131-159, :945-960: DataLoader -> batch[key].to(device) -> step): 16-bit PCM WAV files on
    local disk (page cache warm) -> `NativeFrameLoader` (threaded window reads into pinned memory, host-to-device
    copy, int16 -> f32 on the GPU) -> `_run_batch`.  Returns frames/s over `steps` steps, copies included."""
    import shutil
    import tempfile
    import wave

    import numpy as np

    from audiofakedetect.data_loader import NativeFrameLoader, get_costum_dataset

    root = tempfile.mkdtemp(prefix=f"afd_bench_wav_{rank}_")
    try:
        rng = np.random.default_rng(1234 + rank)
        secs, files = 40, 40  # 2 x 40 files x 40 s: 2 240 one-second frames in the 70 % train split (epochs of 17 batches)
        for name in ("A_real", "B_fake"):
            os.makedirs(os.path.join(root, name))
            for i in range(files):
                pcm = np.clip(rng.standard_normal(22050 * secs) * 3276.8, -32768, 32767).astype(np.int16)
                with wave.open(os.path.join(root, name, f"{i:04d}.wav"), "wb") as f:
                    f.setnchannels(1)
                    f.setsampwidth(2)
                    f.setframerate(22050)
                    f.writeframes(pcm.tobytes())
        ds = get_costum_dataset(data_path=root, save_path=os.path.join(root, "index"), ds_type="train", seconds=1,
                                resample_rate=22050, limit=-1)
        loader = NativeFrameLoader(ds, batch_size, device, shuffle=True, seed=0, drop_last=True, threads=threads)
        done, t0 = 0, None
        epoch = 0
        # untimed: pinned buffers, first file reads, the loader's own probe of the step time (AUTO_PROBE + 1 batches in the
        # caller's thread) and the hand-over to its background thread -- the whole first epoch and two batches of the second
        untimed = len(loader) + 2
        while done < steps + untimed:
            loader.set_epoch(epoch)
            epoch += 1
            for batch in loader:
                if done == untimed:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                trainer._run_batch(0, batch)
                done += 1
                if done >= steps + untimed:
                    # (the clock stops before the loop is left: abandoning an epoch half way makes the loader wait for its
                    # background thread to notice, up to 50 ms that no step of a full epoch pays)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    break
        return {"value": batch_size * steps / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
                "frames_on_disk": len(ds), "reader_threads": threads,
                "loader_prefetch": loader._auto_choice, "loader_measured_step_ms": loader.consumer_ms,
                "path": "16-bit PCM WAV files (page cache) -> afd_wav_read_windows -> pinned int16 -> H2D -> "
                        "afd_pcm16_resample -> the same train step"}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def frontend_lines(device, _native, launches: int = 20):
    """The front-end-only workloads (BASELINE configs[3] and the level-14 transforms of configs[1] / [2]) measured in
    this same process, so that the driver's own run of the default command records them: per workload the mean launch
    time of the transform (HIP events, all launches of the `wpt` class) over `launches` calls on resident frames,
    algorithmic bytes 4 (N + C P T) per frame, and the fraction of the 8 TB/s HBM peak."""
    from audiofakedetect.wavelet_math import fuse_normalization, get_transforms

    out = []
    for name, batch in (("haar-l14-frontend", 4096), ("coif4-l14-frontend", 128), ("coif4-l14-frontend", 4096),
                        ("sym5-l14-frontend", 128), ("sym5-l14-frontend", 4096)):
        args = make_args(name, batch, False)
        transforms, normalize = get_transforms(args, "none", str(device), False, verbose=False)
        fuse_normalization(transforms, normalize)
        x = synthetic_batch(batch, 0, device)["audio"]
        with torch.no_grad():
            for _ in range(3):
                transforms(x)
            torch.cuda.synchronize()
            _native.timing_reset()
            _native.timing_enable(True)
            for _ in range(launches):
                transforms(x)
            torch.cuda.synchronize()
            _native.timing_enable(False)
        k = _native.timing_collect("wpt")
        _native.timing_reset()
        ms = k["total_ms"] / launches
        gbs = k["work"] / (k["total_ms"] * 1e-3) / 1e9
        flops = wpt_direct_flops(WORKLOADS[name][1], WORKLOADS[name][2].bit_length() - 1)
        out.append({"workload": WORKLOADS[name][5], "batch": batch, "ms_per_transform": ms,
                    "frames_per_s": batch / (ms * 1e-3), "achieved_GBps": gbs, "frac_of_hbm_peak": gbs / PEAK_HBM_GBS,
                    "algorithmic_bytes_per_frame": k["work"] / launches / batch,
                    # the other roof: direct-form filter arithmetic over the f32 peak (vector FMA and MFMA alike)
                    "direct_form_flops_per_frame": flops,
                    "frac_of_f32_peak": flops * batch / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS})
        del x, transforms
        torch.cuda.empty_cache()
    return out


def secondary_lines(device, _native, rank: int, steps: int = 20, warmup: int = 5, cpu_threads: int = 16):
    """The other BASELINE configurations measured in this same process after the timed region, so that the driver's
    own run of the default command records them: configs[2] per GPU (packets-sym5 level 14), the level-8 models the
    reference ships (coif4, sym5) and its default wavelet sym8, configs[0] (STFT + DCNN) and configs[4] (STFT + LCNN evaluation, bf16 matrix
    products).  Per workload: `steps` steps after `warmup` untimed ones, wall clock between device synchronisations;
    then two instrumented steps for the dominant kernel class and its fraction of the roof that bounds it.
    Never `value`."""
    import gc

    out = []
    for name in ("sym5-l14", "coif4-l8", "sym5-l8", "sym8-l8", "stft", "stft-lcnn-eval-bf16"):
        kind = WORKLOADS[name][4]
        try:
            torch.manual_seed(0)
            args, trainer, _ = build(name, 128, False, device)
            batch = synthetic_batch(128, rank, device)
            correct = torch.zeros((), dtype=torch.float64, device=device)
            if kind == "eval":
                trainer.model.eval()
            else:
                trainer.model.train()

            def step():
                if kind == "train":
                    trainer._run_batch(0, batch)
                else:
                    with torch.no_grad():
                        o = trainer.model(trainer._features(batch["audio"]))
                        correct.add_((o.argmax(-1) == (batch["label"] != 0)).sum())

            for _ in range(warmup):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / steps
            _native.timing_reset()
            _native.timing_enable(True)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            _native.timing_enable(False)
            kernels = collect(_native)
            _native.timing_reset()
            cand = [k for k in kernels if k == "wpt" or k in MFMA_CLASSES]
            dom = max(cand, key=lambda k: kernels[k]["total_ms"])
            r = roofline_of(dom, kernels[dom], 2, load_pmc(name, 128))
            entry = {"workload": WORKLOADS[name][5], "batch": 128, "steps": steps, "warmup": warmup, "ms_per_step": ms,
                     "frames_per_s": 128 / (ms * 1e-3), "dtype": "bf16" if name.endswith("bf16") else "f32",
                     "dominant_class": dom, "dominant_class_ms_per_step": kernels[dom]["total_ms"] / 2,
                     "bound": r["bound"], "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"],
                     "frac": r["frac"], "traffic": r["traffic"], "algorithmic_bytes": r["algorithmic_bytes"],
                     "classes_ms_per_step": {k: round(v["total_ms"] / 2, 4) for k, v in kernels.items()}}
            if kind == "eval":
                entry.update(eval_quality(trainer, rank, device))
                entry.update(eval_large_batch(trainer, rank, device, _native))
            out.append(entry)
            del trainer, batch, args
        except Exception as exc:  # noqa: BLE001 - the headline figure does not depend on this leg
            out.append({"workload": WORKLOADS[name][5], "error": f"{type(exc).__name__}: {exc}"})
        gc.collect()
        torch.cuda.empty_cache()
    # BASELINE configs[0] names a CPU path of its own ("STFT + DCNN, batch=128, CPU reference path"): the torch-CPU
    # restatement of that configuration at the protocol's B = 128 (front end and full train step), beside its GPU figure
    for entry in out:
        if entry.get("workload") == WORKLOADS["stft"][5] and "error" not in entry and rank == 0:
            try:
                entry["cpu_baseline"] = cpu_baseline("stft", 128, 128, steps=3, fe_batches=5, threads=cpu_threads)
            except Exception as exc:  # noqa: BLE001
                entry["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


class _Batches(list):
    """A list of resident batches that looks like a DataLoader to `Trainer.val_test_loop`."""

    class _DS:
        key = "audio"

    dataset = _DS()


def eval_quality(trainer, rank: int, device, batches: int = 8) -> dict:
    """configs[4]'s deliverable: accuracy and EER of the evaluation loop (reference train_classifier.py:365-497,
    :347-363) on synthetic cross-generator labels {0: A_real, 1: B_melgan, 2: C_hifigan}, binarised `!= 0` as the
    reference does, EER on the hard predictions as the reference computes it.  Random-init weights and noise frames:
    the figures are chance level by construction -- they show that the path (features, forward, argmax, per-label
    counts, EER) runs at the benchmark precision, not that the model separates anything."""
    data = _Batches(synthetic_batch(128, rank + 100 + i, device) for i in range(batches))
    acc, eer = trainer.val_test_loop(data, name="bench")
    names = {0: "A_real", 1: "B_melgan", 2: "C_hifigan"}
    return {"eval_frames": 128 * batches, "accuracy": acc, "eer": eer,
            "per_label_accuracy": {names.get(k, str(k)): v for k, v in trainer.last_eval["per_label"].items()},
            "eval_note": "synthetic noise frames, random-init weights, labels drawn from {A_real, B_melgan, C_hifigan}: "
                         "chance level by construction"}


def eval_large_batch(trainer, rank: int, device, _native, batch: int = 1024, steps: int = 10) -> dict:
    """The same evaluation forward at the batch a GPU wants (B = 1024): ms per batch, frames/s and the bf16 class's
    fraction of the bf16 peak."""
    big = synthetic_batch(batch, rank, device)
    trainer.model.eval()

    def step():
        with torch.no_grad():
            return trainer.model(trainer._features(big["audio"])).argmax(-1)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    _native.timing_reset()
    _native.timing_enable(True)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    _native.timing_enable(False)
    kernels = collect(_native)
    _native.timing_reset()
    out = {"batch": batch, "ms_per_step": ms, "frames_per_s": batch / (ms * 1e-3),
           "classes_ms_per_step": {k: round(v["total_ms"] / 2, 4) for k, v in kernels.items()}}
    if "lcnn_bf16" in kernels:
        k = kernels["lcnn_bf16"]
        out["lcnn_bf16_issued_TFLOPs"] = k["issued"] / (k["total_ms"] * 1e-3) / 1e12
        out["lcnn_bf16_frac_of_bf16_peak"] = out["lcnn_bf16_issued_TFLOPs"] / PEAK_BF16_MFMA_TFLOPS
    return {"large_batch": out}


def load_pmc(workload: str, batch: int):
    """Newest committed rocprofv3 --pmc summary for this workload/batch, or None."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
        try:
            with open(path) as fh:
                pmc = json.load(fh)
        except (OSError, ValueError):
            continue
        if pmc.get("workload") == workload and pmc.get("batch") == batch and "steps_in_trace" in pmc:
            pmc["_path"] = os.path.relpath(path, ROOT)
            return pmc
    return None


def class_traffic(pmc, cls: str):
    """HBM bytes per step of one timing class from the PMC summary (fetch corrected per the guide)."""
    if not pmc:
        return None
    tot = 0.0
    seen = False
    for nm in CLASS_KERNELS.get(cls, ()):
        kk = pmc["kernels"].get(nm)
        if kk:
            seen = True
            tot += kk.get("fetch_bytes_total", 0.0) + kk.get("write_bytes_total", 0.0)
    return tot / pmc["steps_in_trace"] if seen else None


def collect(_native) -> dict:
    out = {}
    for name in _native.KERNEL_CLASSES:
        c = _native.timing_collect(name)
        if c["launches"]:
            c["avg_ms"] = c["total_ms"] / c["launches"]
            out[name] = c
    return out


def roofline_of(cls: str, k: dict, steps_timed: int, pmc) -> dict:
    """The roofline object of one kernel class from its summed launch figures."""
    sec = k["total_ms"] * 1e-3
    if cls == "wpt" or (cls == "stft" and not k["issued"]):
        ach = k["work"] / sec / 1e9
        r = {"kernel": cls, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
             "frac": ach / PEAK_HBM_GBS}
        algo_bytes = k["work"] / steps_timed
    else:
        ach = k["issued"] / sec / 1e12
        peak = PEAK_BF16_MFMA_TFLOPS if cls == "lcnn_bf16" else PEAK_F32_MFMA_TFLOPS
        r = {"kernel": cls, "bound": "mfma", "achieved": ach, "peak": peak,
             "unit": "TFLOP/s", "frac": ach / peak,
             "algorithmic_TFLOPs": k["work"] / sec / 1e12,
             "note": "achieved = flops issued on the matrix cores (tile padding included; Winograd F(2x2,3x3) "
                     "= 16 GEMMs per 2x2 tile, F(4x4,3x3) = 36 per 4x4 tile) / summed launch time; algorithmic_TFLOPs = the layers' "
                     "direct-form flops over the same time"}
        algo_bytes = k["bytes"] / steps_timed
        if cls == "stft":  # the STFT launch reports its algorithmic BYTES as work and its folded matrix flops as issued
            algo_bytes = k["work"] / steps_timed
            r["algorithmic_TFLOPs"] = None
    r["traffic"] = class_traffic(pmc, cls)
    r["algorithmic_bytes"] = algo_bytes
    r["traffic_unit"] = "HBM bytes per step, all launches of the class"
    if pmc:
        r["traffic_source"] = pmc["_path"] + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"
    r["launches_per_step"] = k["launches"] / steps_timed
    r["avg_launch_ms"] = k["avg_ms"]
    return r


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU (default 128; 4096 for haar-l14-frontend)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="coif4-l14")
    ap.add_argument("--cpu-frames", type=int, default=8,
                    help="CPU baseline: frames per DCNN train step on the host (0 = skip the whole CPU leg)")
    ap.add_argument("--cpu-fe-frames", type=int, default=128, help="CPU baseline: frames per front-end batch (protocol: 128)")
    ap.add_argument("--cpu-threads", type=int, default=CPU_SHARE_PER_GPU,
                    help="torch threads of the CPU baseline (0 = every physical core)")
    ap.add_argument("--no-frontends", dest="frontends", action="store_false",
                    help="skip the front-end-only measurements appended to the default workload's line")
    ap.add_argument("--no-secondary", dest="secondary", action="store_false",
                    help="skip the other BASELINE configurations appended to the default workload's line")
    ap.add_argument("--e2e-steps", type=int, default=10,
                    help="steps of the end-to-end leg (WAV files -> loader -> H2D -> train step; 0 = skip)")
    ap.add_argument("--cpu-only", action="store_true", help="only run the CPU baseline leg")
    ap.add_argument("--spawn", action="store_true",
                    help="start the ranks through a child torch.distributed.run even for --gpus 1")
    a = ap.parse_args()
    batch_size = a.batch if a.batch is not None else DEFAULT_BATCH.get(a.workload, 128)
    kind = WORKLOADS[a.workload][4]

    if a.cpu_only:
        print(json.dumps(cpu_baseline(a.workload, max(1, a.cpu_frames), a.cpu_fe_frames, threads=a.cpu_threads)), flush=True)
        return
    launched = "WORLD_SIZE" in os.environ  # under torch.distributed.run (the driver's N > 1 form)
    if not launched and (a.gpus > 1 or a.spawn):
        argv = [x for x in sys.argv[1:] if x != "--spawn"]
        sys.exit(spawn_ranks(a.gpus, argv))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # a process group whenever the ranks come from the launcher (also for one rank: the RCCL path --
    # replica broadcast, SyncBN statistics, gradient all-reduce -- is then what runs); AFD_FORCE_DDP=1
    # does the same for a plain single-process run
    ddp = launched or bool(os.environ.get("AFD_FORCE_DDP"))
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("LOCAL_RANK", "0")
        dist.init_process_group(backend="nccl", device_id=device)

    from audiofakedetect import _native

    _native.load()
    torch.manual_seed(0)
    args, trainer, transforms = build(a.workload, batch_size, ddp, device)
    batch = synthetic_batch(batch_size, rank, device)
    if trainer is not None:
        trainer.model.train()

    def sync():
        if ddp:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    log(f"built {a.workload}: features {args.input_dim}, batch/GPU {batch_size}, world {world}")
    correct = torch.zeros((), dtype=torch.float64, device=device)

    def step():
        if kind == "train":
            trainer._run_batch(0, batch)
        elif kind == "eval":
            # evaluation step (reference val_test_loop, train_classifier.py:365-497): features,
            # forward, argmax, compare with the binarised label; counts stay on the device
            with torch.no_grad():
                out = trainer.model(trainer._features(batch["audio"]))
                correct.add_((out.argmax(-1) == (batch["label"] != 0)).sum())
        else:
            with torch.no_grad():
                transforms(batch["audio"])

    if kind == "eval":
        trainer.model.eval()
    for i in range(a.warmup):
        step()
        torch.cuda.synchronize()
        if kind != "frontend":
            log(f"warmup step {i} done")
    sync()
    from audiofakedetect import ops as _ops

    _ops.collective_counters(reset=True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    coll = _ops.collective_counters(reset=True)
    if ddp:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    step_ms = 1e3 * elapsed / a.steps
    log(f"timed {a.steps} steps: {step_ms:.3f} ms/step")

    # ---- per-kernel-class timing (HIP events around every launch, on the stream it is issued on), after the
    # timed region: three steps as they are timed above (>= 20 for the front-end workloads, whose class is one or
    # two launches per step).  Every kernel runs on the caller's stream, so a launch's duration is the kernel's own.
    timed_steps = max(20, a.steps) if kind == "frontend" else 3
    _native.timing_reset()
    _native.timing_enable(True)
    for _ in range(timed_steps):
        step()
    torch.cuda.synchronize()
    _native.timing_enable(False)
    kernels = collect(_native)
    _native.timing_reset()

    pmc = load_pmc(a.workload, batch_size)
    roofline = None
    if kernels:
        cand = [k for k in kernels if k == "wpt" or k in MFMA_CLASSES]
        if kind == "frontend":
            cand = [k for k in cand if k in ("wpt", "stft")]
        dom = max(cand, key=lambda k: kernels[k]["total_ms"])
        roofline = roofline_of(dom, kernels[dom], timed_steps, pmc)
        roofline["share_of_step"] = kernels[dom]["total_ms"] / timed_steps / step_ms
        roofline["timing"] = (f"HIP events around every launch of the class in {timed_steps} step(s) run as the "
                              "timed steps are (one stream: a launch's duration is the kernel's own time)")
    classes = {}
    for name, k in kernels.items():
        c = {"launches_per_step": k["launches"] / timed_steps, "ms_per_step": k["total_ms"] / timed_steps,
             "algorithmic_bytes_per_step": (k["work"] if name in ("wpt", "stft") and not k["bytes"] else k["bytes"]) / timed_steps,
             "hbm_bytes_per_step_pmc": class_traffic(pmc, name)}
        if k["issued"]:
            c["issued_TFLOPs"] = k["issued"] / (k["total_ms"] * 1e-3) / 1e12
            c["algorithmic_TFLOPs"] = None if name == "stft" else k["work"] / (k["total_ms"] * 1e-3) / 1e12
            c["mfma_frac"] = c["issued_TFLOPs"] / (PEAK_BF16_MFMA_TFLOPS if name == "lcnn_bf16" else PEAK_F32_MFMA_TFLOPS)
        else:
            c["achieved_GBps"] = c["algorithmic_bytes_per_step"] / (c["ms_per_step"] * 1e-3) / 1e9 if c["ms_per_step"] else None
        classes[name] = c
    classes_sum = sum(c["ms_per_step"] for c in classes.values())
    frontend = None
    if "wpt" in kernels:
        k = kernels["wpt"]
        gbs = k["work"] / (k["total_ms"] * 1e-3) / 1e9
        frontend = {"kernel": "wpt", "launches_timed": k["launches"], "ms_per_step": k["total_ms"] / timed_steps,
                    "achieved_GBps": gbs, "frac_of_hbm_peak": gbs / PEAK_HBM_GBS,
                    "algorithmic_bytes_per_frame": k["work"] / timed_steps / batch_size,
                    "hbm_bytes_per_step_pmc": class_traffic(pmc, "wpt")}

    # (before the other legs build their models: the process is then in the state a trainer is in -- its objects frozen
    # out of the garbage collector after the second step, nothing else allocated since)
    e2e = None
    # (single-process runs only: with several ranks a failure of this leg on one rank would leave the others
    # waiting in a collective)
    if kind == "train" and a.e2e_steps > 0 and world == 1:
        try:
            e2e = end_to_end(trainer, batch_size, rank, device, a.e2e_steps)
            log(f"end to end (WAV files -> loader -> H2D -> step): {e2e['ms_per_step']:.3f} ms/step")
        except Exception as exc:  # noqa: BLE001 - the headline figure does not depend on this leg
            e2e = {"error": f"{type(exc).__name__}: {exc}"}
    fe_lines = None
    if a.workload == "coif4-l14" and world == 1 and a.frontends:
        try:
            fe_lines = frontend_lines(device, _native)
            log("front ends: " + "; ".join(f"{f['workload']} B={f['batch']}: {f['ms_per_transform']:.4f} ms, "
                                           f"{f['frac_of_hbm_peak']:.3f}" for f in fe_lines))
        except Exception as exc:  # noqa: BLE001 - the headline figure does not depend on this leg
            fe_lines = [{"error": f"{type(exc).__name__}: {exc}"}]
    secondary = None
    if a.workload == "coif4-l14" and world == 1 and a.secondary:
        secondary = secondary_lines(device, _native, rank, cpu_threads=a.cpu_threads)
        log("secondary: " + "; ".join(f"{f['workload']}: {f['ms_per_step']:.3f} ms" if "ms_per_step" in f else
                                      f"{f['workload']}: {f['error']}" for f in secondary))
    cpu = None
    log(f"kernel classes (ms/step): { {k: round(v['ms_per_step'], 3) for k, v in classes.items()} }")
    if rank == 0 and world == 1 and a.cpu_frames > 0 and kind != "eval":
        cpu = cpu_baseline(a.workload, a.cpu_frames, a.cpu_fe_frames, threads=a.cpu_threads)
        log(f"cpu baseline: {cpu['value']:.4f} frames/s")

    devices = [torch.cuda.get_device_name(device)]
    rccl = None
    if ddp:
        gathered = [None] * world
        dist.all_gather_object(gathered, f"rank {rank}: cuda:{local_rank} {devices[0]}")
        devices = gathered
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001 - version query only
            rccl = "unknown"
    if rank == 0:
        loss = trainer.loss_list[-1][2] if (trainer is not None and trainer.loss_list) else None
        quality = eval_quality(trainer, rank, device) if kind == "eval" and world == 1 else None
        line = {
            "metric": METRIC, "value": world * batch_size * a.steps / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": step_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if a.workload.endswith("bf16") else "f32",
            "data": "synthetic",
            "config": {"workload": WORKLOADS[a.workload][5], "batch_per_gpu": batch_size,
                       "global_batch": batch_size * world, "frame": "1s@22050Hz mono f32",
                       "features": list(args.input_dim[1:]), "flattend_size": args.get("flattend_size"),
                       "optimizer": "Adam lr 4e-4 wd 1e-3" if kind == "train" else None,
                       "parallelism": f"dp{world}"},
            "roofline": roofline, "cpu_baseline": cpu, "frontend": frontend, "end_to_end": e2e, "eval": quality,
            "frontend_only": fe_lines, "secondary": secondary,
            "world": {"size": world, "backend": "rccl (torch.distributed nccl)" if ddp else None,
                      "rccl_version": rccl, "devices": devices,
                      "collectives": ({"what": "gradient arena all-reduce + packed SyncBatchNorm sums (all-reduce)",
                                       "per_step": coll["count"] / a.steps, "bytes_per_step": coll["bytes"] / a.steps,
                                       "issued": "only with more than one rank (or AFD_FORCE_COLLECTIVES=1)",
                                       "direct_rccl": bool(_ops._direct_rccl)}
                                      if ddp and kind == "train" else None)},
            "classes": classes, "class_timing_steps": timed_steps, "last_loss": loss,
            # every library launch belongs to a class; what is left of the step is the framework's own small kernels
            # (fills, copies, the loss read-back) and gaps between launches
            "classes_sum_ms_per_step": classes_sum, "classes_share_of_step": classes_sum / step_ms if step_ms else None,
        }
        print(json.dumps(line), flush=True)
    if ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
