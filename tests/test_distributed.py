"""Data-parallel path on world_size 2 (gloo, 127.0.0.1): host logic on CPU, full DCNN step on GPU.

The reference tests its DDP aggregation only with hand-built "gathered" literals
(tests/test_trainer.py:14-117); here two real processes run the collectives.
"""

import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_world2(mode, timeout=300, one_gpu_per_rank=False, world=2, extra_env=None):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                   WORLD_SIZE=str(world), LOCAL_RANK=str(rank if one_gpu_per_rank else 0), OMP_NUM_THREADS="2",
                   HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), mode],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode(errors="replace"))
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out[-3000:]}"
        assert f"rank {rank} ok" in out


def test_world2_gloo_cpu_gradient_allreduce_bn_stats_sampler():
    _run_world2("cpu_sync")


@pytest.mark.gpu
@pytest.mark.parametrize("geometry", ["sym5l8", "coif4l8", "stft", "coif4l14"])
def test_world2_sharded_dcnn_step_equals_full_batch_step(geometry):
    """Two ranks (both on cuda:0, gloo) against the single-process full-batch step at every model geometry: the shipped
    level-8 / STFT shapes (4 frames per rank) and the level-14 benchmark shape (2 frames per rank).  Each geometry
    activates a different set of fused units -- the four input folds, both BatchNorm-backward epilogues, the one-pass
    block 2 -- and each unit's packed BatchNorm sums must come out of a real two-rank reduction: 8 + 8 + 1 collectives,
    sharded logits / gradients / running statistics / Adam update equal to the full batch's -- the gradients to 1e-4 with
    the handful of max-pool near ties / PReLU zero crossings that went the other way (the shard sums its statistics in
    another order) aligned with the full-batch run first, as tests/test_dcnn_gpu.py does against the reference's run."""
    _run_world2("gpu_dcnn", extra_env={"AFD_TEST_GEOMETRY": geometry}, timeout=600)


def _gpu_count():
    import torch

    return torch.cuda.device_count()


@pytest.mark.gpu
@pytest.mark.skipif(_gpu_count() < 2, reason="needs two MI355X: one RCCL rank per GPU")
def test_world2_rccl_sharded_dcnn_step_equals_full_batch_step():
    """The same check over RCCL (backend "nccl"), one rank per GPU: replica broadcast, packed SyncBN
    statistics and the flat gradient all-reduce travel over xGMI."""
    _run_world2("gpu_dcnn_rccl", one_gpu_per_rank=True)


@pytest.mark.gpu
@pytest.mark.skipif(_gpu_count() < 2, reason="needs two MI355X: one RCCL rank per GPU")
def test_world2_direct_rccl_sharded_dcnn_step_equals_full_batch_step():
    """The sharded step with the in-step collectives issued by the library itself on the compute stream
    (ops.enable_direct_rccl: a second communicator from a broadcast unique id)."""
    _run_world2("gpu_dcnn_rccl_direct", one_gpu_per_rank=True)


@pytest.mark.gpu
def test_one_rank_direct_rccl_step_equals_the_c10d_step():
    """What one GPU can check of the direct path: librccl is found in the process, the communicator comes up from a
    broadcast unique id, ncclAllReduce runs on the compute stream, and the forced-collectives step equals the step
    through torch.distributed."""
    _run_world2("gpu_one_rank_direct_rccl", world=1)


@pytest.mark.gpu
@pytest.mark.skipif(_gpu_count() < 2, reason="needs two MI355X")
def test_bench_two_gpus_spawns_ranks_and_reports_the_world():
    """`python bench.py --gpus 2` (no launcher): two RCCL ranks, weak scaling, one JSON line."""
    import json

    root = os.path.dirname(HERE)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "sym5-l8",
                          "--steps", "3", "--warmup", "2", "--cpu-frames", "0"], cwd=root, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["world"]["size"] == 2 and len(d["world"]["devices"]) == 2
    assert d["config"]["global_batch"] == 2 * d["config"]["batch_per_gpu"] and d["scaling"] == "weak"
