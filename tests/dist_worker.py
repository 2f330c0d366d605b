"""Worker of tests/test_distributed.py: one rank of a world_size-2 job (gloo rendezvous on 127.0.0.1)."""

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "audiodeepfake-detection_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def cpu_sync(rank, world):
    """Host-side data-parallel logic on CPU tensors: replica broadcast, flat-arena gradient
    all-reduce == full-batch gradient, packed BatchNorm statistics == global-batch statistics,
    sampler shards are disjoint."""
    from audiofakedetect import ops
    from audiofakedetect.train_classifier import DataParallelRCCL, sync_gradients
    from audiofakedetect.data_loader import SyntheticFrames
    from torch.utils.data.distributed import DistributedSampler

    torch.manual_seed(100 + rank)  # replicas start different ...
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))
    wrapped = DataParallelRCCL(net)  # ... and are made equal by the broadcast
    ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))
    gather = [torch.zeros(47) for _ in range(world)]
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    dist.all_gather(gather, flat)
    assert all(torch.equal(g, gather[0]) for g in gather), "replicas differ after broadcast"
    ref.load_state_dict(net.state_dict())

    opt = ops.FusedAdam(net.parameters(), lr=1e-3)  # arena logic is device agnostic
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 6, generator=g)
    y = torch.randint(0, 2, (8,), generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    opt.zero_grad()
    torch.nn.functional.cross_entropy(wrapped(xs), ys).backward()
    scale = sync_gradients(wrapped, opt)
    assert abs(scale - 1.0 / world) < 1e-12
    torch.nn.functional.cross_entropy(ref(x), y).backward()
    full = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    assert torch.allclose(opt.flat_grad * scale, full, atol=1e-6), "all-reduced grads != full-batch grads"

    # a backward pass that raised leaves its one-shot hook queued: the next step must not inherit it
    from audiofakedetect.train_classifier import start_gradient_allreduce
    start_gradient_allreduce(opt)
    start_gradient_allreduce(opt)
    assert len(ops._end_of_backward) == 1, "stale end-of-backward hook survived"
    ops._end_of_backward.clear()

    # packed BN statistics: [sum | sumsq | count] all-reduce -> global mean / biased var
    c = 3
    data = torch.randn(8, c, 5, generator=torch.Generator().manual_seed(9)).double()
    local = data[rank * 4:(rank + 1) * 4]
    sums = torch.zeros(2 * c + 1, dtype=torch.float64)
    sums[:c] = local.sum((0, 2))
    sums[c:2 * c] = (local * local).sum((0, 2))
    rm, rv = torch.zeros(c), torch.ones(c)
    mean, invstd, cnt = ops.bn_finalize(sums, c, float(local.numel() // c), 1e-5, True, rm, rv,
                                        torch.zeros((), dtype=torch.long), 0.1)
    assert float(cnt) == data.numel() // c
    assert torch.allclose(mean.double(), data.mean((0, 2)), atol=1e-6)
    var = data.var((0, 2), unbiased=False)
    assert torch.allclose(invstd.double(), torch.rsqrt(var + 1e-5), atol=1e-5)
    assert torch.allclose(rv.double(), 0.9 + 0.1 * data.var((0, 2), unbiased=True), atol=1e-6)

    # sampler: disjoint shards that cover the set (reference train_classifier.py:119-127)
    ds = SyntheticFrames(16, 64)
    sampler = DistributedSampler(ds, shuffle=True, seed=0, drop_last=True)
    sampler.set_epoch(1)
    mine = torch.tensor(list(iter(sampler)))
    allidx = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allidx, mine)
    assert sorted(torch.cat(allidx).tolist()) == list(range(16))


def gpu_dcnn(rank, world):
    """Both ranks on cuda:0 (gloo moves the GPU tensors): sharded DCNN step with cross-rank
    BatchNorm statistics and the flat gradient all-reduce == single-process full-batch step."""
    from audiofakedetect import ops
    from audiofakedetect.models import DCNN
    from audiofakedetect.train_classifier import DataParallelRCCL, start_gradient_allreduce, sync_gradients
    from audiofakedetect.utils import DotDict

    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    # geometry: the shipped sym5 level-8 model by default; AFD_TEST_GEOMETRY picks the others -- every fused unit of the
    # step (input folds, BatchNorm-backward epilogues, the block-2 pass) has its own packed-sum exchange, and which of
    # them run depends on the feature shape
    packets, frames_t, add, per_rank = {"sym5l8": (256, 95, 1, 4), "coif4l8": (256, 109, 0, 4), "stft": (256, 101, 0, 4),
                                        "coif4l14": (16384, 24, 0, 2)}[os.environ.get("AFD_TEST_GEOMETRY", "sym5l8")]
    flat = 40 * (packets // 8 - 24)
    total = per_rank * world

    def make(ddp):
        torch.manual_seed(3)
        a = DotDict(input_dim=[per_rank, 1, packets, frames_t], ochannels1=64, ochannels2=64, ochannels3=96,
                    ochannels4=128, ochannels5=32, kernel1=3, dropout_cnn=0.0, dropout_lstm=0.0,
                    time_dim_add=add, flattend_size=flat, ddp=ddp)
        return DCNN(a).cuda().train()

    g = torch.Generator().manual_seed(5)
    x = torch.randn(total, 1, packets, frames_t, generator=g).cuda()
    y = torch.randint(0, 2, (total,), generator=g).cuda()
    mine = slice(rank * per_rank, (rank + 1) * per_rank)

    def tapped_forward(model, inp):
        ops.debug_taps = []
        try:
            return model(inp), ops.debug_taps
        finally:
            ops.debug_taps = None

    ref = make(False)  # no cross-rank statistics: plain full-batch step in this process
    ropt = ops.FusedAdam(ref.parameters(), lr=4e-4, weight_decay=1e-3)
    ropt.zero_grad()
    rout, rtaps = tapped_forward(ref, x)

    net = make(True)
    wrapped = DataParallelRCCL(net)
    opt = ops.FusedAdam(net.parameters(), lr=4e-4, weight_decay=1e-3)
    opt.zero_grad()
    ops.collective_counters(reset=True)
    out, taps = tapped_forward(wrapped, x[mine])
    # The shard's BatchNorm statistics are the full batch's up to the rounding of a different summation order, so a
    # 2x2 max-pool near tie or a PReLU zero crossing may go the other way in a handful of positions; each re-routes one
    # gradient element, which at 8 frames is 1e-3 of the gradient norm (tests/test_dcnn_gpu.py documents the same for
    # the reference's own fp32 run).  Those decisions are aligned with the full-batch run's before backward -- what is
    # compared is the arithmetic of the sharded step, to 1e-4, and the number of such positions is bounded.
    assert len(taps) == len(rtaps)
    flips = decisions = 0
    for (kind, t), (rkind, r) in zip(taps, rtaps):
        assert kind == rkind and t.shape[1:] == r.shape[1:] and r.shape[0] == total, (kind, t.shape, r.shape)
        r = r[mine]
        diff = (t != r) if kind == "pool" else ((t <= 0) != (r <= 0))
        decisions += diff.numel()
        nd = int(diff.sum())
        if nd:
            flips += nd
            if kind == "pool":
                t.data[diff] = r[diff]
            else:
                assert t.data[diff].abs().max().item() <= 2e-5
                tiny = torch.full_like(t.data[diff], 1e-30)
                t.data[diff] = torch.where(r[diff] <= 0, -tiny, tiny)
    assert flips <= max(8, decisions // 100000), f"{flips} routing differences in {decisions} decisions"
    ops.CrossEntropyLoss()(rout, y).backward()
    loss = ops.CrossEntropyLoss()(out, y[mine])
    start_gradient_allreduce(opt)  # as left behind by a step whose backward raised before its callbacks ran
    start_gradient_allreduce(opt)  # the all-reduce is issued (once) by the end-of-backward hook
    loss.backward()
    assert getattr(opt, "_pending_allreduce", None) is not None
    scale = sync_gradients(wrapped, opt)
    # the step's exchanges: one packed-sum all-reduce per BatchNorm forward and backward (8 + 8: reference
    # models.py:260-289, SyncBatchNorm) and ONE all-reduce over the gradient arena -- every packed-sum site of the fused
    # units has therefore been through a two-rank reduction
    coll = ops.collective_counters(reset=True)
    assert coll["count"] == 17, coll
    assert coll["bytes"] >= opt.flat_grad.numel() * 4, coll
    grads = (opt.flat_grad * scale).clone()
    bn_rm = net.cnn[3].running_mean.clone()

    err = (out - rout[mine]).abs().max().item()
    assert err <= 1e-4, f"sharded logits differ from full-batch logits: {err}"
    rel = ((grads - ropt.flat_grad).norm() / ropt.flat_grad.norm()).item()
    assert rel <= 1e-4, f"all-reduced gradients differ from full-batch gradients: {rel} ({flips} routing positions aligned)"
    assert torch.allclose(bn_rm, ref.cnn[3].running_mean, atol=1e-5)
    opt.step(grad_scale=scale)
    ropt.step()
    d = (opt.flat - ropt.flat).abs()
    assert int((d > 2e-5).sum()) <= d.numel() // 100, "Adam update differs"


def gpu_one_rank_direct(rank, world):
    """One RCCL rank with every collective of the step forced (AFD_FORCE_COLLECTIVES): the step through
    torch.distributed and the step through the library's own communicator on the compute stream
    (ops.enable_direct_rccl) give the same loss, gradients and BatchNorm running statistics -- a sum over one rank is
    the identity either way."""
    from audiofakedetect import ops
    from audiofakedetect.models import DCNN
    from audiofakedetect.train_classifier import DataParallelRCCL, start_gradient_allreduce, sync_gradients
    from audiofakedetect.utils import DotDict

    assert world == 1
    os.environ["AFD_FORCE_COLLECTIVES"] = "1"
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 1, 256, 95, generator=g).cuda()
    y = torch.randint(0, 2, (4,), generator=g).cuda()

    def step():
        torch.manual_seed(3)
        a = DotDict(input_dim=[4, 1, 256, 95], ochannels1=64, ochannels2=64, ochannels3=96, ochannels4=128,
                    ochannels5=32, kernel1=3, dropout_cnn=0.0, dropout_lstm=0.0, time_dim_add=1, flattend_size=320, ddp=True)
        net = DCNN(a).cuda().train()
        wrapped = DataParallelRCCL(net)
        opt = ops.FusedAdam(net.parameters(), lr=4e-4, weight_decay=1e-3)
        opt.zero_grad()
        loss = ops.CrossEntropyLoss()(wrapped(x), y)
        start_gradient_allreduce(opt)
        loss.backward()
        assert getattr(opt, "_pending_allreduce", None) is not None
        scale = sync_gradients(wrapped, opt)
        torch.cuda.synchronize()
        return loss.detach().clone(), opt.flat_grad.clone(), net.cnn[3].running_mean.clone(), scale

    l0, g0, rm0, s0 = step()
    assert ops.enable_direct_rccl() and _native_world() == 1
    l1, g1, rm1, s1 = step()
    ops.disable_direct_rccl()
    assert s0 == s1 == 1.0
    # (not bit for bit: a few sums of the step -- PReLU slope gradients, BatchNorm sums -- are float / double atomics
    # whose order differs from run to run on either path)
    assert abs(float(l0) - float(l1)) <= 1e-6, (float(l0), float(l1))
    rel = ((g0 - g1).norm() / g0.norm()).item()
    assert rel <= 1e-5, f"gradients through the direct communicator differ from the c10d ones: {rel}"
    assert torch.allclose(rm0, rm1, atol=1e-6)
    assert float(g0.abs().sum()) > 0.0


def _native_world():
    from audiofakedetect import _native

    return _native.load().afd_rccl_world()


if __name__ == "__main__":
    mode = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    # "gpu_dcnn_rccl": one rank per GPU over RCCL (backend "nccl" is RCCL on ROCm); the others over gloo
    backend = "nccl" if "_rccl" in mode else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    try:
        if mode == "gpu_dcnn_rccl_direct":  # the same sharded step with the in-step collectives on the compute stream
            from audiofakedetect import ops

            assert ops.enable_direct_rccl()
        {"cpu_sync": cpu_sync, "gpu_dcnn": gpu_dcnn, "gpu_dcnn_rccl": gpu_dcnn, "gpu_dcnn_rccl_direct": gpu_dcnn,
         "gpu_one_rank_direct_rccl": gpu_one_rank_direct}[mode](rank, world)
        if mode == "gpu_dcnn_rccl_direct":
            ops.disable_direct_rccl()
    finally:
        dist.destroy_process_group()
    print(f"rank {rank} ok")
