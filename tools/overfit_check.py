"""Sanity run: the train step on ONE fixed synthetic batch must drive the loss down (all kernels of the step in the loop).
    python3 tools/overfit_check.py <workload> [steps] [batch]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "audiodeepfake-detection_amd"))
import bench

w = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60; b = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda:0")
torch.manual_seed(0)
args, trainer, transforms = bench.build(w, b, False, dev)
batch = bench.synthetic_batch(b, 0, dev)
trainer.model.train()
first = last = None
for i in range(steps):
    trainer._run_batch(0, batch)
    lv = trainer.loss_list[-1][2]
    if not (lv == lv) or abs(lv) == float("inf"):
        raise SystemExit(f"step {i}: loss {lv}")
    first = lv if first is None else first
    last = lv
    if i % 10 == 0 or i == steps - 1:
        print(f"step {i:3d} loss {lv:.4f} acc {trainer.accuracy_list[-1][2]:.3f}", flush=True)
print("loss %.4f -> %.4f" % (first, last))
if not last < 0.7 * first:
    raise SystemExit("the loss did not fall")
