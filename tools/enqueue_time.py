"""Is a step bound by the GPU or by the host that enqueues it?
    python3 tools/enqueue_time.py <workload> [batch]
Prints the host time to ENQUEUE a step (no synchronisation inside the loop) next to the step time."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "audiodeepfake-detection_amd"))
import bench

w = sys.argv[1]; b = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda:0")
torch.manual_seed(0)
args, trainer, transforms = bench.build(w, b, False, dev)
batch = bench.synthetic_batch(b, 0, dev)
kind = bench.WORKLOADS[w][4]
if kind == "eval":
    trainer.model.eval()
else:
    trainer.model.train()

def step():
    if kind == "train":
        trainer._run_batch(0, batch)
    else:
        with torch.no_grad():
            trainer.model(trainer._features(batch["audio"]))

for _ in range(5): step()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{w}: enqueue {1e3*(t1-t0)/n:.3f} ms/step, complete {1e3*(t2-t0)/n:.3f} ms/step", flush=True)
