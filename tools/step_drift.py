"""Does the step time drift with the number of steps?  (development) per-step wall time, allocator and gc state.
    python3 tools/step_drift.py [workload] [steps] [gc: on|off|freeze]"""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "audiodeepfake-detection_amd"))
import torch
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "coif4-l8"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mode = sys.argv[3] if len(sys.argv) > 3 else "on"
dev = torch.device("cuda:0")
torch.manual_seed(0)
args, trainer, _ = bench.build(wl, 128, False, dev)
batch = bench.synthetic_batch(128, 0, dev)
trainer.model.train()
for _ in range(5): trainer._run_batch(0, batch)
torch.cuda.synchronize()
if mode == "off": gc.disable()
if mode == "freeze": gc.collect(); gc.freeze()
ts = []
for i in range(steps):
    t0 = time.perf_counter()
    trainer._run_batch(0, batch)
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
    if (i + 1) % (steps // 10) == 0:
        seg = ts[-(steps // 10):]
        print(f"steps {i + 1 - steps // 10:4d}-{i + 1:4d}: mean {sum(seg) / len(seg):.3f} ms  min {min(seg):.3f}  max {max(seg):.3f}  "
              f"alloc {torch.cuda.memory_allocated() / 2**20:.0f} MiB reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB  "
              f"gc counts {gc.get_count()} objects {len(gc.get_objects())}", flush=True)
