"""Which host lines issue the small framework (aten) operations of a train step (development): a TorchDispatchMode over
two steps of a workload, every aten op on a GPU tensor listed with the audiofakedetect source line that called it.
    python3 tools/glue_trace.py [workload]"""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "audiodeepfake-detection_amd"))
import torch
import bench
from torch.utils._python_dispatch import TorchDispatchMode
wl = sys.argv[1] if len(sys.argv) > 1 else "coif4-l8"
dev = torch.device("cuda:0")
torch.manual_seed(0)
args, trainer, _ = bench.build(wl, 128, False, dev)
batch = bench.synthetic_batch(128, 0, dev)
trainer.model.train()
for _ in range(3): trainer._run_batch(0, batch)
torch.cuda.synchronize()
count = collections.Counter()
SKIP = ("aten.view", "aten.detach", "aten._unsafe_view", "aten.permute", "aten.select", "aten.slice", "aten.alias",
        "aten.empty", "aten.as_strided", "aten.t.", "aten.transpose", "aten.expand", "aten.unsqueeze", "aten.squeeze",
        "aten.reshape", "aten.is_", "aten.size", "aten.stride", "aten.storage_offset", "aten.sym_", "aten.dim", "aten.numel",
        "aten._local_scalar_dense", "aten.lift_fresh", "aten.empty_like", "aten.new_empty", "aten.unbind", "aten.split")
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            where = "?"
            for fr in reversed(traceback.extract_stack()):
                if "audiofakedetect" in fr.filename and "glue_trace" not in fr.filename:
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:90]}"
                    break
            count[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with Mode():
    for _ in range(2): trainer._run_batch(0, batch)
torch.cuda.synchronize()
for (op, where), n in sorted(count.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{n / 2:5.1f}/step  {op:34s} {where}")
