"""Turns the rocprofv3 --pmc passes of tools/pmc_traffic.sh into profiles/<name>.json.

usage: pmc_json.py <workload> <batch> <out.json>

FETCH_SIZE / WRITE_SIZE are reported in KB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE is
TCC_EA0_RDREQ x 64 B -- a 128-byte read request is tallied as 64 -- so it under-reports a kernel by the share of its
read requests that are 128 bytes wide (exactly 1/2 for 16-byte-per-lane coalesced streams, 0 for narrow loads).  The
share is MEASURED per kernel, not kept in a hand list: pass 3 counts the read requests by size, and
    fetch_bytes = 32 n32 + 64 n64 + 128 n128,   fetch_factor = fetch_bytes / FETCH_SIZE (1 ... 2)
(pass 4: the chip's own count of the same traffic in 32-byte units, TCC_EA0_RDREQ_DRAM_32B, recorded beside it).
Without pass 3 the raw value is kept and `fetch_calibrated` is false.  WRITE_SIZE is exact for 16-byte-per-lane
streaming stores.  Totals are over every dispatch of a kernel in the trace; `steps_in_trace` = the number of bench
steps the trace covers (warm-up, timed and the instrumented extra steps), so that bytes per step = total /
steps_in_trace (bench.py's roofline.traffic).  `classes`: per timing class of bench.py, PMC bytes per step beside the
algorithmic bytes per step of the same run's bench line; a class whose PMC bytes are below 0.9 x its algorithmic bytes
is listed in `below_algorithmic` (physically impossible for tensors that do not fit the 256 MiB Infinity Cache:
tests/test_bench_contract_gpu.py asserts the list is empty for the committed summary).
"""
import collections, csv, glob, json, os, re, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
workload, batch, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
out = {"note": __doc__.strip(), "workload": workload, "batch": batch, "kernels": {}}


def read_pass(i):
    """{counter: {kernel: total}}, {kernel: launches}, bench line of pass i (None if the pass is missing)."""
    files = glob.glob(f"gpurun_out/pmc_pass{i}/*/*counter_collection.csv")
    if not files:
        return None, None, None
    f = max(files, key=os.path.getmtime)
    line = None
    try:
        line = json.loads(open(f"gpurun_out/pmc_pass{i}/bench.json").read().strip().splitlines()[-1])
    except Exception:
        pass
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        k = m.group(1) if m else "other"
        agg[r["Counter_Name"]][k] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in disp.items()}, line


line = None
fetch, n_f, l1 = read_pass(1)
write, n_w, l2 = read_pass(2)
sizes, n_s, l3 = read_pass(3)
dram, n_d, l4 = read_pass(4)
line = l1 or l2
for k, v in (fetch or {}).get("FETCH_SIZE", {}).items():
    e = out["kernels"].setdefault(k, {})
    e["fetch_raw_bytes_total"] = v * 1024.0
    e["launches_in_trace"] = n_f[k]
for k, v in (write or {}).get("WRITE_SIZE", {}).items():
    e = out["kernels"].setdefault(k, {})
    e["write_bytes_total"] = v * 1024.0
    e.setdefault("launches_in_trace", n_w[k])
have_sizes = bool(sizes) and sum(sizes.get("TCC_EA0_RDREQ_128B_sum", {}).values()) + sum(sizes.get("TCC_EA0_RDREQ_64B_sum", {}).values()) > 0
for k, e in out["kernels"].items():
    raw = e.get("fetch_raw_bytes_total", 0.0)
    if have_sizes and k in n_s:
        # the passes are separate runs of the same deterministic command: scale by the launch counts if they differ
        scale = e["launches_in_trace"] / n_s[k] if n_s[k] else 1.0
        n32 = sizes.get("TCC_EA0_RDREQ_32B_sum", {}).get(k, 0.0) * scale
        n64 = sizes.get("TCC_EA0_RDREQ_64B_sum", {}).get(k, 0.0) * scale
        n128 = sizes.get("TCC_EA0_RDREQ_128B_sum", {}).get(k, 0.0) * scale
        by_size = 32.0 * n32 + 64.0 * n64 + 128.0 * n128
        e["read_requests"] = {"32B": n32, "64B": n64, "128B": n128}
        e["fetch_bytes_total"] = by_size
        e["fetch_factor"] = by_size / raw if raw else None
        e["fetch_calibrated"] = True
    else:
        e["fetch_bytes_total"] = raw
        e["fetch_factor"] = 1.0
        e["fetch_calibrated"] = False
    if dram and k in n_d:
        scale = e["launches_in_trace"] / n_d[k] if n_d[k] else 1.0
        e["fetch_dram_32B_units_bytes_total"] = 32.0 * dram.get("TCC_EA0_RDREQ_DRAM_32B_sum", {}).get(k, 0.0) * scale
        e["read_requests_total"] = dram.get("TCC_EA0_RDREQ_sum", {}).get(k, 0.0) * scale
# steps the trace covers: warm-up + timed + the instrumented ones bench.py adds after the timed region
steps = None
if line:
    steps = line["warmup"] + line["steps"] + line["class_timing_steps"]
    out["bench_line"] = {k: line[k] for k in ("value", "ms_per_step", "steps", "warmup")}
out["steps_in_trace"] = steps
# per timing class: PMC bytes per step against the algorithmic bytes of the same run
if line and steps:
    import bench  # CLASS_KERNELS: rocprof kernel names of each timing class

    out["classes"] = {}
    out["below_algorithmic"] = []
    for cls, c in (line.get("classes") or {}).items():
        names = bench.CLASS_KERNELS.get(cls, ())
        tot = sum(out["kernels"].get(nm, {}).get("fetch_bytes_total", 0.0) + out["kernels"].get(nm, {}).get("write_bytes_total", 0.0)
                  for nm in names) / steps
        algo = c.get("algorithmic_bytes_per_step")
        out["classes"][cls] = {"hbm_bytes_per_step_pmc": tot, "algorithmic_bytes_per_step": algo,
                               "ratio": tot / algo if algo else None}
        if algo and algo > 2.0 * 268435456 and tot < 0.9 * algo:
            out["below_algorithmic"].append(cls)
json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
per = max(steps or 1, 1)
for k, e in sorted(out["kernels"].items(), key=lambda kv: -(kv[1].get("fetch_bytes_total", 0) + kv[1].get("write_bytes_total", 0)))[:18]:
    print(f"{k:32s} launches {e.get('launches_in_trace', 0):4d}  fetch {e.get('fetch_bytes_total', 0)/1e9/per:8.3f} GB (x{e.get('fetch_factor') or 0:.2f} of FETCH_SIZE)  "
          f"write {e.get('write_bytes_total', 0)/1e9/per:8.3f} GB per step")
for cls, c in sorted((out.get("classes") or {}).items()):
    print(f"class {cls:16s} pmc {c['hbm_bytes_per_step_pmc']/1e9:8.3f} GB  algorithmic {(c['algorithmic_bytes_per_step'] or 0)/1e9:8.3f} GB  ratio {c['ratio'] if c['ratio'] is None else round(c['ratio'], 3)}")
print("below algorithmic:", out.get("below_algorithmic"))
