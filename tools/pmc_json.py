"""Turns the two rocprofv3 --pmc passes of tools/pmc_traffic.sh into profiles/<name>.json.

usage: pmc_json.py <workload> <batch> <out.json>

FETCH_SIZE / WRITE_SIZE are reported in KB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts
64 B per 128-B request of wide (16 B/lane) coalesced reads, so it is doubled for the kernels whose streaming
loads are dwordx4 (WIDE below); other widths are uncalibrated and recorded raw.  WRITE_SIZE is exact for
16-byte-per-lane streaming stores.  Totals are over every dispatch of a kernel in the trace; `steps_in_trace`
= the number of bench steps the trace covers (warm-up, timed and the instrumented extra steps), so that
bytes per step = total / steps_in_trace (bench.py's roofline.traffic).
"""
import collections, csv, glob, json, os, re, sys

workload, batch, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
# Calibration of the patch-row readers (round 3): for wino44_wgrad_kernel on the block-3 shape the RAW FETCH_SIZE
# (7.24 GB) equals TCC_MISS_sum x 64 B (7.5 GB) and the two tensors read once (8.3 GB) -- the 16-byte + 8-byte loads
# at 4-byte alignment of the Winograd patch rows are counted in full, so the Winograd kernels are NOT doubled
# (round 2 doubled wino44_conv_kernel and carried a caveat); the x2 stays for the aligned 16-byte streaming readers.
WIDE = ("conv3x3_kernel", "wgrad3x3_kernel", "wgrad3x3p_kernel", "conv1x1_kernel",
        "wpt_haar14_kernel", "conv_wgrad2_kernel", "bn_stats_kernel", "bn_apply_fwd_kernel", "bn_bwd_stats_kernel",
        "bn_bwd_apply_kernel", "prelu_pool_fwd_kernel", "prelu_pool_bwd_kernel")
out = {"note": __doc__.strip(), "workload": workload, "batch": batch, "kernels": {}}
line = None
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    f = max(glob.glob(f"gpurun_out/pmc_{counter}/*/*counter_collection.csv"), key=os.path.getmtime)
    try:
        line = json.loads(open(f"gpurun_out/pmc_{counter}/bench.json").read().strip().splitlines()[-1])
    except Exception:
        pass
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"]); k = m.group(1) if m else "other"
        agg[k] += float(r["Counter_Value"]) * 1024.0; n[k] += 1
    for k in agg:
        e = out["kernels"].setdefault(k, {})
        if counter == "FETCH_SIZE":
            e["fetch_raw_bytes_total"] = agg[k]
            e["fetch_bytes_total"] = agg[k] * (2.0 if k in WIDE else 1.0)
            e["fetch_corrected_x2"] = k in WIDE
        else:
            e["write_bytes_total"] = agg[k]
        e["launches_in_trace"] = n[k]
# steps the trace covers: warm-up + timed + the instrumented ones bench.py adds after the timed region
steps = None
if line:
    # bench.py reports how many instrumented steps it ran after the timed region (`class_timing_steps`)
    steps = line["warmup"] + line["steps"] + line["class_timing_steps"]
    out["bench_line"] = {k: line[k] for k in ("value", "ms_per_step", "steps", "warmup")}
out["steps_in_trace"] = steps
json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
for k, e in sorted(out["kernels"].items(), key=lambda kv: -(kv[1].get("fetch_bytes_total", 0) + kv[1].get("write_bytes_total", 0)))[:14]:
    print(f"{k:28s} launches {e['launches_in_trace']:4d}  fetch {e.get('fetch_bytes_total', 0)/1e9/max(steps or 1,1):8.3f} GB  write {e.get('write_bytes_total', 0)/1e9/max(steps or 1,1):8.3f} GB per step")
