# usage: tools/pmc.sh <conv name> <counters...>
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/pmc && mkdir -p gpurun_out/pmc
timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc -- python3 tools/conv_bench.py $name > gpurun_out/pmc/out.txt 2>&1
f=$(find gpurun_out/pmc -name "*counter_collection.csv" | head -1)
python3 - "$f" <<PY
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in rows:
    m = re.search(r"(\w+_kernel<[^>]*>|\w+_kernel)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:30]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k, d in agg.items():
    if "conv" not in k and "wgrad" not in k: continue
    n = len(disp[k]); print(k, "dispatches", n, "grid", )
    for c, v in sorted(d.items()): print("   %-28s %.4g per dispatch" % (c, v / n))
PY
