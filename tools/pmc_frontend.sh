# usage: tools/pmc_frontend.sh <workload> <tag> <counters...>   (rocprofv3 --pmc pass over the front-end bench)
w=$1; tag=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/pmcfe_$tag && mkdir -p gpurun_out/pmcfe_$tag
timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmcfe_$tag -- python3 bench.py --workload $w --batch ${AFD_FE_BATCH:-128} --steps 5 --warmup 2 --cpu-frames 0 > gpurun_out/pmcfe_$tag/out.txt 2> gpurun_out/pmcfe_$tag/err.txt
f=$(find gpurun_out/pmcfe_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" <<PY
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in rows:
    m = re.search(r"(\w+_kernel)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:30]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k, d in agg.items():
    if "wpt" not in k: continue
    n = len(disp[k]); print(k, "dispatches", n)
    for c, v in sorted(d.items()): print("   %-28s %.5g per dispatch" % (c, v / n))
PY
