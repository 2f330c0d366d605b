# HBM traffic counters of the headline bench, one counter per pass (gfx950: FETCH_SIZE costs 3 TCC slots)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c && mkdir -p gpurun_out/pmc_$c
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 > gpurun_out/pmc_$c/bench.json 2> gpurun_out/pmc_$c/bench.err
  f=$(find gpurun_out/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c <<PY
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(float); n = collections.Counter()
for r in rows:
    if r["Counter_Name"] != sys.argv[2]: continue
    m = re.search(r"(\w+_kernel)", r["Kernel_Name"]); k = m.group(1) if m else "other"
    agg[k] += float(r["Counter_Value"]); n[k] += 1
for k in sorted(agg, key=lambda k: -agg[k])[:12]:
    print("%s %-28s launches %4d  total %10.1f MB  per launch %9.2f MB (counter unit: KB)" % (sys.argv[2], k, n[k], agg[k]/1024, agg[k]/1024/n[k]))
PY
done
