# per-kernel times of a front-end workload (development helper): tools/prof_frontend.sh <workload> [batch]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/fe_$1; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --workload $1 ${2:+--batch $2} --steps 30 --warmup 5 --cpu-frames 0 > $O/out.txt 2> $O/err.txt
python3 - $O <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/prof/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'wpt' in n or 'haar' in n:
        print(f"{float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']:>4}  {n[:100]}")
PY
