# per-kernel times of the packet front end (development helper)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/fe; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 tools/frontend_bench.py > $O/out.txt 2> $O/err.txt
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/fe/prof/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'wpt' in n or 'haar' in n:
        print(f"{float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']:>4}  {n[:110]}")
PY
