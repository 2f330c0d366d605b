# rocprofv3 kernel statistics of one bench command (development helper):  tools/prof_one.sh <tag> <bench flags...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/p1_$tag; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py "$@" > $O/line.json 2> $O/err.txt
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) gpurun_out/p1_${tag}_kernel_stats.csv
python3 - gpurun_out/p1_${tag}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print(f"{r['Name'][:110]:110s} {int(r['Calls']):5d} {float(r['TotalDurationNs'])/1e6:9.3f} ms {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):5.1f}%")
PY
