# rocprofv3 kernel trace of a python tool: tools/prof_py.sh <tag> <script.py> [args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$tag && mkdir -p gpurun_out/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 "$@" > gpurun_out/prof_$tag/out.txt 2> gpurun_out/prof_$tag/err.log
f=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    nm=re.sub(r'\(anonymous namespace\)::','',r['Name'])[:90]
    print(f"{int(r['Calls']):5d} calls avg {float(r['AverageNs'])/1e3:9.1f} us  {nm}")
PY
