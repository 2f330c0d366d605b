# rocprofv3 kernel trace of the bench step: tools/prof_step.sh <tag> [bench args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$tag && mkdir -p gpurun_out/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 5 --warmup 3 --cpu-frames 0 --e2e-steps 0 --no-frontends --no-secondary "$@" > gpurun_out/prof_$tag/line.json 2> gpurun_out/prof_$tag/err.log
f=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/prof_${tag}_kernel_stats.csv
python3 - "$f" gpurun_out/prof_$tag/line.json <<'PY'
import csv,json,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
# steps in the trace = warm-up + timed + the class-timing steps bench.py runs after the timed region
line=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1])
nsteps=line["warmup"]+line["steps"]+int(line.get("class_timing_steps") or 0)
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:28]:
    nm=re.sub(r'\(anonymous namespace\)::','',r['Name'])[:100]
    print(f"{float(r['TotalDurationNs'])/1e6/nsteps:8.3f} ms/step {int(r['Calls']):5d} calls avg {float(r['AverageNs'])/1e3:9.1f} us  {nm}")
print(f'total kernel ms/step ({nsteps} steps in the trace)', tot/1e6/nsteps)
PY
