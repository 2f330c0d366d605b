# rocprofv3 kernel statistics of the headline step under two environments (development helper):
#   tools/prof_ab.sh <tagA> "<env A>" <tagB> "<env B>"     (env = NAME=VAL, one assignment or "-")
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {
  tag=$1; envs=$2
  O=gpurun_out/pab_$tag; rm -rf $O; mkdir -p $O
  if [ "$envs" != "-" ]; then export $envs; fi
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 5 --warmup 3 --cpu-frames 0 --e2e-steps 0 --no-frontends --no-secondary > $O/line.json 2> $O/err.txt
  if [ "$envs" != "-" ]; then unset ${envs%%=*}; fi
  cp $(find $O/prof -name "*kernel_stats.csv" | head -1) gpurun_out/pab_${tag}_kernel_stats.csv
}
run $1 "$2" && run $3 "$4"
python3 - $1 $3 <<'PY'
import csv, sys
a, b = sys.argv[1:3]
def load(t):
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(f"gpurun_out/pab_{t}_kernel_stats.csv"))}
A, B = load(a), load(b)
rows = sorted(A, key=lambda k: -A[k][0] * A[k][1])[:32]
print(f"{'kernel':100s} calls {a:>10s} {b:>10s}  (us per launch)")
for k in rows:
    print(f"{k[:100]:100s} {A[k][0]:5d} {A[k][1]:10.1f} {B.get(k, (0, 0))[1]:10.1f}")
PY
