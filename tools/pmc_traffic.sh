# HBM traffic counters of one bench workload, one counter per pass (gfx950: FETCH_SIZE costs 3 TCC slots):
#   tools/pmc_traffic.sh <workload> <batch> <steps> <out json>
# rocprofv3 gets the program itself after `--` (python3 bench.py ...), no wrappers.
w=$1; b=$2; st=$3; out=$4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c && mkdir -p gpurun_out/pmc_$c
  timeout -k 10 500 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --workload $w --batch $b --steps $st --warmup 1 --cpu-frames 0 --e2e-steps 0 --no-frontends --no-secondary > gpurun_out/pmc_$c/bench.json 2> gpurun_out/pmc_$c/bench.err || exit 1
done
python3 tools/pmc_json.py $w $b $out
