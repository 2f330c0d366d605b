# Round-end measurement set: GPU tests, headline bench under rocprofv3 --kernel-trace --stats,
# the other workloads' bench lines.  Output under gpurun_out/round/.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/round; rm -rf $O; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 && tail -2 $O/pytest_gpu.log
python3 bench.py > $O/bench_coif4l14.json 2> $O/bench_coif4l14.err && cat $O/bench_coif4l14.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 3 --warmup 1 --cpu-frames 0 > $O/bench_prof.json 2> $O/bench_prof.err
for w in coif4-l8 sym5-l8 stft stft-lcnn-eval; do
  python3 bench.py --workload $w --cpu-frames 0 --steps 10 --warmup 3 > $O/bench_$w.json 2> $O/bench_$w.err && cat $O/bench_$w.json
done
python3 tools/frontend_bench.py > $O/frontend.log 2>&1 && cat $O/frontend.log
