# Headline bench line + rocprofv3 kernel statistics of the same command, with and without the
# backward-weight stream, + the HBM traffic counters (separate --pmc passes).  Output: gpurun_out/final/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; rm -rf $O; mkdir -p $O
python3 bench.py > $O/bench_line.json 2> $O/bench_line.err && cut -c1-200 $O/bench_line.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 5 --warmup 2 --cpu-frames 0 > $O/bench_prof.json 2> $O/bench_prof.err
echo "prof two-stream done"
AFD_WGRAD_STREAM=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -- python3 bench.py --steps 5 --warmup 2 --cpu-frames 0 > $O/bench_prof_serial.json 2> $O/bench_prof_serial.err
echo "prof serial done"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c && mkdir -p gpurun_out/pmc_$c
  timeout -k 10 500 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 > gpurun_out/pmc_$c/bench.json 2> gpurun_out/pmc_$c/bench.err
  echo "pmc $c done"
done
ls gpurun_out/pmc_FETCH_SIZE/*/ | head -3
