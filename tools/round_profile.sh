# Round-end evidence: rocprofv3 kernel statistics of the headline bench and the other workloads, copied
# into profiles/ (run through gpurun; rocprofv3 gets the program itself after `--`).
#   tools/round_profile.sh <round tag, e.g. r05>
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # name, bench flags...
  name=$1; shift
  O=gpurun_out/rp_$name; rm -rf $O; mkdir -p $O
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py "$@" > $O/line.json 2> $O/err.txt || return 1
  cp $(find $O/prof -name "*kernel_stats.csv" | head -1) profiles/${tag}_${name}_kernel_stats.csv
  cp $O/line.json profiles/${tag}_${name}_line.json
  echo "$name done"
}
run bench_coif4l14_b128 --steps 5 --warmup 3 --cpu-frames 0 --e2e-steps 0 --no-frontends --no-secondary &&
run bench_sym5l14_b128 --workload sym5-l14 --steps 5 --warmup 3 --cpu-frames 0 --e2e-steps 0 &&
run bench_coif4l8_b128 --workload coif4-l8 --steps 10 --warmup 3 --cpu-frames 0 --e2e-steps 0 &&
run bench_sym5l8_b128 --workload sym5-l8 --steps 10 --warmup 3 --cpu-frames 0 --e2e-steps 0 &&
run bench_stft_b128 --workload stft --steps 10 --warmup 3 --cpu-frames 0 --e2e-steps 0 &&
run bench_sym8l8_b128 --workload sym8-l8 --steps 10 --warmup 3 --cpu-frames 0 --e2e-steps 0 &&
run frontend_coif4l14_b128 --workload coif4-l14-frontend --steps 30 --warmup 5 --cpu-frames 0 &&
run frontend_coif4l14_b4096 --workload coif4-l14-frontend --batch 4096 --steps 20 --warmup 5 --cpu-frames 0 &&
run frontend_sym5l14_b128 --workload sym5-l14-frontend --steps 30 --warmup 5 --cpu-frames 0 &&
run frontend_sym5l14_b4096 --workload sym5-l14-frontend --batch 4096 --steps 20 --warmup 5 --cpu-frames 0 &&
run frontend_haarl14_b4096 --workload haar-l14-frontend --steps 30 --warmup 5 --cpu-frames 0 &&
run lcnn_eval_bf16 --workload stft-lcnn-eval-bf16 --steps 30 --warmup 5 --cpu-frames 0 &&
run lcnn_eval_bf16_b1024 --workload stft-lcnn-eval-bf16 --batch 1024 --steps 20 --warmup 5 --cpu-frames 0 &&
cp profiles/${tag}_*kernel_stats.csv profiles/${tag}_*_line.json gpurun_out/
