# Round-5 evidence set in one gpurun call (about 15 GPU-minutes): kernel statistics of every workload, the default bench
# line, HBM traffic summaries (headline step, LCNN bf16 at B = 1024), hardware counters of the step and LCNN kernels.
bash tools/round_profile.sh r06 > gpurun_out/round_profile.log 2>&1
python3 bench.py > gpurun_out/r06_bench_default_line.json 2> gpurun_out/r06_bench_default.err
bash tools/pmc_traffic.sh coif4-l14 128 2 profiles/r06_pmc_traffic_coif4-l14.json > gpurun_out/pmc_traffic.log 2>&1
cp profiles/r06_pmc_traffic_coif4-l14.json gpurun_out/
bash tools/pmc_traffic.sh stft-lcnn-eval-bf16 1024 3 profiles/r06_pmc_traffic_stft-lcnn-eval-bf16.json > gpurun_out/pmc_traffic_lcnn.log 2>&1
cp profiles/r06_pmc_traffic_stft-lcnn-eval-bf16.json gpurun_out/
bash tools/pmc_kernels.sh step "wino|conv|bn_|prelu|dil" gpurun_out/r06_pmc_step_raw.md -- python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --e2e-steps 0 --no-frontends --no-secondary > gpurun_out/pmc_step.log 2>&1
python3 tools/pmc_derive.py gpurun_out/r06_pmc_step_raw.md gpurun_out/r06_pmc_step.md "Hardware counters of the headline step's main kernels (packets-coif4 level 14 + DCNN, B = 128)"
bash tools/pmc_kernels.sh lcnn "lcnn|gemm|lstm|stft" gpurun_out/r06_pmc_lcnn_raw.md -- python3 bench.py --workload stft-lcnn-eval-bf16 --batch 1024 --steps 3 --warmup 1 --cpu-frames 0 > gpurun_out/pmc_lcnn.log 2>&1
python3 tools/pmc_derive.py gpurun_out/r06_pmc_lcnn_raw.md gpurun_out/r06_pmc_lcnn.md "Hardware counters of the LCNN bf16 evaluation forward (STFT + LCNN, B = 1024)"
tail -3 gpurun_out/round_profile.log
# validation sweeps at the same build: random geometries through the C ABI (fold, gradient sums, F(4x4) layers, wavelet packets)
# and the loss of a fixed batch under 40 optimizer steps for every shipped model geometry
python3 tools/fold_fuzz.py 60 > gpurun_out/r06_fold_fuzz.txt 2>&1
python3 tools/gradsum_fuzz.py 60 > gpurun_out/r06_gradsum_fuzz.txt 2>&1
python3 tools/wino44_fuzz.py 60 > gpurun_out/r06_wino44_fuzz.txt 2>&1
python3 tools/wpt_fuzz.py 80 > gpurun_out/r06_wpt_fuzz.txt 2>&1
: > gpurun_out/r06_overfit.txt
for w in coif4-l14 sym5-l14 coif4-l8 sym5-l8 stft; do
  echo "== $w" >> gpurun_out/r06_overfit.txt
  python3 tools/overfit_check.py $w 40 32 2>/dev/null >> gpurun_out/r06_overfit.txt
done
for f in fold_fuzz gradsum_fuzz wino44_fuzz wpt_fuzz overfit; do tail -n 2 gpurun_out/r06_$f.txt; done
