# HBM traffic counters of one bench workload, one rocprofv3 --pmc pass per counter group (MI355X_MICROARCH.md, HBM:
# FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2; counters in their own runs, no tracing beside them):
#   tools/pmc_traffic.sh <workload> <batch> <steps> <out json>
#   pass 1  FETCH_SIZE                      (the guide's counter; on gfx950 it tallies a 128-byte request as 64)
#   pass 2  WRITE_SIZE
#   pass 3  TCC_EA0_RDREQ_{32B,64B,128B}_sum  the read requests by size: the per-kernel calibration of pass 1
#   pass 4  TCC_EA0_RDREQ_DRAM_32B_sum         the same bytes counted in 32-byte units by the chip itself (cross-check)
# rocprofv3 gets the program itself after `--` (python3 bench.py ...), no wrappers.
w=$1; b=$2; st=$3; out=$4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1)); d=gpurun_out/pmc_pass$i
  rm -rf $d && mkdir -p $d
  echo "$c" > $d/counters.txt
  timeout -k 10 500 rocprofv3 --pmc $c --output-format csv -d $d -- python3 bench.py --workload $w --batch $b --steps $st --warmup 1 --cpu-frames 0 --e2e-steps 0 --no-frontends --no-secondary > $d/bench.json 2> $d/bench.err || { echo "pass $i ($c) failed"; [ $i -le 2 ] && exit 1; }
done
python3 tools/pmc_json.py $w $b $out
