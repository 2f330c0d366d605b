# Hardware counters per kernel for any command: one rocprofv3 --pmc pass per counter group (counters in their own runs,
# no tracing beside them), summarised per kernel into a markdown table.
#   tools/pmc_kernels.sh <tag> <kernel name filter (regex)> <out.md> -- <program> [args...]
# Groups that the hardware cannot collect together (or counters this chip does not have) are skipped and noted.
tag=$1; filt=$2; out=$3; shift 3; [ "$1" = "--" ] && shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/pmck_$tag; rm -rf $D; mkdir -p $D
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $group --output-format csv -d $D/g$i -- "$@" > $D/g$i.out 2> $D/g$i.err || echo "group $i ($group): pass failed" >> $D/failed.txt
done <<'GROUPS'
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCC_HIT_sum TCC_MISS_sum
FETCH_SIZE
WRITE_SIZE
GROUPS
python3 tools/pmc_kernels.py $D "$filt" "$out" "$@"
