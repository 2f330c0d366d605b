# timings of tools/hbm_time.py (one process per kernel family; environment switches are read once per process)
cd $GRAFT_REPO_ROOT
out=gpurun_out/hbm_sweep.txt; : > $out
for k in conv1 pool dil; do timeout -k 10 120 python3 tools/hbm_time.py $k 2>&1 | grep -v amdgpu.ids >> $out || exit 1; done
cat $out
