# per-launch kernel timeline of one headline step (development helper)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/step; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 ${BENCH_ARGS} > $O/bench.json 2> $O/bench.err
python3 - <<'PY'
import csv,glob,re
f=glob.glob('gpurun_out/step/prof/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'wpt2_top' in r['Kernel_Name'] or 'stft_mfma' in r['Kernel_Name']]
s=idx[-1]; tot=0
for r in rows[s:]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6; tot+=d
    if d>0.2:
        m=re.search(r'(\w+_kernel(<[^>]*>)?)',r['Kernel_Name'])
        print(f"{d:8.3f} ms  {m.group(1) if m else r['Kernel_Name'][:60]}  grid={r.get('Grid_Size_X')} wg={r.get('Workgroup_Size_X')} vgpr={r.get('VGPR_Count')}")
print('total',tot)
PY
