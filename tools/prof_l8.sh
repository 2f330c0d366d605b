cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/l8; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 10 --warmup 2 --cpu-frames 0 --workload coif4-l8 > $O/bench.json 2> $O/bench.err
python3 - <<'PY'
import csv,glob,re,collections,json
f=glob.glob('gpurun_out/l8/prof/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'wpt_fused' in r['Kernel_Name']]
s,e=idx[-3],idx[-2]
agg=collections.defaultdict(float); cnt=collections.Counter(); tot=0
for r in rows[s:e]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6; tot+=d
    m=re.search(r'(\w+_kernel)',r['Kernel_Name']); k=m.group(1) if m else 'other'; agg[k]+=d; cnt[k]+=1
wall=(int(rows[e]['Start_Timestamp'])-int(rows[s]['Start_Timestamp']))/1e6
print('launches',e-s,'kernel ms',round(tot,3),'wall ms',round(wall,3), json.load(open('gpurun_out/l8/bench.json'))['ms_per_step'])
for k,v in sorted(agg.items(), key=lambda kv:-kv[1])[:16]: print(f"{v:7.3f} ms x{cnt[k]:3d} {k}")
PY
