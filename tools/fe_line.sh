# front-end-only bench lines, condensed: tools/fe_line.sh <workload> [batch]
w=$1; b=${2:-128}
python bench.py --workload $w --batch $b --steps 30 --warmup 5 --cpu-frames 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$w B=$b: %.4f ms/step  launch avg %.4f ms x %.1f  frac %.3f' % (d['ms_per_step'], r['avg_launch_ms'], r['launches_per_step'], r['frac']))"
