# one condensed line per bench run: tools/bench_line.sh <workload> [extra bench args]
w=$1; shift
python bench.py --workload $w --steps 10 --warmup 5 --cpu-frames 0 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['classes']
print('$w: %.3f ms/step  %.0f frames/s  ' % (d['ms_per_step'], d['value']) + ' '.join('%s %.2f' % (k, v['ms_per_step']) for k, v in c.items()))"
