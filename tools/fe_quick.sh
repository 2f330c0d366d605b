# front-end-only lines of the four level-14 workloads (development helper): ms per transform and fraction of 8 TB/s
for spec in "coif4-l14-frontend 4096" "coif4-l14-frontend 128" "sym5-l14-frontend 4096" "sym5-l14-frontend 128" "haar-l14-frontend 4096" "coif4-l8-frontend 128"; do
  set -- $spec
  python3 bench.py --workload $1 --batch $2 --steps 30 --warmup 5 --cpu-frames 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 B=$2: %.4f ms  frac %.3f' % (d['ms_per_step'], r['frac']))"
done
