#!/bin/bash
# dev: one bench step per value of an environment switch: tools/dev_sweep.sh VAR "v1 v2 ..." [workload] [class]
var=$1; vals=$2; wl=${3:-coif4-l14}; cls=${4:-conv_wgrad}
for e in $vals; do
  env $var=$e python3 bench.py --workload $wl --steps 8 --warmup 4 --cpu-frames 0 --e2e-steps 0 --no-frontends --no-secondary 2>/dev/null \
    | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$e', round(d['ms_per_step'],3), {k:round(v['ms_per_step'],3) for k,v in d['classes'].items() if k in '$cls'.split(',')})"
done
