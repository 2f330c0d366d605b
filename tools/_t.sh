set -e
timeout -k 10 600 python -m pytest tests/test_nn_gpu.py tests/test_dcnn_gpu.py tests/test_lcnn_gpu.py -q -x 2>&1 | tail -2
python bench.py --no-secondary --steps 10 --warmup 5 > gpurun_out/b_c11.json 2> gpurun_out/b_c11.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/b_c11.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],3), {k:round(v["ms_per_step"],3) for k,v in d["classes"].items()})
PY
