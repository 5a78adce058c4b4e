#!/bin/bash
# The banded block chain on the GPU box: its tests, then BASELINE config 5's per-GPU share (bench.py --workload long).
# Output under gpurun_out/<tag>_band_*.
set -o pipefail
mkdir -p gpurun_out
tag=${1:-r04}
timeout -k 10 600 python -m pytest tests/test_gpu_band.py -x -q > gpurun_out/${tag}_band_tests.log 2>&1 || { tail -30 gpurun_out/${tag}_band_tests.log; exit 1; }
tail -3 gpurun_out/${tag}_band_tests.log
timeout -k 10 300 python bench.py --workload long --steps 6 --warmup 2 --no-cpu > gpurun_out/${tag}_band.json 2> gpurun_out/${tag}_band.err || { tail -20 gpurun_out/${tag}_band.err; exit 1; }
python - <<PY
import json
l = json.loads(open("gpurun_out/${tag}_band.json").read().strip().splitlines()[-1])
print("ms_per_step", l["ms_per_step"], "value", l["value"], "waves_per_cu", l["roofline"].get("band_waves_per_cu"), "lds", l["roofline"].get("band_lds_per_wave"))
PY
