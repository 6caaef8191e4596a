#!/bin/bash
# round 2, GPU call 33: group kernel with alternating cursors (one barrier less), aggregation sort with wave-local stages
mkdir -p gpurun_out/c33
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py tests/test_gpu_sharded.py -m gpu -x -q > gpurun_out/c33/pytest.log 2>&1
tail -3 gpurun_out/c33/pytest.log
timeout 600 python bench.py --no-cpu-baseline --steps 12 --warmup 0 > gpurun_out/c33/bench.log 2> gpurun_out/c33/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c33/bench.log").read().strip().splitlines()[-1])
print("value", round(d["value"] / 1e6, 1), "ms/step", round(d["ms_per_step"], 1), "group_ms", [r["group_ms"] for r in d["iterations"]], "repsort_ms", [r["repsort_ms"] for r in d["iterations"]])
PY
