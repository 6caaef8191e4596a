#!/bin/bash
# round 2, GPU call 22: sequences beyond 1024 windows go straight to the 48-scores tier's queue
mkdir -p gpurun_out/c22
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -x -q > gpurun_out/c22/pytest.log 2>&1
tail -3 gpurun_out/c22/pytest.log
timeout 600 python bench.py --no-cpu-baseline --steps 12 --warmup 0 > gpurun_out/c22/bench.log 2> gpurun_out/c22/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c22/bench.log").read().strip().splitlines()[-1])
print("value", round(d["value"] / 1e6, 1), "ms/step", round(d["ms_per_step"], 1), "extract_ms", [r["extract_ms"] for r in d["iterations"]])
PY
PROBE_LENGTHS=1000,1500,2500 timeout 120 python tools/extract_probe.py 3e8 2>&1 | tail -3
