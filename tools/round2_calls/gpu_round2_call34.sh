#!/bin/bash
# round 2, GPU call 34: pair re-scoring with the next 16 residues requested ahead
mkdir -p gpurun_out/c34
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -x -q > gpurun_out/c34/pytest.log 2>&1
tail -3 gpurun_out/c34/pytest.log
timeout 600 python bench.py --no-cpu-baseline --steps 12 --warmup 0 > gpurun_out/c34/bench.log 2> gpurun_out/c34/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c34/bench.log").read().strip().splitlines()[-1])
print("value", round(d["value"] / 1e6, 1), "ms/step", round(d["ms_per_step"], 1), "rescore_ms", [r["rescore_ms"] for r in d["iterations"]])
PY
