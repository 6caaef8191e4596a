#!/bin/bash
# round 2, GPU call 18: group kernel with the next bucket's records fetched behind the last phase
mkdir -p gpurun_out/c18
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -x -q > gpurun_out/c18/pytest.log 2>&1
tail -3 gpurun_out/c18/pytest.log
timeout 600 python bench.py --no-cpu-baseline --steps 12 --warmup 0 > gpurun_out/c18/bench.log 2> gpurun_out/c18/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c18/bench.log").read().strip().splitlines()[-1])
print("value", round(d["value"] / 1e6, 1), "ms/step", round(d["ms_per_step"], 1), "group_ms", [r["group_ms"] for r in d["iterations"]], "extract_ms", [r["extract_ms"] for r in d["iterations"]])
PY
timeout 300 python bench.py --config c2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2: value', round(d['value']/1e6,1), 'ms/step', round(d['ms_per_step'],2), d['roofline']['stage_ms_per_step'])"
