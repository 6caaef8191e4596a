#!/bin/bash
# round 2, GPU call 25: phase-split build (H, the in-tree library) against F and D; parity; 12-step bench
mkdir -p gpurun_out/c25
for v in H F D; do
  echo "== variant $v"
  PLASSHIP_LIB=$PWD/plass_amd/variants/lib$v.so PROBE_LENGTHS=100,250,400,700,1000,1500 timeout 150 python tools/extract_probe.py 3e8 2>&1 | tail -6
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -x -q > gpurun_out/c25/pytest.log 2>&1
tail -3 gpurun_out/c25/pytest.log
timeout 600 python bench.py --no-cpu-baseline --steps 12 --warmup 0 > gpurun_out/c25/bench.log 2> gpurun_out/c25/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c25/bench.log").read().strip().splitlines()[-1])
print("value", round(d["value"] / 1e6, 1), "ms/step", round(d["ms_per_step"], 1), "extract_ms", [r["extract_ms"] for r in d["iterations"]])
PY
