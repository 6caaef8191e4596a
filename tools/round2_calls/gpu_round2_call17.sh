#!/bin/bash
# round 2, GPU call 17: 8-scores-per-lane extraction tier on/off
mkdir -p gpurun_out/c17
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -x -q > gpurun_out/c17/pytest.log 2>&1
tail -3 gpurun_out/c17/pytest.log
for v in 1 0; do
  PLASSHIP_TIERM=$v timeout 600 python bench.py --no-cpu-baseline --steps 12 --warmup 0 > gpurun_out/c17/bench_tierm_$v.log 2> gpurun_out/c17/bench_tierm_$v.err
  python - $v <<'PY'
import json, sys
d = json.loads(open("gpurun_out/c17/bench_tierm_%s.log" % sys.argv[1]).read().strip().splitlines()[-1])
print("tierM", sys.argv[1], "value", round(d["value"] / 1e6, 1), "ms/step", round(d["ms_per_step"], 1), "extract_ms", [r["extract_ms"] for r in d["iterations"]])
PY
  PLASSHIP_TIERM=$v PROBE_LENGTHS=300,400,500,700 timeout 120 python tools/extract_probe.py 3e8 2>&1 | tail -4 | sed "s/^/tierM=$v /"
done
