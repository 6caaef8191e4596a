#!/bin/bash
# round 2, GPU call 16: 4-scores-per-lane extraction tier on/off; failure protocol test; full suite
mkdir -p gpurun_out/c16
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py tests/test_gpu_sharded.py -m gpu -x -q > gpurun_out/c16/pytest.log 2>&1
tail -4 gpurun_out/c16/pytest.log
for v in 1 0; do
  PLASSHIP_TIER0=$v timeout 600 python bench.py --no-cpu-baseline --steps 8 --warmup 0 > gpurun_out/c16/bench_tier0_$v.log 2> gpurun_out/c16/bench_tier0_$v.err
  python - $v <<'PY'
import json, sys
d = json.loads(open("gpurun_out/c16/bench_tier0_%s.log" % sys.argv[1]).read().strip().splitlines()[-1])
print("tier0", sys.argv[1], "value", round(d["value"] / 1e6, 1), "ms/step", round(d["ms_per_step"], 1), "extract_ms", [r["extract_ms"] for r in d["iterations"]])
PY
  PLASSHIP_TIER0=$v PROBE_LENGTHS=100,250,400,700 timeout 120 python tools/extract_probe.py 3e8 2>&1 | tail -6 | sed "s/^/tier0=$v /"
done
