#!/bin/bash
# round 2, GPU call 20: rep-side partitions with direct line writes on/off
mkdir -p gpurun_out/c20
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -x -q > gpurun_out/c20/pytest.log 2>&1
tail -3 gpurun_out/c20/pytest.log
for v in 1 0; do
  PLASSHIP_DIRECT_LINES=$v timeout 600 python bench.py --no-cpu-baseline --steps 12 --warmup 0 > gpurun_out/c20/bench_direct_$v.log 2> gpurun_out/c20/bench_direct_$v.err
  python - $v <<'PY'
import json, sys
d = json.loads(open("gpurun_out/c20/bench_direct_%s.log" % sys.argv[1]).read().strip().splitlines()[-1])
print("direct", sys.argv[1], "value", round(d["value"] / 1e6, 1), "ms/step", round(d["ms_per_step"], 1), "repsort_ms", [r["repsort_ms"] for r in d["iterations"]])
PY
done
