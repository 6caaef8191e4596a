#!/bin/bash
# round 2, GPU call 27: 48-scores tier at 3 (in-tree build H2) or 4 (V4) wavefronts per SIMD; parity and bench with H2
mkdir -p gpurun_out/c27
for v in H2 V4; do
  echo "== variant $v"
  PLASSHIP_LIB=$PWD/plass_amd/variants/lib$v.so PROBE_LENGTHS=100,250,1500,2500 timeout 150 python tools/extract_probe.py 3e8 2>&1 | tail -4
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -x -q > gpurun_out/c27/pytest.log 2>&1
tail -3 gpurun_out/c27/pytest.log
timeout 600 python bench.py --no-cpu-baseline --steps 12 --warmup 0 > gpurun_out/c27/bench.log 2> gpurun_out/c27/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c27/bench.log").read().strip().splitlines()[-1])
print("value", round(d["value"] / 1e6, 1), "ms/step", round(d["ms_per_step"], 1), "extract_ms", [r["extract_ms"] for r in d["iterations"]])
PY
