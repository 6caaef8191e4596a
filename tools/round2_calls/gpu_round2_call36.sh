#!/bin/bash
# round 2, GPU call 36: pair re-scoring forced to 8 wavefronts per SIMD (64 VGPRs, 32 bytes of scratch)
mkdir -p gpurun_out/c36
timeout 600 python bench.py --no-cpu-baseline --steps 12 --warmup 0 > gpurun_out/c36/bench.log 2> gpurun_out/c36/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c36/bench.log").read().strip().splitlines()[-1])
print("value", round(d["value"] / 1e6, 1), "ms/step", round(d["ms_per_step"], 1), "rescore_ms", [r["rescore_ms"] for r in d["iterations"]])
PY
