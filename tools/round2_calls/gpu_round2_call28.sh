#!/bin/bash
# round 2, GPU call 28: 48-scores tier at 4 wavefronts per SIMD, assembleBigKernel with four wavefronts per workgroup: full suite + bench
mkdir -p gpurun_out/c28
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c28/pytest.log 2>&1
tail -3 gpurun_out/c28/pytest.log
timeout 600 python bench.py --no-cpu-baseline --steps 12 --warmup 0 > gpurun_out/c28/bench.log 2> gpurun_out/c28/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c28/bench.log").read().strip().splitlines()[-1])
print("value", round(d["value"] / 1e6, 1), "ms/step", round(d["ms_per_step"], 1), "extract_ms", [r["extract_ms"] for r in d["iterations"]], "assemble_ms", [r["assemble_ms"] for r in d["iterations"]])
print(d["roofline"]["stage_ms_per_step"])
PY
