#!/bin/bash
# three traversals of the 12-iteration chain in one process: the arena must not fragment, times must repeat
mkdir -p gpurun_out/soak
PLASSHIP_POOL_STATS=1 timeout 900 python bench.py --no-cpu-baseline --steps 36 --warmup 0 > gpurun_out/soak/bench.log 2> gpurun_out/soak/bench.err
tail -2 gpurun_out/soak/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/soak/bench.log").read().strip().splitlines()[-1])
ms = [round(r["ms"], 1) for r in d["iterations"]]
print("value", round(d["value"] / 1e6, 1), "ms/step", round(d["ms_per_step"], 1))
for c in range(3): print("chain", c, ms[12 * c:12 * c + 12])
PY
