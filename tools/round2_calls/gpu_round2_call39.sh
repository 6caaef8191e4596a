#!/bin/bash
# round 2, GPU call 39: records per rep-sort bucket (PLASSHIP_TUNE_SORTBUCKET), 6 steps each
mkdir -p gpurun_out/c39
for v in 512 1024 2048 256; do
  PLASSHIP_TUNE_SORTBUCKET=$v timeout 600 python bench.py --no-cpu-baseline --steps 6 --warmup 0 > gpurun_out/c39/b.log 2> gpurun_out/c39/b.err
  python - $v <<'PY'
import json, sys
d = json.loads(open("gpurun_out/c39/b.log").read().strip().splitlines()[-1])
print("sortbucket", sys.argv[1], "ms/step", round(d["ms_per_step"], 1), "repsort_ms", [r["repsort_ms"] for r in d["iterations"]], "N_c", d["iterations"][2]["N_c"])
PY
done
