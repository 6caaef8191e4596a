#!/bin/bash
# round 2, GPU call 21: direct line writes on the k-mer side as well (experiment)
mkdir -p gpurun_out/c21
for v in 1 0; do
  PLASSHIP_DIRECT_LINES_K=$v timeout 600 python bench.py --no-cpu-baseline --steps 6 --warmup 0 > gpurun_out/c21/bench_directk_$v.log 2> gpurun_out/c21/bench_directk_$v.err
  python - $v <<'PY'
import json, sys
d = json.loads(open("gpurun_out/c21/bench_directk_%s.log" % sys.argv[1]).read().strip().splitlines()[-1])
print("directK", sys.argv[1], "value", round(d["value"] / 1e6, 1), "ms/step", round(d["ms_per_step"], 1), "partition_ms", [r["partition_ms"] for r in d["iterations"]], "N_c", [r["N_c"] for r in d["iterations"]][:3])
PY
done
