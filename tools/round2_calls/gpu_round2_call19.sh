#!/bin/bash
# round 2, GPU call 19: the SHARDED code path (1-rank RCCL group, native communicator) on skewed C3-model sets: 12.5 M and 25 M reads
mkdir -p gpurun_out/c19
for p in 6250000 12500000; do
  PLASS_BENCH_FORCE_DIST=1 PLASS_BENCH_VERBOSE=1 timeout 900 python bench.py --no-cpu-baseline --pairs $p --steps 6 --warmup 0 > gpurun_out/c19/bench_sharded_$p.log 2> gpurun_out/c19/bench_sharded_$p.err
  echo "pairs $p rc $?"
  tail -c 600 gpurun_out/c19/bench_sharded_$p.err
  python - $p <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/c19/bench_sharded_%s.log" % sys.argv[1]).read().strip().splitlines()[-1])
    print("sharded 1-rank", sys.argv[1], "value", round(d["value"] / 1e6, 1), "ms/step", round(d["ms_per_step"], 1), [r["ms"] for r in d["iterations"]], d.get("sharded_mode_error"), d["config"]["parallelism"][:60])
except Exception as e:
    print("no bench line:", e)
PY
done
