#!/bin/bash
# round 4, GPU call 29: grid of the thread-per-sequence extraction after the residency cap (PLASSHIP_TUNE_SHORT = workgroups per CU; default 72 on large sets)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call29; mkdir -p $O
for v in 288 576; do
  PLASSHIP_TUNE_SHORT=$v PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 6 --warmup 0 --no-cpu-baseline --no-wall --no-verify > $O/bench_short$v.log 2> $O/bench_short$v.err
  echo SHORT=$v $(grep -o "extract [0-9.]* (short [0-9.]* wave [0-9.]*)" $O/bench_short$v.err | tr '\n' ';')
done
