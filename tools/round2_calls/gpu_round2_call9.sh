#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c9
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/c9/pytest.log 2>&1
for SK in 0 1 2 4 8 16 32 64 127; do
  echo "== skip $SK"; PLASSHIP_DBG_EXTRACT_SKIP=$SK PROBE_LENGTHS=100,250,700 timeout 200 python tools/extract_probe.py 3e8 | tail -3
done > gpurun_out/c9/extract_skip.log 2>&1
( timeout 900 python bench.py --no-cpu-baseline ) > gpurun_out/c9/bench_c3.log 2> gpurun_out/c9/bench_c3.err
tail -4 gpurun_out/c9/pytest.log; cat gpurun_out/c9/extract_skip.log; tail -2 gpurun_out/c9/bench_c3.err | cut -c1-200
