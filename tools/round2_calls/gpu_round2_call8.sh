#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c8
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/c8/pytest.log 2>&1
( timeout 600 python tools/extract_probe.py 4e8 ) > gpurun_out/c8/extract_probe.log 2>&1
( timeout 300 python bench.py --pairs 5000000 --no-cpu-baseline ) > gpurun_out/c8/bench_c3_5M.log 2> gpurun_out/c8/bench_c3_5M.err
tail -4 gpurun_out/c8/pytest.log; cat gpurun_out/c8/extract_probe.log
