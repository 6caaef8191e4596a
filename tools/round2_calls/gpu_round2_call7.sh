#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c7
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/c7/pytest.log 2>&1
( timeout 600 python tools/extract_probe.py 4e8 ) > gpurun_out/c7/extract_probe.log 2>&1
( timeout 300 python bench.py --pairs 5000000 --no-cpu-baseline ) > gpurun_out/c7/bench_c3_5M.log 2> gpurun_out/c7/bench_c3_5M.err
( timeout 900 python bench.py --no-cpu-baseline ) > gpurun_out/c7/bench_c3.log 2> gpurun_out/c7/bench_c3.err
tail -4 gpurun_out/c7/pytest.log; cat gpurun_out/c7/extract_probe.log; tail -3 gpurun_out/c7/bench_c3.err | cut -c1-250
