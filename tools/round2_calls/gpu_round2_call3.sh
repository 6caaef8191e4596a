#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c3
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 ) > gpurun_out/c3/pytest.log 2>&1
( timeout 300 python bench.py --pairs 5000000 --no-cpu-baseline ) > gpurun_out/c3/bench_c3_5M.log 2> gpurun_out/c3/bench_c3_5M.err
( PLASSHIP_LEGACY_PARTITION=1 timeout 300 python bench.py --pairs 5000000 --no-cpu-baseline ) > gpurun_out/c3/bench_c3_5M_legacy.log 2> gpurun_out/c3/bench_c3_5M_legacy.err
( PLASS_BENCH_VERBOSE=1 PLASSHIP_POOL_STATS=1 timeout 900 python bench.py --no-cpu-baseline ) > gpurun_out/c3/bench_c3.log 2> gpurun_out/c3/bench_c3.err
( timeout 300 python bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline ) > gpurun_out/c3/bench_c2.log 2> gpurun_out/c3/bench_c2.err
tail -15 gpurun_out/c3/pytest.log; tail -4 gpurun_out/c3/bench_c3_5M.err; tail -25 gpurun_out/c3/bench_c3.err
