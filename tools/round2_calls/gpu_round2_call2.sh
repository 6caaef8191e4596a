#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c2
export TMPDIR=/tmp
( timeout 300 tools/linepart_bench 1073741824 30 ; timeout 200 tools/linepart_bench4 1073741824 30; timeout 100 tools/linepart_bench4 67108864 30 | tail -3 ) > gpurun_out/c2/linepart.log 2>&1
( PLASS_BENCH_VERBOSE=1 PLASSHIP_POOL_STATS=1 timeout 600 python bench.py --no-cpu-baseline ) > gpurun_out/c2/bench_c3.log 2> gpurun_out/c2/bench_c3.err
( PLASS_BENCH_VERBOSE=1 timeout 300 python bench.py --pairs 5000000 --no-cpu-baseline ) > gpurun_out/c2/bench_c3_5M.log 2> gpurun_out/c2/bench_c3_5M.err
( timeout 600 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/c2/pytest_sharded.log 2>&1
cat gpurun_out/c2/linepart.log | cut -c1-230; tail -30 gpurun_out/c2/bench_c3.err; tail -3 gpurun_out/c2/pytest_sharded.log
