#!/bin/bash
# first GPU call of round 2: parity suite (incl. row N2), the line-store partition microbenchmark, and the bench on the headline config
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c1
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/c1/pytest.log 2>&1
( timeout 300 tools/linepart_bench 67108864 30 ; timeout 300 tools/linepart_bench 1073741824 30 ) > gpurun_out/c1/linepart.log 2>&1
( timeout 900 python bench.py ) > gpurun_out/c1/bench_c3.log 2> gpurun_out/c1/bench_c3.err
( timeout 300 python bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline ) > gpurun_out/c1/bench_c2.log 2> gpurun_out/c1/bench_c2.err
tail -3 gpurun_out/c1/pytest.log; tail -12 gpurun_out/c1/linepart.log; head -c 1500 gpurun_out/c1/bench_c3.log; tail -5 gpurun_out/c1/bench_c3.err
