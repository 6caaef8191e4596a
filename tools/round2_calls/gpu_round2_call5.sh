#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c5
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/c5/pytest.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c5/prof5M -o p5 -- python $R/bench.py --pairs 5000000 --no-cpu-baseline ) > gpurun_out/c5/prof5M.log 2>&1
DB=$(find gpurun_out/c5/prof5M -name '*_results.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py "$DB" --timeline 3 > gpurun_out/c5/prof5M_summary.txt 2>&1
find gpurun_out/c5/prof5M -name '*.db' -delete
( timeout 300 python bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline ) > gpurun_out/c5/bench_c2.log 2> gpurun_out/c5/bench_c2.err
( timeout 900 python bench.py ) > gpurun_out/c5/bench_c3.log 2> gpurun_out/c5/bench_c3.err
tail -5 gpurun_out/c5/pytest.log; head -32 gpurun_out/c5/prof5M_summary.txt | cut -c1-160; tail -3 gpurun_out/c5/bench_c3.err | cut -c1-250
