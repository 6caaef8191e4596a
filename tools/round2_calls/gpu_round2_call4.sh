#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c4
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/c4/pytest.log 2>&1
# kernel trace of the line path on the C3 model at 10 M reads (12 timed iterations after a full warm-up traversal)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c4/prof5M -o p5 -- python $R/bench.py --pairs 5000000 --no-cpu-baseline ) > gpurun_out/c4/prof5M.log 2>&1
DB=$(find gpurun_out/c4/prof5M -name '*_results.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py "$DB" --timeline 3 > gpurun_out/c4/prof5M_summary.txt 2>&1
find gpurun_out/c4/prof5M -name '*.db' -size +40M -delete
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c4/profc2 -o pc2 -- python $R/bench.py --config c2 --no-cpu-baseline ) > gpurun_out/c4/profc2.log 2>&1
DB=$(find gpurun_out/c4/profc2 -name '*_results.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py "$DB" --timeline 2 > gpurun_out/c4/profc2_summary.txt 2>&1
find gpurun_out/c4/profc2 -name '*.db' -size +40M -delete
( PLASS_BENCH_VERBOSE=1 PLASSHIP_POOL_STATS=1 timeout 900 python bench.py --no-cpu-baseline ) > gpurun_out/c4/bench_c3.log 2> gpurun_out/c4/bench_c3.err
tail -5 gpurun_out/c4/pytest.log; head -40 gpurun_out/c4/prof5M_summary.txt | cut -c1-160; tail -12 gpurun_out/c4/bench_c3.err | cut -c1-250
