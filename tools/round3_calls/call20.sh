#!/bin/bash
# GPU call 20: staged rows in the level-1 tag scatter (workgroup-private parts of the lists): parity + bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_call20; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py tests/test_gpu_sharded.py -m gpu -q -x --timeout 1200 --durations=5 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" > $O/summary.txt
B="python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall"
( timeout 600 $B ) > $O/bench_default.log 2> $O/bench_default.err
cat $O/summary.txt; tail -4 $O/pytest.log
