#!/bin/bash
# round 3, GPU call 5: record cache with the location map (no kill records), the barrier fix of the direct-line partition: full GPU suite + bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_call5; mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 2400 python -m pytest tests -m gpu -q --timeout 1700 --durations=12 ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
tail -c 3000 $O/pytest_gpu.log; tail -c 1500 $O/bench.log; tail -c 600 $O/bench.err
