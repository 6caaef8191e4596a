#!/bin/bash
# round 3, GPU call 7: after the record cache was taken out: parity + sharded + large tests, bench c5 (first run), the driver's command
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_call7; mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 2000 python -m pytest tests -m gpu -q --timeout 1700 --durations=8 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
( time timeout 600 python bench.py --config c5 ) > $O/bench_c5.log 2> $O/bench_c5.err
echo "c5 rc=$?" | tee -a $O/summary.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
tail -c 1200 $O/pytest.log; tail -c 1500 $O/bench_c5.log; tail -c 500 $O/bench_c5.err; tail -c 1200 $O/bench.log
