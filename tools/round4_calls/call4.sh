#!/bin/bash
# round 4, GPU call 4 (call 3 again with the heap slack capped at 8 GB: 3x the data ran the 50 M-read chain out of memory): the whole GPU suite on the append-only DB heap (assemble.hip: buildOutputDB), the 12-iteration chain with and
# without it (PLASSHIP_TUNE_DBHEAP=2), the driver's command, and --config c5 with its new verification traversal.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call4; mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 1400 --durations=8 -x ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
( time PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall ) > $O/bench_heap12.log 2> $O/bench_heap12.err
echo "bench heap rc=$?" | tee -a $O/summary.txt
( time PLASSHIP_TUNE_DBHEAP=2 PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify ) > $O/bench_noheap12.log 2> $O/bench_noheap12.err
echo "bench noheap rc=$?" | tee -a $O/summary.txt
( time PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-wall ) > $O/bench_driver.log 2> $O/bench_driver.err
echo "bench driver rc=$?" | tee -a $O/summary.txt
( time timeout 900 python bench.py --config c5 --no-cpu-baseline --steps 10 --warmup 0 ) > $O/bench_c5.log 2> $O/bench_c5.err
echo "bench c5 rc=$?" | tee -a $O/summary.txt
tail -c 1500 $O/pytest_gpu.log; grep N_k $O/bench_heap12.err | sed 's/.*extended=/extended=/' | cut -c1-120; echo; grep N_k $O/bench_noheap12.err | sed 's/.*extended=/extended=/' | cut -c1-120; tail -c 400 $O/bench_driver.log; echo; tail -c 1200 $O/bench_c5.log
