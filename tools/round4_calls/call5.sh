#!/bin/bash
# round 4, GPU call 5: incremental extraction of the extended sequences in same-seed iterations (kmermatch.hip section 2d): the
# large / parity tests, the 12-iteration chain with and without it (PLASSHIP_TUNE_KMINCR=2), the driver's command.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call5; mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 1500 python -m pytest tests/test_gpu_large.py tests/test_gpu_parity.py tests/test_gpu_chain_cli.py -m gpu -q --timeout 1400 --durations=6 -x ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
( time PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall ) > $O/bench_incr12.log 2> $O/bench_incr12.err
echo "bench incr rc=$?" | tee -a $O/summary.txt
( time PLASSHIP_TUNE_KMINCR=2 PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify ) > $O/bench_noincr12.log 2> $O/bench_noincr12.err
echo "bench noincr rc=$?" | tee -a $O/summary.txt
( time PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline ) > $O/bench_driver.log 2> $O/bench_driver.err
echo "bench driver rc=$?" | tee -a $O/summary.txt
tail -c 1200 $O/pytest_gpu.log; grep N_k $O/bench_incr12.err | awk '{print $4,$5,$6,$7,$8,$9,$10,$11}'; grep N_k $O/bench_noincr12.err | awk '{print $4,$5,$6,$7,$8,$9,$10,$11}'; tail -c 900 $O/bench_driver.log
