#!/bin/bash
# round 4, GPU call 2: the parity / large tests on the reworked cache writes (incl. the new > 4 GiB fixture), the driver's bench
# command, and the 12-iteration chain with the selected-window cache switched off (PLASSHIP_TUNE_KMCACHE=2) for the A/B rows.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call2; mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 1500 python -m pytest tests/test_gpu_large.py tests/test_gpu_parity.py tests/test_gpu_orfs.py -m gpu -q --timeout 1400 --durations=8 -x ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
( time PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-wall ) > $O/bench_driver.log 2> $O/bench_driver.err
echo "bench driver rc=$?" | tee -a $O/summary.txt
( time PLASSHIP_TUNE_KMCACHE=2 PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify ) > $O/bench_nocache.log 2> $O/bench_nocache.err
echo "bench nocache rc=$?" | tee -a $O/summary.txt
( time PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify ) > $O/bench_cache12.log 2> $O/bench_cache12.err
tail -c 1800 $O/pytest_gpu.log; tail -c 700 $O/bench_driver.log; echo; grep N_k $O/bench_nocache.err | awk '{print $4,$5,$6,$7,$8,$9,$10}'; grep N_k $O/bench_cache12.err | awk '{print $4,$5,$6,$7,$8,$9,$10}'
