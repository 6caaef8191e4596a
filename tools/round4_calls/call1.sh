#!/bin/bash
# round 4, GPU call 1: the whole GPU suite on the strand-tie rule, the late overflow check, the header concatdbs and the
# selected-window cache; then the 12-iteration chain at 50 M reads with and without the cache (digests verified in both).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call1; mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 1400 --durations=12 -x ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
( time PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall ) > $O/bench_cache.log 2> $O/bench_cache.err
echo "bench cache rc=$?" | tee -a $O/summary.txt
( time PLASSHIP_TUNE_KMCACHE=0 PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall ) > $O/bench_nocache.log 2> $O/bench_nocache.err
echo "bench nocache rc=$?" | tee -a $O/summary.txt
tail -c 2500 $O/pytest_gpu.log; tail -c 1500 $O/bench_cache.log; echo; tail -c 600 $O/bench_nocache.log
