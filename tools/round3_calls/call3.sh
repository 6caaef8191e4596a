#!/bin/bash
# round 3, GPU call 3: backtrace of the abort in test_full_size_sharded_equals_single_and_properties, then the record cache with
# half-size arenas: cache parity test, three-way cross-check at 50 M reads, bench.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_call3; mkdir -p $O
export PYTHONUNBUFFERED=1
( timeout 600 /opt/rocm/bin/rocgdb -batch -ex "handle SIGUSR1 nostop noprint" -ex run -ex bt -ex "thread apply all bt 14" --args python -m pytest tests/test_gpu_sharded.py -m gpu -q --timeout 400 -k test_full_size ) > $O/gdb.log 2>&1
echo "gdb rc=$?" | tee -a $O/summary.txt
( time timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "record_cache or chained or stale" ) > $O/cache.log 2>&1
echo "cache rc=$?" | tee -a $O/summary.txt
( time timeout 1800 python -m pytest tests/test_gpu_large.py -m gpu -q --timeout 1700 -k "headline" ) > $O/large.log 2>&1
echo "headline rc=$?" | tee -a $O/summary.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -n "SIGABRT\|#[0-9]" $O/gdb.log | head -60; tail -c 1500 $O/cache.log; tail -c 2500 $O/large.log; tail -c 1800 $O/bench.log; tail -c 600 $O/bench.err
