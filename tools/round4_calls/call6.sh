#!/bin/bash
# round 4, GPU call 6: after the arena fixes (context buffers released before the trim, long-lived buffers top-down): the headline
# three-way test, the fused driver's wall clock, and knob runs (heap slack 16 GB, grid of the cached extraction kernel).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call6; mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 1500 python -m pytest tests/test_gpu_large.py -m gpu -q --timeout 1400 --durations=6 -x ) > $O/pytest_large.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
( time PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline ) > $O/bench_driver.log 2> $O/bench_driver.err
echo "bench driver rc=$?" | tee -a $O/summary.txt
for kv in DBHEAP_GB=16 CACHED=64 CACHED=16 SHORT=36; do
  ( time env PLASSHIP_TUNE_$kv PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify ) > $O/bench_$kv.log 2> $O/bench_$kv.err
  echo "bench $kv rc=$?" | tee -a $O/summary.txt
done
( time PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify ) > $O/bench_default12.log 2> $O/bench_default12.err
tail -c 800 $O/pytest_large.log; tail -c 1500 $O/bench_driver.log; echo
for f in $O/bench_*.log; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    j=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1]); print(round(j["value"]/1e6,1), round(j["ms_per_step"],2), j["roofline"]["stage_ms_per_step"])
except Exception as e: print("ERR", e)
PY
done
