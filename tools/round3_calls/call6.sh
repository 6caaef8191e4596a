#!/bin/bash
# round 3, GPU call 6: record cache v3 (immutable store + alive bits), A/B runs of the bench (cache off, tier-0 occupancy, 1024-thread group kernel)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_call6; mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_large.py -m gpu -q --timeout 1400 --durations=6 -k "record_cache or chained or stale or full_size or native or headline or large_chain" ) > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
for v in "PLASSHIP_KMER_CACHE=0" "PLASSHIP_TUNE_TIER0_WPE=5" "PLASSHIP_TUNE_GROUP_1024=0"; do
  ( env $v timeout 600 python bench.py --gpus 1 --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify ) > $O/bench_$(echo $v | tr '=' '_').log 2>> $O/bench.err
  echo "$v rc=$?" | tee -a $O/summary.txt
done
tail -c 1500 $O/pytest.log; for f in $O/bench*.log; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],1),'ms/step', {k:round(v,1) for k,v in d['roofline']['stage_ms_per_step'].items()})
    print([ (r['iteration'], round(r['ms'])) for r in d['iterations'][:12]])
except Exception as e: print('unreadable', e)
PY
done; tail -c 400 $O/bench.err
