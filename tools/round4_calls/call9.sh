#!/bin/bash
# round 4, GPU call 9: A/B of the two-pass thread-per-pair rescoring (self hits in their own launch), parity of the rescore paths.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call9; mkdir -p $O
export PYTHONUNBUFFERED=1
( time PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall ) > $O/bench_split.log 2> $O/bench_split.err
( time PLASSHIP_TUNE_RESCORE_SPLIT=2 PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify ) > $O/bench_nosplit.log 2> $O/bench_nosplit.err
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 800 -x ) > $O/pytest_parity.log 2>&1
tail -3 $O/pytest_parity.log
for f in $O/bench_split.log $O/bench_nosplit.log; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1]); print(sys.argv[1], round(j["value"]/1e6,1), round(j["ms_per_step"],2), (j.get("verify") or {}).get("match"), j["roofline"]["stage_ms_per_step"]["rescore_stage"], [round(r["rescore_ms"],1) for r in j["iterations"]])
except Exception as e: print("ERR", e)
PY
done
