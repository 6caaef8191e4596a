#!/bin/bash
# round 4, GPU call 34: rescoring with every lane walking its own wraps (one scoring round per wavefront instead of one per wrap index), parity
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call34; mkdir -p $O
export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 500 -x ) > $O/pytest_parity.log 2>&1
tail -2 $O/pytest_parity.log
PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall > $O/bench_wrap.log 2> $O/bench_wrap.err
PLASSHIP_TUNE_RESCORE_WPE=4 PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify > $O/bench_wrap_wpe4.log 2> $O/bench_wrap_wpe4.err
for f in $O/bench_wrap.log $O/bench_wrap_wpe4.log; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1]); print(sys.argv[1], round(j["value"]/1e6,1), round(j["ms_per_step"],2), (j.get("verify") or {}).get("match"), j["roofline"]["stage_ms_per_step"]["rescore_stage"], [round(r["rescore_ms"],1) for r in j["iterations"]])
except Exception as e: print("ERR", e)
PY
done
