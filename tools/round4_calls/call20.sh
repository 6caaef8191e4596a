#!/bin/bash
# round 4, GPU call 20: 16-bit window score without the unused parts of the last 64-bit multiplication
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call20; mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 800 -x ) > $O/pytest_parity.log 2>&1
tail -3 $O/pytest_parity.log
( time PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall ) > $O/bench12.log 2> $O/bench12.err
python - "$O/bench12.log" <<'PY'
import json,sys
try:
    j=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1]); print(sys.argv[1], round(j["value"]/1e6,1), round(j["ms_per_step"],2), (j.get("verify") or {}).get("match"), [round(r["extract_ms"],1) for r in j["iterations"]])
except Exception as e: print("ERR", e)
PY
grep -o "extract [0-9.]* (short [0-9.]* wave [0-9.]*)" $O/bench12.err | tr '\n' ';'; echo
