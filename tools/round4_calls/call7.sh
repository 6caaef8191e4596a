#!/bin/bash
# round 4, GPU call 7: where the fused driver's wall clock goes (pool statistics, per-iteration time stamps, heaps / cache off), and the
# A/B of the self hits on the 16-lane rescore kernel.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call7; mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 900 python tools/chain_wall_probe.py ) > $O/chain_wall_probe.log 2>&1
( time PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall ) > $O/bench_self16.log 2> $O/bench_self16.err
( time PLASSHIP_TUNE_RESCORE_SELF=2 PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify ) > $O/bench_self1.log 2> $O/bench_self1.err
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 800 -x ) > $O/pytest_parity.log 2>&1
tail -c 6000 $O/chain_wall_probe.log; tail -3 $O/pytest_parity.log
for f in $O/bench_self16.log $O/bench_self1.log; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1]); print(sys.argv[1], round(j["value"]/1e6,1), round(j["ms_per_step"],2), (j.get("verify") or {}).get("match"), j["roofline"]["stage_ms_per_step"]["rescore_stage"], [round(r["rescore_ms"],1) for r in j["iterations"]])
except Exception as e: print("ERR", e)
PY
done
