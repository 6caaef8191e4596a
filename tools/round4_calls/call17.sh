#!/bin/bash
# round 4, GPU call 17: protein tiers' lists in work classes (A/B), new cyclecheck tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call17; mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 500 -x -k "cycle" ) > $O/pytest_cyc.log 2>&1
tail -3 $O/pytest_cyc.log
( time PLASSHIP_TUNE_ASM_CLASSES=1 PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall ) > $O/bench_classes.log 2> $O/bench_classes.err
( time PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify ) > $O/bench_plain.log 2> $O/bench_plain.err
for f in $O/bench_classes.log $O/bench_plain.log; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1]); print(sys.argv[1], round(j["value"]/1e6,1), round(j["ms_per_step"],2), (j.get("verify") or {}).get("match"), j["roofline"]["stage_ms_per_step"]["assemble_stage"], [round(r["assemble_ms"],1) for r in j["iterations"]])
except Exception as e: print("ERR", e)
PY
done
