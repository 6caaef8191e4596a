#!/bin/bash
# round 4, GPU call 18: fewer resident wavefronts in the thread-per-sequence extraction (open write lines vs L2)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call18; mkdir -p $O
export PYTHONUNBUFFERED=1
for kb in 0 4 8 16; do
  ( time PLASSHIP_TUNE_SHORT_PAD_KB=$kb PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 6 --warmup 0 --no-cpu-baseline --no-wall --no-verify ) > $O/bench_pad$kb.log 2> $O/bench_pad$kb.err
  python - "$O/bench_pad$kb.log" <<'PY'
import json,sys
try:
    j=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1]); print(sys.argv[1], round(j["ms_per_step"],2), [round(r["extract_ms"],1) for r in j["iterations"]])
except Exception as e: print("ERR", e)
PY
  grep -h "extract (short" $O/bench_pad$kb.err | head -3 | cut -c1-200
done
