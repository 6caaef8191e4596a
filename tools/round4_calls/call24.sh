#!/bin/bash
# round 4, GPU call 24: cache-line address requested at the start of a sequence instead of at its end (c2 regression of the wave tiers)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call24; mkdir -p $O
for r in 1 2 3; do PLASS_BENCH_VERBOSE=1 timeout 300 python bench.py --config c2 --no-wall --no-cpu-baseline > $O/bench_c2_$r.log 2> $O/bench_c2_$r.err; python - "$O/bench_c2_$r.log" <<'PY'
import json,sys
j=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1]); print(sys.argv[1], round(j["ms_per_step"],3), round(j["value"]/1e6,1), (j.get("verify") or {}).get("match"), [round(r["extract_ms"],2) for r in j["iterations"]])
PY
done
( time PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall ) > $O/bench12.log 2> $O/bench12.err
python - "$O/bench12.log" <<'PY'
import json,sys
j=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1]); print(sys.argv[1], round(j["value"]/1e6,1), round(j["ms_per_step"],2), (j.get("verify") or {}).get("match"), [round(r["extract_ms"],1) for r in j["iterations"]])
PY
grep -o "extract [0-9.]* (short [0-9.]* wave [0-9.]*)" $O/bench12.err | tr '\n' ';'; echo
