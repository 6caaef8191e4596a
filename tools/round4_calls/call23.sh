#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call23; mkdir -p $O
for v in "X=1" "PLASSHIP_TUNE_KMCACHE=2" "PLASSHIP_TUNE_DBHEAP=2"; do
  for r in 1 2; do env $v PLASS_BENCH_VERBOSE=1 timeout 300 python bench.py --config c2 --no-wall --no-cpu-baseline > $O/bench_c2_${v}_$r.log 2> $O/bench_c2_${v}_$r.err; python - "$O/bench_c2_${v}_$r.log" <<'PY'
import json,sys
j=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1]); print(sys.argv[1], round(j["ms_per_step"],3), round(j["value"]/1e6,1), [round(r["extract_ms"],2) for r in j["iterations"]])
PY
  done
done
grep -o "extract [0-9.]* (short [0-9.]* wave [0-9.]*)" $O/bench_c2_X=1_2.err | tr '\n' ';'; echo
grep -o "extract [0-9.]* (short [0-9.]* wave [0-9.]*)" $O/bench_c2_PLASSHIP_TUNE_KMCACHE=2_2.err | tr '\n' ';'; echo
