#!/bin/bash
# round 4: per-query clock64() instrumentation of the thread-per-query nucleotide extension kernel (PLASSHIP_PROBE_NUCL).  The instrumentation was
# TEMPORARY code in assemble.hip, removed again before the kernel's rewrite was committed; this script is kept as the record of how the figures under
# "probe" in profiles/r04_ab_knobs.txt were taken.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_probe; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_nucl.py -m gpu -q --timeout 800 -x -k "nucl or guided or strand or hairpin or penguin" ) > $O/pytest_nucl.log 2>&1
tail -3 $O/pytest_nucl.log
PLASSHIP_PROBE_NUCL=1 timeout 600 python bench.py --config c5 --steps 10 --warmup 0 --no-cpu-baseline > $O/probe.log 2> $O/probe.err
grep -c PROBE $O/probe.err
python - "$O/probe.log" <<'PY'
import json,sys
j=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1]); print(round(j["ms_per_step"],2), (j.get("verify") or {}).get("match"), [round(r["assemble_ms"],1) for r in j["iterations"]])
PY
