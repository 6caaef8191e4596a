#!/bin/bash
# round 4, GPU call 28: configs[4], which pairs the thread-per-pair rescoring keeps (PLASSHIP_TUNE_RESCORE_SHORT; default 512)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call28; mkdir -p $O
for v in 1024 2048 4096; do
  PLASSHIP_TUNE_RESCORE_SHORT=$v timeout 600 python bench.py --config c5 --steps 10 --warmup 0 --no-cpu-baseline --no-verify > $O/bench_c5_short$v.log 2> $O/bench_c5_short$v.err
  python - "$O/bench_c5_short$v.log" <<'PY'
import json,sys
j=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1]); print(sys.argv[1], round(j["ms_per_step"],2), [round(r["rescore_ms"],1) for r in j["iterations"]])
PY
done
