#!/bin/bash
# round 4, GPU call 22: the driver's command again (bench.py now runs the CPU-baseline / drop-in CLI leg before the wall-clock leg), c2 A/B of the residency cap
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call22; mkdir -p $O
export PYTHONUNBUFFERED=1
PLASS_BENCH_VERBOSE=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2> $O/bench_driver_cmd.err; tail -c 600 $O/bench_driver_cmd.log; echo
for v in "X=1" "PLASSHIP_TUNE_SHORT_PAD_KB=1"; do
  for r in 1 2; do env $v timeout 300 python bench.py --config c2 --no-wall --no-cpu-baseline > $O/bench_c2_${v}_$r.log 2>/dev/null; python - "$O/bench_c2_${v}_$r.log" <<'PY'
import json,sys
j=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1]); print(sys.argv[1], round(j["ms_per_step"],3), round(j["value"]/1e6,1))
PY
  done
done
