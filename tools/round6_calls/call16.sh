#!/bin/bash
# round 6, GPU call 16: the driver's command with the chain child run twice (behind the parent / after an idle wait)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call16; mkdir -p $O
export PYTHONUNBUFFERED=1
PLASS_BENCH_VERBOSE=1 timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2> $O/bench_driver_cmd.err; tail -c 400 $O/bench_driver_cmd.log; echo
python - <<'P'
import json
d=json.loads([x for x in open('gpurun_out/r06_call16/bench_driver_cmd.log') if x.startswith('{')][-1])
print(d['ms_per_step'], d['value'], d['verify'].get('match')); print(json.dumps(d['wall_to_contigs'], indent=1)); print(d['roofline'].get('traffic'))
P
