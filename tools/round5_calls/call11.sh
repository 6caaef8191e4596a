#!/bin/bash
# round 5, GPU call 11: wide register queues (32 / 64 lanes per query): deferred hits re-scored one after the other by the whole group when a round
# defers fewer than N of them, each by its own lane otherwise (N = 1: always own lane, rounds 2-5)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call11; mkdir -p $O
export PYTHONUNBUFFERED=1
for v in 1 3 6 12 100; do
  env PLASSHIP_TUNE_ASM_OWN=$v timeout 400 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify > $O/bench_own$v.log 2> $O/bench_own$v.err
  python - "$O/bench_own$v.log" $v <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); r=d["roofline"]; st=r["stage_ms_per_step"]
print("ASM_OWN", sys.argv[2], "ms/step %.1f" % d["ms_per_step"], "tiers", round(st["assembleGroupKernel<16>"],1), round(st["assembleGroupKernel<32>+<64>"],1), round(st["assembleBigKernel"],1), "assemble stage", round(st["assemble_stage"],1))
PY
done
