#!/bin/bash
# round 4, GPU call 30: the N > 1 bench line with per-module exchange figures, exercised in a 1-rank native-RCCL group (12.5 M reads) and in the single-GPU path
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call30; mkdir -p $O
MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 PLASS_BENCH_FORCE_DIST=1 timeout 400 python bench.py --pairs 6250000 --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify > $O/bench_12M_sharded_1rank.log 2> $O/bench_12M_sharded_1rank.err
python - $O/bench_12M_sharded_1rank.log <<'PY'
import json,sys
j=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1]); print(round(j["ms_per_step"],2), json.dumps(j.get("exchange"))[:900])
PY
timeout 300 python bench.py --config c2 --no-wall --no-cpu-baseline > $O/bench_c2.log 2>/dev/null; tail -c 200 $O/bench_c2.log; echo
timeout 300 python -m pytest tests/test_gpu_sharded.py -m gpu -q -x --timeout 280 -k "bench or torch or one_rank" > $O/pytest_sharded.log 2>&1; tail -2 $O/pytest_sharded.log
