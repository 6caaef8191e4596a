#!/bin/bash
# round 5, GPU call 8: list entries of the partition's LIST input one tile ahead of the records (level 2 of both partitions)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call8; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 500 > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
env PLASS_BENCH_VERBOSE=1 timeout 400 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall > $O/bench_default.log 2> $O/bench_default.err
python - "$O/bench_default.log" <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); r=d["roofline"]
print("ms/step %.1f" % d["ms_per_step"], "verify", d["verify"]["match"] if d.get("verify") else None, {k: round(v,1) for k,v in r["stage_ms_per_step"].items()}, "wall", r["module_wall_ms_per_step"])
PY
