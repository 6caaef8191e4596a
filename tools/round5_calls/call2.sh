#!/bin/bash
# round 5, GPU call 2: the whole GPU suite (sparse alignment lists, owner-filtered sharded extraction in both modes, side streams, deep-chain
# oracle tests), then the 12-iteration chain once with side streams on / off
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call2; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1700 python -m pytest tests -m gpu -q --timeout 1500 --durations=15 > $O/pytest_gpu.log 2>&1; tail -32 $O/pytest_gpu.log
for v in default "PLASSHIP_TUNE_SIDE_STREAMS=2"; do
  n=$(echo $v | tr -c 'A-Za-z0-9\n' '_')
  if [ "$v" = default ]; then env PLASS_BENCH_VERBOSE=1 timeout 400 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall > $O/bench_$n.log 2> $O/bench_$n.err
  else env $v PLASS_BENCH_VERBOSE=1 timeout 400 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall > $O/bench_$n.log 2> $O/bench_$n.err; fi
  python - "$O/bench_$n.log" "$v" <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if not l: print(sys.argv[2], "NO LINE"); sys.exit()
d=json.loads(l[-1]); r=d["roofline"]
print(sys.argv[2], "ms/step %.1f" % d["ms_per_step"], "verify", d["verify"]["match"] if d.get("verify") else None, {k: round(v,1) for k,v in r["stage_ms_per_step"].items()}, "wall", r["module_wall_ms_per_step"])
print("   extract per iteration", [it["extract_ms"] for it in d["iterations"]], "assemble", [it["assemble_ms"] for it in d["iterations"]])
PY
done
