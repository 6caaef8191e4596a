#!/bin/bash
# round 5, GPU call 6: lazy self hits (identity pairs left as stubs by rescorediagonal, scored when a consumer reads them); the whole GPU suite,
# the 12-iteration chain once with the stubs on / off, the driver's command
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call6; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1700 python -m pytest tests -m gpu -q --timeout 1500 > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
summ() { python - "$1" "$2" <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if not l: print(sys.argv[2], "NO LINE"); sys.exit()
d=json.loads(l[-1]); r=d["roofline"]
print(sys.argv[2], "ms/step %.1f" % d["ms_per_step"], "M overlaps/s %.1f" % (d["value"]/1e6), "verify", d["verify"]["match"] if d.get("verify") else None, {k: round(v,1) for k,v in r["stage_ms_per_step"].items()}, "wall", r["module_wall_ms_per_step"])
print("   per iteration ms", [round(it["ms"],1) for it in d["iterations"][:12]], "rescore", [round(it["rescore_ms"],1) for it in d["iterations"][:12]])
PY
}
env PLASS_BENCH_VERBOSE=1 timeout 400 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall > $O/bench_default.log 2> $O/bench_default.err; summ $O/bench_default.log default
env PLASSHIP_TUNE_LAZY_SELF=2 timeout 400 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify > $O/bench_lazy_off.log 2> $O/bench_lazy_off.err; summ $O/bench_lazy_off.log LAZY_SELF=2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-wall > $O/bench_driver.log 2> $O/bench_driver.err; summ $O/bench_driver.log driver_cmd
