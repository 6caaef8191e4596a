#!/bin/bash
# round 5, GPU call 4: the tests call 2 failed (guided chain at depth: DB layout behind appended entries; CLI exit code 95; injected failure
# beyond the collectives of an owner-filtered iteration), the guided / nucleotide large tests, then the 12-iteration chain once
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call4; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_deep.py tests/test_gpu_large_nucl.py tests/test_gpu_chain_cli.py "tests/test_gpu_parity.py::test_cli_unsupported_fails_loudly" "tests/test_gpu_parity.py::test_golden_guided_modules" "tests/test_gpu_sharded.py::test_sharded_rank_local_failure_ends_the_call_on_every_rank" -m gpu -q --timeout 800 > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
env PLASS_BENCH_VERBOSE=1 timeout 400 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall > $O/bench_default.log 2> $O/bench_default.err
python - "$O/bench_default.log" <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); r=d["roofline"]
print("ms/step %.1f" % d["ms_per_step"], "verify", d["verify"]["match"] if d.get("verify") else None, {k: round(v,1) for k,v in r["stage_ms_per_step"].items()}, "wall", r["module_wall_ms_per_step"])
print("furthest_below", r.get("furthest_below"))
PY
timeout 300 python bench.py --config c2 --no-wall --no-cpu-baseline > $O/bench_c2.log 2> $O/bench_c2.err; python -c "
import json,sys
d=json.loads([x for x in open('$O/bench_c2.log') if x.startswith('{')][-1]); print('c2 ms/step %.2f' % d['ms_per_step'], 'verify', d.get('verify'))"
