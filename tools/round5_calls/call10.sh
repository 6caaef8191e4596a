#!/bin/bash
# round 5, GPU call 10: the extension kernels' scoring loops skip the two byte-range masks on interior words; configs[1] bench line again (in the
# evidence run its first timed iteration waited 4.4 s in a hipMalloc while the driver scrubbed the 270 GB the previous process had returned)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call10; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 300 python bench.py --config c2 --no-wall > $O/bench_c2.log 2>/dev/null; python -c "
import json; d=json.loads([x for x in open('$O/bench_c2.log') if x.startswith('{')][-1]); print('c2 ms/step %.2f M/s %.1f' % (d['ms_per_step'], d['value']/1e6), 'verify', d['verify']['match'], [round(i['ms'],1) for i in d['iterations']])"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deep.py tests/test_gpu_large.py -m gpu -q -x --timeout 800 > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
env PLASS_BENCH_VERBOSE=1 timeout 400 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall > $O/bench_default.log 2> $O/bench_default.err
python - "$O/bench_default.log" <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); r=d["roofline"]
print("ms/step %.1f" % d["ms_per_step"], "verify", d["verify"]["match"] if d.get("verify") else None, {k: round(v,1) for k,v in r["stage_ms_per_step"].items()}, "wall", r["module_wall_ms_per_step"])
PY
