#!/bin/bash
# round 4, GPU call 13: cyclecheck with 8-byte LDS table entries (two phases), lazy histogram, multi-pass workgroup tier.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call13; mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_nucl.py tests/test_gpu_chain_cli.py -m gpu -q --timeout 800 -x -k "cycle or nucl or penguin or chain" ) > $O/pytest_cyc.log 2>&1
tail -15 $O/pytest_cyc.log
( time timeout 600 python bench.py --config c5 --steps 10 --warmup 0 --no-cpu-baseline ) > $O/bench_c5.log 2> $O/bench_c5.err
python - "$O/bench_c5.log" <<'PY'
import json,sys
try:
    j=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1]); print(sys.argv[1], round(j["value"]/1e6,1), round(j["ms_per_step"],2), (j.get("verify") or {}).get("match"), [round(r["aln2nucl_or_cyclecheck_ms"],1) for r in j["iterations"]])
except Exception as e: print("ERR", e)
PY
