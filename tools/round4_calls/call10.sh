#!/bin/bash
# round 4, GPU call 10: configs[4] (c5) with the nucleotide pre-screen, the lazy query copy and the thread / wavefront routing by arena slice.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call10; mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_nucl.py tests/test_gpu_sharded.py -m gpu -q --timeout 800 -x -k "nucl or guided or strand or hairpin or penguin" ) > $O/pytest_nucl.log 2>&1
tail -3 $O/pytest_nucl.log
for tb in 16384 4096 65536 1000000000; do
  ( time PLASSHIP_TUNE_NUCL_THREAD_BYTES=$tb PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --config c5 --steps 10 --warmup 0 --no-cpu-baseline ) > $O/bench_c5_tb$tb.log 2> $O/bench_c5_tb$tb.err
  python - "$O/bench_c5_tb$tb.log" <<'PY'
import json,sys
try:
    j=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1]); print(sys.argv[1], round(j["value"]/1e6,1), round(j["ms_per_step"],2), (j.get("verify") or {}).get("match"), [round(r["assemble_ms"],1) for r in j["iterations"]])
except Exception as e: print("ERR", e)
PY
done
