#!/bin/bash
# round 4, GPU call 16: nucleotide sequences sorted into the extraction tiers by window count (4-scores tier for reads)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call16; mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_nucl.py tests/test_gpu_chain_cli.py tests/test_gpu_sharded.py -m gpu -q --timeout 800 -x -k "cycle or nucl or penguin or chain or guided or strand or hairpin or flag or cli" ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for v in "X=1" "PLASSHIP_TUNE_CLASSIFY=2"; do
  ( time env $v timeout 600 python bench.py --config c5 --steps 10 --warmup 0 --no-cpu-baseline ) > $O/bench_c5_$v.log 2> $O/bench_c5_$v.err
  python - "$O/bench_c5_$v.log" <<'PY'
import json,sys
try:
    j=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1]); print(sys.argv[1], round(j["value"]/1e6,1), round(j["ms_per_step"],2), (j.get("verify") or {}).get("match"), [round(r["extract_ms"],1) for r in j["iterations"]])
except Exception as e: print("ERR", e)
PY
done
