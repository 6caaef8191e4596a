#!/bin/bash
# round 5, GPU call 5: tagScatterKernel with 32 768-tag chunks, appendOutKernel as thread-per-entry index + wavefront copies, CAS heap reservation;
# parity / chain / deep / large tests, the 12-iteration chain once (default; 16 GB of heap slack), and the sharded orchestration in a 1-rank
# native-RCCL group at 12.5 M reads against the single-GPU path (owner-filtered = default up to 4 ranks; exchange for comparison)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call5; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deep.py tests/test_gpu_large.py tests/test_gpu_chain_cli.py tests/test_gpu_large_nucl.py -m gpu -q -x --timeout 1000 > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
summ() { python - "$1" "$2" <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if not l: print(sys.argv[2], "NO LINE"); sys.exit()
d=json.loads(l[-1]); r=d["roofline"]
print(sys.argv[2], "ms/step %.1f" % d["ms_per_step"], "verify", d["verify"]["match"] if d.get("verify") else None, {k: round(v,1) for k,v in r["stage_ms_per_step"].items()}, "wall", r["module_wall_ms_per_step"])
print("   per iteration ms", [round(it["ms"],1) for it in d["iterations"]])
PY
}
env PLASS_BENCH_VERBOSE=1 timeout 400 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall > $O/bench_default.log 2> $O/bench_default.err; summ $O/bench_default.log default
env PLASSHIP_TUNE_DBHEAP_GB=16 PLASS_BENCH_VERBOSE=1 timeout 400 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify > $O/bench_heap16.log 2> $O/bench_heap16.err; summ $O/bench_heap16.log DBHEAP_GB=16
timeout 300 python bench.py --pairs 6250000 --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify > $O/bench_12M_single.log 2>/dev/null; summ $O/bench_12M_single.log 12M_single
MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 PLASS_BENCH_FORCE_DIST=1 timeout 300 python bench.py --pairs 6250000 --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify > $O/bench_12M_sharded_1rank_filtered.log 2> $O/bench_12M_sharded_1rank_filtered.err; summ $O/bench_12M_sharded_1rank_filtered.log 12M_1rank_owner_filtered
MASTER_ADDR=127.0.0.1 MASTER_PORT=29562 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 PLASS_BENCH_FORCE_DIST=1 PLASSHIP_TUNE_SHARD_EXTRACT=1 timeout 300 python bench.py --pairs 6250000 --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify > $O/bench_12M_sharded_1rank_exchange.log 2> $O/bench_12M_sharded_1rank_exchange.err; summ $O/bench_12M_sharded_1rank_exchange.log 12M_1rank_exchange
