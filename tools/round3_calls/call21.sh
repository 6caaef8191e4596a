#!/bin/bash
# GPU call 21: exchange 1 with the rank's own lines gathered straight into the receive buffer: parity, then 12.5 M reads single vs 1-rank sharded
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_call21; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_large.py tests/test_gpu_parity.py -m gpu -q -x --timeout 1200 --durations=5 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" > $O/summary.txt
MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 PLASS_BENCH_FORCE_DIST=1 timeout 400 python bench.py --pairs 6250000 --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify > $O/bench_12M_sharded_1rank.log 2> $O/bench_12M_sharded_1rank.err
cat $O/summary.txt; tail -4 $O/pytest.log
