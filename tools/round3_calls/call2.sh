#!/bin/bash
# round 3, GPU call 2: the record cache (static store + KILL records) — parity suite, cached vs uncached chain, the large oracle-checked
# chains, the three-way cross-check at 50 M reads (all 12 iterations), the fused C++ drivers, then the driver's bench command.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_call2; mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_orfs.py tests/test_gpu_chain_cli.py -m gpu -q --timeout 600 --durations=8 ) > $O/parity.log 2>&1
echo "parity rc=$?" | tee -a $O/summary.txt
( time timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -q --timeout 400 --durations=5 ) > $O/sharded.log 2>&1
echo "sharded rc=$?" | tee -a $O/summary.txt
( time timeout 1800 python -m pytest tests/test_gpu_large.py tests/test_gpu_large_nucl.py -m gpu -q --timeout 1700 --durations=8 ) > $O/large.log 2>&1
echo "large rc=$?" | tee -a $O/summary.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
tail -c 2500 $O/parity.log; tail -c 800 $O/sharded.log; tail -c 2000 $O/large.log; tail -c 1500 $O/bench.log; tail -c 600 $O/bench.err
