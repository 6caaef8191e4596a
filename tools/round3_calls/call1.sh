#!/bin/bash
# round 3, GPU call 1: the sharded line-store port, the multi-rank native communicator over the RCCL stub, the three-way
# cross-check at 50 M reads, the 12.5 M-read protein chain and the 5 M-read nucleotide / guided chains against the CPU oracle's
# digests, then the rest of the GPU suite and the driver's bench command.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_call1; mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 1500 python -m pytest tests/test_gpu_sharded.py -m gpu -q --timeout 400 --durations=8 ) > $O/sharded.log 2>&1
echo "sharded rc=$?" | tee -a $O/summary.txt
( time timeout 1500 python -m pytest tests/test_gpu_large.py tests/test_gpu_large_nucl.py -m gpu -q --timeout 1500 --durations=8 ) > $O/large.log 2>&1
echo "large rc=$?" | tee -a $O/summary.txt
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_orfs.py -m gpu -q --timeout 600 --durations=8 ) > $O/parity.log 2>&1
echo "parity rc=$?" | tee -a $O/summary.txt
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
tail -c 1500 $O/sharded.log; tail -c 1500 $O/large.log; tail -c 600 $O/parity.log; tail -c 1200 $O/bench.log
