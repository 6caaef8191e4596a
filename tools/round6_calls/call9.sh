#!/bin/bash
# round 6, GPU call 9: the whole GPU suite on the committed sources (incl. tests/test_gpu_c5_full.py in its own module) + the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call9; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q --timeout 1500 --durations=8 > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err; tail -c 600 $O/bench.log; echo
