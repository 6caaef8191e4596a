#!/bin/bash
# round 5, GPU call 16: the whole GPU suite on the final sources and test layout (what is left of the GPU budget)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call16; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 318 python -m pytest tests -m gpu -q --timeout 300 --durations=8 > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
