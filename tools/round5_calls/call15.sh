#!/bin/bash
# round 5, GPU call 15 (last): the native-communicator tests in their own module (tests/test_gpu_native_comm.py: the child process of
# test_native_comm_multi_rank_nucl_and_guided twice found 96 MB of free HBM behind test_gpu_sharded.py's long-lived contexts) + tests/test_gpu_sharded.py
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call15; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_gpu_native_comm.py tests/test_gpu_sharded.py -m gpu -q --timeout 250 --durations=5 > $O/pytest_gpu.log 2>&1; tail -9 $O/pytest_gpu.log
