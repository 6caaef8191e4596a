#!/bin/bash
# round 6, GPU call 37: the sharded tests (2-8 ranks on one GPU, both extraction modes, native communicator through the stub) with the row kernels FORCED — in the
# exchange mode a rank extracts a sub-range of the ids into a slot array of its own (slotBias != 0), which the rows would only meet at 8 GPUs and 50 M reads
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call37; mkdir -p $O
export PYTHONUNBUFFERED=1
PLASSHIP_TUNE_ROWTIER=3 timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_native_comm.py -m gpu -x -q --timeout 1200 > $O/pytest_sharded_rows.log 2>&1; tail -3 $O/pytest_sharded_rows.log
