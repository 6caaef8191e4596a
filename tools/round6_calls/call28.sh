#!/bin/bash
# round 6, GPU call 28: the parity, deep-chain and large tests with re-used device blocks poisoned (PLASSHIP_POOL_POISON): nothing this round added may read
# memory it has not written (row kernels, position cache, look-aheads)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call28; mkdir -p $O
export PYTHONUNBUFFERED=1
PLASSHIP_POOL_POISON=165 timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deep.py tests/test_gpu_large.py tests/test_gpu_large_nucl.py -m gpu -x -q --timeout 1500 > $O/pytest_poison.log 2>&1; tail -3 $O/pytest_poison.log
PLASSHIP_POOL_POISON=90 PLASSHIP_TUNE_ROWTIER=3 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 1000 > $O/pytest_poison_rows.log 2>&1; tail -2 $O/pytest_poison_rows.log
