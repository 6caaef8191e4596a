#!/bin/bash
# round 4, GPU call 36: sharded / chain / CLI tests on the final sources (the parity file ran in call 34)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call36; mkdir -p $O
timeout 110 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_chain_cli.py tests/test_gpu_orfs.py -m gpu -q --timeout 100 -x > $O/pytest_rest.log 2>&1; tail -2 $O/pytest_rest.log
