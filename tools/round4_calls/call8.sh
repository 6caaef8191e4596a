#!/bin/bash
# round 4, GPU call 8: the fused driver's wall clock with the arena reserved in the background while the DB files are read, and the
# data file written with pwrite() on all host threads (two runs of the same command: page-cache state differs)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call8; mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 900 python tools/chain_wall_probe.py ) > $O/chain_wall_probe.log 2>&1
( time timeout 600 python -m pytest tests/test_gpu_chain_cli.py tests/test_gpu_orfs.py -m gpu -q --timeout 500 -x ) > $O/pytest_cli.log 2>&1
tail -c 4000 $O/chain_wall_probe.log; tail -3 $O/pytest_cli.log
