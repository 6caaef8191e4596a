#!/bin/bash
# round 6, GPU call 17: the fused driver's wall clock with the process leaving without taking its context apart (cli_main.cpp), against the orderly teardown
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call17; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python tools/chain_wall_probe.py > $O/wall_probe.txt 2>$O/err.txt; grep -E "=== |chain:|plasship io" $O/wall_probe.txt | cut -c1-260
