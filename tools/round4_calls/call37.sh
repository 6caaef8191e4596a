#!/bin/bash
# round 4, GPU call 37: tools/rescore_loop_bench (scoring loop of the thread-per-pair rescoring: product's index arithmetic against two table indices per permute)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call37; mkdir -p $O
timeout 100 tools/rescore_loop_bench 4194304 115 > $O/rescore_loop_bench.log 2>&1; cat $O/rescore_loop_bench.log
