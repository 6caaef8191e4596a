#!/bin/bash
# round 4, GPU call 25: the driver's command on the evidence run's sources with the PMC files of that run in place (roofline.traffic quoted) and
# bench.py's CPU-baseline leg before the wall-clock leg
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call25; mkdir -p $O
PLASS_BENCH_VERBOSE=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2> $O/bench_driver_cmd.err; tail -c 300 $O/bench_driver_cmd.log; echo
