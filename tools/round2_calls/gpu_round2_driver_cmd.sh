#!/bin/bash
# the driver's bench command, plain (after profiles/r02_pmc_traffic.json was refreshed: roofline.traffic is read from it)
mkdir -p gpurun_out/final
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_driver_cmd.log 2> gpurun_out/final/bench_driver_cmd.err
tail -c 300 gpurun_out/final/bench_driver_cmd.log
