#!/bin/bash
# round 2, GPU call 10: full GPU suite (large parity test, host boundary), host boundary throughput
mkdir -p gpurun_out/c10
timeout 1200 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/c10/pytest.log 2>&1
tail -15 gpurun_out/c10/pytest.log
timeout 300 python tools/io_probe.py 2500000 > gpurun_out/c10/io_probe.log 2>&1
PLASSHIP_HOST_THREADS=1 timeout 300 python tools/io_probe.py 2500000 > gpurun_out/c10/io_probe_1thread.log 2>&1
cat gpurun_out/c10/io_probe.log gpurun_out/c10/io_probe_1thread.log
