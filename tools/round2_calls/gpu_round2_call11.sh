#!/bin/bash
# round 2, GPU call 11: full GPU suite, host waits per iteration (1 M-read set), kernel trace of the driver's bench command
mkdir -p gpurun_out/c11
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/c11/pytest.log 2>&1
tail -15 gpurun_out/c11/pytest.log
timeout 300 python bench.py --config c2 --no-cpu-baseline > gpurun_out/c11/bench_c2.log 2> gpurun_out/c11/bench_c2.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c11/bench_c2.log").read().strip().splitlines()[-1])
print("c2:", d["value"], d["ms_per_step"], [r["host_waits"] for r in d["iterations"]])
PY
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c11/prof -o driver -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/c11/bench_driver.log 2> $GRAFT_REPO_ROOT/gpurun_out/c11/bench_driver.err
cd $GRAFT_REPO_ROOT
tail -c 1500 gpurun_out/c11/bench_driver.log
find gpurun_out/c11/prof -name "*kernel_trace*" -size +20M -delete
ls -la gpurun_out/c11/prof/* | head
