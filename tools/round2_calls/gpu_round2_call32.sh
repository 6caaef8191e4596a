#!/bin/bash
# round 2, GPU call 32: SWAR identity count in the pair re-scoring, DB writer with two sequences in flight per group: parity + kernel trace
mkdir -p gpurun_out/c32
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py tests/test_gpu_orfs.py -m gpu -x -q > gpurun_out/c32/pytest.log 2>&1
tail -3 gpurun_out/c32/pytest.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c32/prof -o t -- python $R/bench.py --steps 12 --warmup 0 --no-cpu-baseline > $R/gpurun_out/c32/bench.log 2> $R/gpurun_out/c32/bench.err
cd $R
python tools/rocpd_summary.py gpurun_out/c32/prof/t_results.db 2>/dev/null | head -28 | cut -c1-150
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c32/bench.log").read().strip().splitlines()[-1])
print("value", round(d["value"] / 1e6, 1), "ms/step", round(d["ms_per_step"], 1))
PY
