#!/bin/bash
# round 2, GPU call 14: parallel deferred re-scoring in assembleBigKernel, 8-lane writeOutKernel: parity, then a kernel trace of the 50 M-read bench
mkdir -p gpurun_out/c14
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -x -q > gpurun_out/c14/pytest.log 2>&1
tail -3 gpurun_out/c14/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c14/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/c14/bench.log 2> $GRAFT_REPO_ROOT/gpurun_out/c14/bench.err
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/c14/prof/trace_results.db > gpurun_out/c14/kernel_stats.txt 2>&1
head -32 gpurun_out/c14/kernel_stats.txt | cut -c1-150
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c14/bench.log").read().strip().splitlines()[-1])
print("value", round(d["value"] / 1e6, 1), "ms/step", round(d["ms_per_step"], 1), "assemble_ms", [r["assemble_ms"] for r in d["iterations"]])
PY
PLASSHIP_TUNE_WRITEOUT_G=16 timeout 300 python bench.py --steps 4 --warmup 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('writeout G16: assemble_ms', [r['assemble_ms'] for r in d['iterations']])"
find gpurun_out/c14 -name "*.db" -size +30M -delete
