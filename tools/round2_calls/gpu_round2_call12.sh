#!/bin/bash
# round 2, GPU call 12: wide group kernel (512 threads, 4096 slots) on/off at 50 M reads, parity, PMC traffic passes
mkdir -p gpurun_out/c12
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -x -q > gpurun_out/c12/pytest.log 2>&1
tail -3 gpurun_out/c12/pytest.log
for v in "1 4" "1 2" "0 4"; do
  set -- $v
  PLASSHIP_GROUP_WIDE=$1 PLASSHIP_TUNE_GROUP_WPE=$2 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/c12/bench_wide$1_wpe$2.log 2> gpurun_out/c12/bench_wide$1_wpe$2.err
  python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/c12/bench_wide%s_wpe%s.log" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
print("wide", sys.argv[1], "wpe", sys.argv[2], "value", round(d["value"] / 1e6, 1), "ms/step", round(d["ms_per_step"], 1), "group_ms", [r["group_ms"] for r in d["iterations"]][:12], "waits", d["iterations"][3]["host_waits"])
PY
done
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/c12/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/c12/pmc_$c.log 2> $GRAFT_REPO_ROOT/gpurun_out/c12/pmc_$c.err
done
cd $GRAFT_REPO_ROOT
python tools/rocpd_traffic.py gpurun_out/c12/pmc_FETCH_SIZE/pmc_results.db gpurun_out/c12/pmc_WRITE_SIZE/pmc_results.db 40 gpurun_out/c12/r02_pmc_traffic.json > gpurun_out/c12/r02_pmc_hbm_traffic.txt 2>&1
head -30 gpurun_out/c12/r02_pmc_hbm_traffic.txt
ls -la gpurun_out/c12/pmc_*/ | head
find gpurun_out/c12 -name "*.db" -size +30M -delete
