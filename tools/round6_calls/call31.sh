#!/bin/bash
# round 6, GPU call 31: repRunsKernel / placeRunsKernel (wave per bucket, parked 96 %) taking four groups of 64 triples per trip
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call31; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deep.py tests/test_gpu_sharded.py -m gpu -x -q --timeout 1200 > $O/pytest.log 2>&1; tail -2 $O/pytest.log
run() { env "$@" timeout 400 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall 2>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); s=d['roofline']['stage_ms_per_step']
print('%-44s %.1f verify %s | ext %.1f+%.1f part %.1f grp %.1f sort2 %.1f red %.1f resc %.1f asm %.1f' % (sys.argv[1], d['ms_per_step'], d['verify'].get('match'), s['extractShortKernel'], s['extractKernel'], s['hash_partition(all passes)'], s['groupKernel'], s['rep_sort(partition+aggSortKernel)'], s['run_reduce(reduceRunsKernel+CSR)'], s['rescoreKernel'], s['assemble_stage']))" "$*" | tee -a $O/sweep.txt; tail -2 $O/err.txt | grep -v amdgpu.ids; }
run X=0
run X=0
