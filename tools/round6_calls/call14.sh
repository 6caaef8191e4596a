#!/bin/bash
# round 6, GPU call 14: the whole GPU suite with the row tier, the position cache and the early exits of the wave tiers; A/B of the early exit; driver's command
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call14; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q --timeout 1500 --durations=8 > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
run() { env "$@" timeout 400 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall 2>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); s=d['roofline']['stage_ms_per_step']
print('%-44s %.1f verify %s | ext %.1f+%.1f part %.1f grp %.1f sort2 %.1f resc %.1f asm %.1f' % (sys.argv[1], d['ms_per_step'], d['verify'].get('match'), s['extractShortKernel'], s['extractKernel'], s['hash_partition(all passes)'], s['groupKernel'], s['rep_sort(partition+aggSortKernel)'], s['rescoreKernel'], s['assemble_stage']))
print('      extraction per iteration: ' + ' '.join('%.1f' % r.get('extract_ms', -1) for r in d['iterations']))" "$*" | tee -a $O/sweep.txt; tail -2 $O/err.txt | grep -v amdgpu.ids; }
run X=0
run PLASSHIP_EXTRACT_BLOCKS_PER_CU=64
run PLASSHIP_EXTRACT_BLOCKS_PER_CU=128
run PLASSHIP_EXTRACT_BLOCKS_PER_CU=384
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err; tail -c 300 $O/bench.log; echo
