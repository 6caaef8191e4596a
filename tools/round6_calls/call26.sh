#!/bin/bash
# round 6, GPU call 26: the group kernel's grid again, now that its fetch looks ahead (workgroups per CU: 6 was the default; each workgroup walks a run of consecutive buckets)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call26; mkdir -p $O
export PYTHONUNBUFFERED=1
run() { env "$@" timeout 400 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify 2>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); s=d['roofline']['stage_ms_per_step']
print('%-44s %.1f | ext %.1f+%.1f part %.1f grp %.1f sort2 %.1f resc %.1f asm %.1f' % (sys.argv[1], d['ms_per_step'], s['extractShortKernel'], s['extractKernel'], s['hash_partition(all passes)'], s['groupKernel'], s['rep_sort(partition+aggSortKernel)'], s['rescoreKernel'], s['assemble_stage']))" "$*" | tee -a $O/sweep.txt; tail -2 $O/err.txt | grep -v amdgpu.ids; }
run X=0
for g in 2 3 4 8 12 24; do run PLASSHIP_TUNE_GROUP=$g; done
for g in 16 48 96; do run PLASSHIP_TUNE_AGGSORT=$g; done
