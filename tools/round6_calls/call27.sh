#!/bin/bash
# round 6, GPU call 27: more grid points around call 26's (group 24 per CU as the base)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call27; mkdir -p $O
export PYTHONUNBUFFERED=1
export PLASSHIP_TUNE_GROUP=24
run() { env "$@" timeout 400 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify 2>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); s=d['roofline']['stage_ms_per_step']
print('%-44s %.1f | ext %.1f+%.1f part %.1f grp %.1f sort2 %.1f resc %.1f a16 %.1f a32+64 %.1f big %.1f asm %.1f' % (sys.argv[1], d['ms_per_step'], s['extractShortKernel'], s['extractKernel'], s['hash_partition(all passes)'], s['groupKernel'], s['rep_sort(partition+aggSortKernel)'], s['rescoreKernel'], s['assembleGroupKernel<16>'], s['assembleGroupKernel<32>+<64>'], s['assembleBigKernel'], s['assemble_stage']))" "$*" | tee -a $O/sweep.txt; tail -2 $O/err.txt | grep -v amdgpu.ids; }
run X=0
run PLASSHIP_TUNE_GROUP=48
run PLASSHIP_TUNE_GROUP=96
run PLASSHIP_TUNE_SHORT=288
run PLASSHIP_TUNE_SHORT=576
run PLASSHIP_TUNE_ASM16=4
run PLASSHIP_TUNE_ASM16=6
run PLASSHIP_TUNE_ASM64=5
run PLASSHIP_TUNE_ASMBIG=8
run PLASSHIP_TUNE_RESCORE=64
