#!/bin/bash
# round 6, GPU call 8: grids of the other persistent kernels after call 7's finding (a grid of exactly one resident round leaves a tail; finer shares help)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call8; mkdir -p $O
export PYTHONUNBUFFERED=1
run() { env "$@" timeout 300 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify 2>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); s=d['roofline']['stage_ms_per_step']
print('%-44s %.1f | ext %.1f+%.1f part %.1f grp %.1f sort2 %.1f resc %.1f a16 %.1f a32+64 %.1f big %.1f asm %.1f' % (sys.argv[1], d['ms_per_step'], s['extractShortKernel'], s['extractKernel'], s['hash_partition(all passes)'], s['groupKernel'], s['rep_sort(partition+aggSortKernel)'], s['rescoreKernel'], s['assembleGroupKernel<16>'], s['assembleGroupKernel<32>+<64>'], s['assembleBigKernel'], s['assemble_stage']))" "$*" | tee -a $O/sweep.txt; tail -2 $O/err.txt | grep -v amdgpu.ids; }
run X=0
for v in 2 4 8 16; do run PLASSHIP_TUNE_ASM_GRIDX=$v; done
for v in 8 16 32; do run PLASSHIP_TUNE_ASMBIG=$v; done
for v in 32 64 128; do run PLASSHIP_TUNE_TIER2=$v; done
for v in 10 20; do run PLASSHIP_TUNE_TIER3=$v; done
for v in 12 24 48; do run PLASSHIP_TUNE_GROUP=$v; done
for v in 32 64 128; do run PLASSHIP_TUNE_AGGSORT=$v; done
for v in 128 256; do run PLASSHIP_TUNE_CACHED=$v; done
for v in 128 256; do run PLASSHIP_TUNE_RESCORE=$v; done
