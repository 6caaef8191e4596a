#!/bin/bash
# round 5, GPU call 13: rescoring kernel at 3 / 4 wavefronts per SIMD (no spills), thresholds and grids around it (the 12 iterations once per setting; stage times from the bench line)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call13; mkdir -p $O
export PYTHONUNBUFFERED=1
run() { env "$@" timeout 200 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); s=d['roofline']['stage_ms_per_step']
print('%-34s %.1f | ext %.1f+%.1f part %.1f grp %.1f sort2 %.1f resc %.1f a16 %.1f a32+64 %.1f big %.1f asm %.1f' % (sys.argv[1], d['ms_per_step'], s['extractShortKernel'], s['extractKernel'], s['hash_partition(all passes)'], s['groupKernel'], s['rep_sort(partition+aggSortKernel)'], s['rescoreKernel'], s['assembleGroupKernel<16>'], s['assembleGroupKernel<32>+<64>'], s['assembleBigKernel'], s['assemble_stage']))" "$*" | tee -a $O/sweep.txt; }
run X=0
run PLASSHIP_TUNE_RESCORE_WPE=4
run PLASSHIP_TUNE_RESCORE_WPE=3
run PLASSHIP_TUNE_RESCORE_WPE=4 PLASSHIP_TUNE_RESCORE_SHORT=768
run PLASSHIP_TUNE_RESCORE_WPE=4 PLASSHIP_TUNE_RESCORE_SHORT=1024
run PLASSHIP_TUNE_RESCORE_WPE=4 PLASSHIP_TUNE_RESCORE_SHORT=2048
run PLASSHIP_TUNE_RESCORE_WPE=4 PLASSHIP_TUNE_RESCORE=24
run PLASSHIP_TUNE_RESCORE_WPE=4 PLASSHIP_TUNE_RESCORE=48
run PLASSHIP_TUNE_RESCORE_WPE=3 PLASSHIP_TUNE_RESCORE_SHORT=1024
run PLASSHIP_TUNE_RESCORE_WPE=4 PLASSHIP_TUNE_ASM_OWN=4
run PLASSHIP_TUNE_RESCORE_WPE=4 PLASSHIP_TUNE_ASM_OWN=8
