#!/bin/bash
# round 5, GPU call 12: knob sweep after this round's changes (the 12 iterations once per setting; stage times from the bench line)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call12; mkdir -p $O
export PYTHONUNBUFFERED=1
run() { env "$@" timeout 200 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); s=d['roofline']['stage_ms_per_step']
print('%-34s %.1f | ext %.1f+%.1f part %.1f grp %.1f sort2 %.1f resc %.1f a16 %.1f a32+64 %.1f big %.1f asm %.1f' % (sys.argv[1], d['ms_per_step'], s['extractShortKernel'], s['extractKernel'], s['hash_partition(all passes)'], s['groupKernel'], s['rep_sort(partition+aggSortKernel)'], s['rescoreKernel'], s['assembleGroupKernel<16>'], s['assembleGroupKernel<32>+<64>'], s['assembleBigKernel'], s['assemble_stage']))" "$*" | tee -a $O/sweep.txt; }
run X=0
run PLASSHIP_TUNE_RESCORE_SHORT=256
run PLASSHIP_TUNE_RESCORE_SHORT=384
run PLASSHIP_TUNE_RESCORE_SHORT=1024
run PLASSHIP_TUNE_RESCORE_WPE=4
run PLASSHIP_TUNE_RESCORE_WPE=6
run PLASSHIP_TUNE_RESCORE=16
run PLASSHIP_TUNE_RESCORE=64
run PLASSHIP_TUNE_ASM16=4
run PLASSHIP_TUNE_ASM16=6
run PLASSHIP_TUNE_ASM64=3
run PLASSHIP_TUNE_ASM64=5
run PLASSHIP_TUNE_ASMBIG=2
run PLASSHIP_TUNE_ASMBIG=8
run PLASSHIP_TUNE_GROUP=4
run PLASSHIP_TUNE_GROUP=8
run PLASSHIP_TUNE_AGGSORT=8
run PLASSHIP_TUNE_AGGSORT=32
run PLASSHIP_TUNE_WRITEOUT=8
run PLASSHIP_TUNE_WRITEOUT=32
run PLASSHIP_TUNE_CACHED=16
run PLASSHIP_TUNE_CACHED=64
run X=1
