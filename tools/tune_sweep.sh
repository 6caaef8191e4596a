#!/bin/bash
# launch-geometry sweep through the PLASSHIP_TUNE_<name> knobs (workgroups per CU of a kernel's grid cap); one bench line per setting
run() { env "$@" timeout 120 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); s=d['roofline']['stage_ms_per_step']
print('%-28s %.3f | short %.2f grp %.2f sort2 %.2f resc %.2f a16 %.2f a32+64 %.2f' % (sys.argv[1], d['ms_per_step'], s['extractShortKernel'], s['groupKernel'], s['rep_sort(partition+aggSortKernel)'], s['rescoreKernel'], s['assembleGroupKernel<16>'], s['assembleGroupKernel<32>+<64>']))" "$*"; }
if [ $# -gt 0 ]; then for cfg in "$@"; do run $cfg; done; exit 0; fi
run X=0
run PLASSHIP_TUNE_ASM16=4
run PLASSHIP_TUNE_ASM16=6
run PLASSHIP_TUNE_ASM32=4 PLASSHIP_TUNE_ASM64=4
run PLASSHIP_TUNE_ASM32=6 PLASSHIP_TUNE_ASM64=6
run PLASSHIP_TUNE_RESCORE=10
run PLASSHIP_TUNE_RESCORE=12
run PLASSHIP_TUNE_RESCORE=14
run PLASSHIP_TUNE_SHORT=18
run PLASSHIP_TUNE_RESCORE=12 PLASSHIP_TUNE_SHORT=16 PLASSHIP_TUNE_AGGSORT=32
run X=1
