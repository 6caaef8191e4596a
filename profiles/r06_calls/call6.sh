#!/bin/bash
# round 6, GPU call 6: knob re-sweep on this round's sources (the 12 iterations once per setting) + the new GPU tests of this round's fixtures
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call6; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_deep.py -m gpu -x -q --timeout 800 -k "circular" > $O/pytest_circular.log 2>&1; tail -2 $O/pytest_circular.log
run() { env "$@" timeout 300 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify 2>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); s=d['roofline']['stage_ms_per_step']
print('%-44s %.1f | ext %.1f+%.1f part %.1f grp %.1f sort2 %.1f resc %.1f a16 %.1f a32+64 %.1f big %.1f asm %.1f' % (sys.argv[1], d['ms_per_step'], s['extractShortKernel'], s['extractKernel'], s['hash_partition(all passes)'], s['groupKernel'], s['rep_sort(partition+aggSortKernel)'], s['rescoreKernel'], s['assembleGroupKernel<16>'], s['assembleGroupKernel<32>+<64>'], s['assembleBigKernel'], s['assemble_stage']))" "$*" | tee -a $O/sweep.txt; tail -2 $O/err.txt | grep -v amdgpu.ids; }
run X=0
run PLASSHIP_TUNE_DBHEAP_GB=24
run PLASSHIP_TUNE_DBHEAP_GB=32
run PLASSHIP_TUNE_TIER0_WPE=6
run PLASSHIP_EXTRACT_BLOCKS_PER_CU=24
run PLASSHIP_EXTRACT_BLOCKS_PER_CU=48
run PLASSHIP_TUNE_CACHED=16
run PLASSHIP_TUNE_CACHED=64
run PLASSHIP_TUNE_SHORT=72
run PLASSHIP_TUNE_SHORT=288
run PLASSHIP_TUNE_WRITEOUT=8
run PLASSHIP_TUNE_WRITEOUT=32
run PLASSHIP_TUNE_RESCORE=16
run PLASSHIP_TUNE_RESCORE=64
