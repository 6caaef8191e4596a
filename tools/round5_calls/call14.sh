#!/bin/bash
# round 5, GPU call 14: rescoring at 4 wavefronts per SIMD and a 768-residue thread-per-pair limit as defaults; the 32-lane extension tier at 4 wavefronts (the 12 iterations once per setting; stage times from the bench line)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call14; mkdir -p $O
export PYTHONUNBUFFERED=1
run() { env "$@" timeout 200 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); s=d['roofline']['stage_ms_per_step']
print('%-34s %.1f | ext %.1f+%.1f part %.1f grp %.1f sort2 %.1f resc %.1f a16 %.1f a32+64 %.1f big %.1f asm %.1f' % (sys.argv[1], d['ms_per_step'], s['extractShortKernel'], s['extractKernel'], s['hash_partition(all passes)'], s['groupKernel'], s['rep_sort(partition+aggSortKernel)'], s['rescoreKernel'], s['assembleGroupKernel<16>'], s['assembleGroupKernel<32>+<64>'], s['assembleBigKernel'], s['assemble_stage']))" "$*" | tee -a $O/sweep.txt; }
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deep.py -m gpu -q -x --timeout 500 > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
run X=0
run PLASSHIP_TUNE_ASM32=4
run PLASSHIP_TUNE_RESCORE_WPE=5
