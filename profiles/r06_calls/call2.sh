#!/bin/bash
# round 6, GPU call 2: the full GPU suite with this round's new tests (reference-only 12-iteration chain, configs[4] at full size, strand-tie membership,
# concatdbs --preserve-keys) and the communicator watchdogs in place; then rescoring with the mode-specialised kernel at 4 / 5 / 6 wavefronts per SIMD
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call2; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q --timeout 1500 --durations=12 > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
run() { env "$@" timeout 300 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall 2>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); s=d['roofline']['stage_ms_per_step']
print('%-40s %.1f verify=%s | ext %.1f+%.1f part %.1f grp %.1f sort2 %.1f resc %.1f a16 %.1f a32+64 %.1f big %.1f asm %.1f' % (sys.argv[1], d['ms_per_step'], d.get('verify',{}).get('match'), s['extractShortKernel'], s['extractKernel'], s['hash_partition(all passes)'], s['groupKernel'], s['rep_sort(partition+aggSortKernel)'], s['rescoreKernel'], s['assembleGroupKernel<16>'], s['assembleGroupKernel<32>+<64>'], s['assembleBigKernel'], s['assemble_stage']))" "$*" | tee -a $O/sweep.txt; tail -2 $O/err.txt | grep -v amdgpu.ids; }
run X=0
run PLASSHIP_TUNE_RESCORE_WPE=5
run PLASSHIP_TUNE_RESCORE_WPE=6
