#!/bin/bash
# round 6, GPU call 1: the persistent lane-refill rescoring kernel (rescore.hip, rescoreRefillKernel) — parity first, then the A/B against the
# lock-step kernel and a sweep of its refill threshold on the 12 iterations of the headline chain (every row verified against the committed digests)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call1; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_orfs.py -m gpu -x -q --timeout 600 > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
run() { env "$@" timeout 300 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall 2>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); s=d['roofline']['stage_ms_per_step']
print('%-40s %.1f verify=%s | ext %.1f+%.1f part %.1f grp %.1f sort2 %.1f resc %.1f a16 %.1f a32+64 %.1f big %.1f asm %.1f' % (sys.argv[1], d['ms_per_step'], d.get('verify',{}).get('match'), s['extractShortKernel'], s['extractKernel'], s['hash_partition(all passes)'], s['groupKernel'], s['rep_sort(partition+aggSortKernel)'], s['rescoreKernel'], s['assembleGroupKernel<16>'], s['assembleGroupKernel<32>+<64>'], s['assembleBigKernel'], s['assemble_stage']))" "$*" | tee -a $O/sweep.txt; tail -2 $O/err.txt; }
run PLASSHIP_TUNE_RESCORE_REFILL=2
run PLASSHIP_TUNE_RESCORE_T=32
run PLASSHIP_TUNE_RESCORE_T=16
run PLASSHIP_TUNE_RESCORE_T=48
run PLASSHIP_TUNE_RESCORE_T=64
run PLASSHIP_TUNE_RESCORE_T=24 PLASSHIP_TUNE_RESCORE_REFILL_WPE=3
run PLASSHIP_TUNE_RESCORE_T=32 PLASSHIP_TUNE_RESCORE_REFILL_WPE=5
