#!/bin/bash
# round 6, GPU call 7: grid of the wave-per-sequence extraction tiers (one-wavefront workgroups per CU) on the 50 M-read chain
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call7; mkdir -p $O
export PYTHONUNBUFFERED=1
run() { env "$@" timeout 300 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify 2>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); s=d['roofline']['stage_ms_per_step']
print('%-44s %.1f | ext %.1f+%.1f part %.1f grp %.1f sort2 %.1f resc %.1f a16 %.1f a32+64 %.1f big %.1f asm %.1f' % (sys.argv[1], d['ms_per_step'], s['extractShortKernel'], s['extractKernel'], s['hash_partition(all passes)'], s['groupKernel'], s['rep_sort(partition+aggSortKernel)'], s['rescoreKernel'], s['assembleGroupKernel<16>'], s['assembleGroupKernel<32>+<64>'], s['assembleBigKernel'], s['assemble_stage']))" "$*" | tee -a $O/sweep.txt; tail -2 $O/err.txt | grep -v amdgpu.ids; }
for b in 40 64 96 128 192 256 512; do run PLASSHIP_EXTRACT_BLOCKS_PER_CU=$b; done
