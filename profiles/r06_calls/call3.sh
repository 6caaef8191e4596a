#!/bin/bash
# round 6, GPU call 3: configs[4] at full size as a test of its own; double hashing in the group kernel's LDS table (A/B); where the fused driver's
# wall clock goes (phase timings of reading / writing the 4.7 GB fragment DB: PLASSHIP_IO_TIMING)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call3; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_c5_full.py -m gpu -x -q --timeout 1500 > $O/pytest_c5.log 2>&1; tail -3 $O/pytest_c5.log
run() { env "$@" timeout 300 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall 2>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); s=d['roofline']['stage_ms_per_step']
print('%-40s %.1f verify=%s | ext %.1f+%.1f part %.1f grp %.1f sort2 %.1f resc %.1f a16 %.1f a32+64 %.1f big %.1f asm %.1f' % (sys.argv[1], d['ms_per_step'], d.get('verify',{}).get('match'), s['extractShortKernel'], s['extractKernel'], s['hash_partition(all passes)'], s['groupKernel'], s['rep_sort(partition+aggSortKernel)'], s['rescoreKernel'], s['assembleGroupKernel<16>'], s['assembleGroupKernel<32>+<64>'], s['assembleBigKernel'], s['assemble_stage']))" "$*" | tee -a $O/sweep.txt; tail -2 $O/err.txt | grep -v amdgpu.ids; }
run X=0
run PLASSHIP_TUNE_GROUP_PROBE=2
timeout 600 python tools/chain_wall_probe.py > $O/wall_probe.log 2>&1; grep -E "plasship io|chain:|wall|pool" $O/wall_probe.log | tail -40
