#!/bin/bash
# round 6, GPU call 13: the row tier with blocks that leave before their set-up when the list is shorter than the grid; counters of the extraction kernels
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call13; mkdir -p $O
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
run() { env "$@" timeout 400 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall 2>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); s=d['roofline']['stage_ms_per_step']
print('%-44s %.1f verify %s | ext %.1f+%.1f part %.1f grp %.1f sort2 %.1f resc %.1f asm %.1f' % (sys.argv[1], d['ms_per_step'], d['verify'].get('match'), s['extractShortKernel'], s['extractKernel'], s['hash_partition(all passes)'], s['groupKernel'], s['rep_sort(partition+aggSortKernel)'], s['rescoreKernel'], s['assemble_stage']))
print('      extraction per iteration: ' + ' '.join('%.1f' % r.get('extract_ms', -1) for r in d['iterations']))" "$*" | tee -a $O/sweep.txt; tail -2 $O/err.txt | grep -v amdgpu.ids; }
run X=0
for g in 32 96 128 192 256; do run PLASSHIP_TUNE_ROWGRID=$g; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU -d $R/$O/pmc_a -o pmc -- python $R/bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify > $R/$O/pmc_a.log 2> $R/$O/pmc_a.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/$O/pmc_b -o pmc -- python $R/bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify > $R/$O/pmc_b.log 2> $R/$O/pmc_b.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $R/$O/pmc_c -o pmc -- python $R/bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify > $R/$O/pmc_c.log 2> $R/$O/pmc_c.err
cd $R
for x in a b c; do python tools/rocpd_pmc.py $O/pmc_$x/pmc_results.db 30 > $O/pmc_$x.txt 2>&1; done
grep -A1 "extract" $O/pmc_a.txt | cut -c1-400
find $O -name "*.db" -size +30M -delete
