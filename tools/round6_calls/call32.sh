#!/bin/bash
# round 6, GPU call 32: slack of a sequence heap as a multiple of the data (a full copy of the DB, 14.6 ms, happens when the heap is full: 13 of 37 steps)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call32; mkdir -p $O
export PYTHONUNBUFFERED=1
run() { env "$@" PLASSHIP_POOL_STATS=1 timeout 400 python bench.py --steps 24 --warmup 0 --no-cpu-baseline --no-wall --no-verify 2>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); s=d['roofline']['stage_ms_per_step']
print('%-54s %.1f | asm %.1f' % (sys.argv[1], d['ms_per_step'], s['assemble_stage']))
print('      assemble per step: ' + ' '.join('%.0f' % r.get('assemble_ms', -1) for r in d['iterations']))" "$*" | tee -a $O/sweep.txt; grep -i "pool\|arena\|peak" $O/err.txt | tail -3 | cut -c1-200; }
run X=0
run PLASSHIP_TUNE_DBHEAP_X10=20 PLASSHIP_TUNE_DBHEAP_GB=32
run PLASSHIP_TUNE_DBHEAP_X10=30 PLASSHIP_TUNE_DBHEAP_GB=48
