#!/bin/bash
# round 4, GPU call 33: lane utilisation of the vector instructions per kernel on the 50 M-read chain (SQ_THREAD_CYCLES_VALU against SQ_ACTIVE_INST_VALU)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call33; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SALU"
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc $P -d $R/$O/pmc -o q -- python $R/bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-verify --no-wall ) > $O/pmc.log 2>&1
DB=$(find $O/pmc -name '*_results.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" 40 > $O/lanes_summary.txt 2>&1
rm -rf $O/pmc
head -20 $O/lanes_summary.txt | cut -c1-330
