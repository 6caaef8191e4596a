#!/bin/bash
# GPU call 22: the counters of call 8 / 11 once more on the final sources (before / after per kernel)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_call22; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P2="SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
P3="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $P -d $R/$O/pmc$i -o q$i -- python $R/bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-verify --no-wall ) > $O/pmc$i.log 2>&1
  DB=$(find $O/pmc$i -name '*_results.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" 44 > $O/pmc${i}_summary.txt 2>&1
  rm -rf $O/pmc$i
  grep -c avg_us $O/pmc${i}_summary.txt
done
