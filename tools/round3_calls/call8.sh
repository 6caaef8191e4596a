#!/bin/bash
# GPU call 8: SQ / TCC counters per kernel on the 12 iterations of the 50 M-read chain (what bounds rescore, group, the extension kernels)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_call8; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCP_[A-Z_0-9a-z]*\|TCC_[A-Z_0-9a-z]*\|TA_[A-Z_0-9a-z]*" | sort -u | tr '\n' ' ' ) > $O/counters_available.txt 2>&1
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P2="SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
P4="TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc $P -d $R/$O/pmc$i -o q$i -- python $R/bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-verify --no-wall ) > $O/pmc$i.log 2>&1
  DB=$(find $O/pmc$i -name '*_results.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" 40 > $O/pmc${i}_summary.txt 2>&1
  rm -rf $O/pmc$i
done
tail -2 $O/pmc1.log | cut -c1-300
