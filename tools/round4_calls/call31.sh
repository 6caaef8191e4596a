#!/bin/bash
# round 4, GPU call 31: SQ counters per kernel on configs[4] (what bounds the round's new kernels: lane-per-queue extension, LDS-table cyclecheck, nucleotide extraction tiers)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call31; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P2="SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $P -d $R/$O/pmc$i -o q$i -- python $R/bench.py --config c5 --steps 10 --warmup 0 --no-cpu-baseline --no-verify ) > $O/pmc$i.log 2>&1
  DB=$(find $O/pmc$i -name '*_results.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" 40 > $O/pmc${i}_summary.txt 2>&1
  rm -rf $O/pmc$i
done
head -12 $O/pmc1_summary.txt | cut -c1-220
