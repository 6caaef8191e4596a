#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c6
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY"
P2="SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $P -d $R/gpurun_out/c6/pmc$i -o q$i -- python $R/bench.py --pairs 5000000 --steps 12 --warmup 0 --no-cpu-baseline ) > gpurun_out/c6/pmc$i.log 2>&1
  DB=$(find gpurun_out/c6/pmc$i -name '*_results.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" 24 > gpurun_out/c6/pmc${i}_summary.txt 2>&1
  find gpurun_out/c6/pmc$i -name '*.db' -delete
done
( PLASS_BENCH_VERBOSE=1 timeout 300 python bench.py --pairs 5000000 --steps 12 --warmup 0 --no-cpu-baseline ) > gpurun_out/c6/bench5M.log 2> gpurun_out/c6/bench5M.err
head -60 gpurun_out/c6/pmc1_summary.txt | cut -c1-330; tail -3 gpurun_out/c6/pmc2.log
