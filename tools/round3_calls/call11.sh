#!/bin/bash
# GPU call 11: memory-side counters per kernel (TLB, L1/L2 requests, DRAM request sizes, latencies) on the 12 iterations
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_call11; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PA="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum"
PB="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
PC="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum"
PD="TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_WRREQ_64B_sum"
PE="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum"
PF="TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_avr TCC_EA0_WRREQ_STALL_sum"
i=0
for P in "$PA" "$PB" "$PC" "$PD" "$PE" "$PF"; do
  i=$((i+1))
  ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc $P -d $R/$O/pmc$i -o q$i -- python $R/bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-verify --no-wall ) > $O/pmc$i.log 2>&1
  DB=$(find $O/pmc$i -name '*_results.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_pmc.py "$DB" 40 > $O/pmc${i}_summary.txt 2>&1
  rm -rf $O/pmc$i
  grep -c "avg_us" $O/pmc${i}_summary.txt
done
