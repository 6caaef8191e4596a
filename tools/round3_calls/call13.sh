#!/bin/bash
# GPU call 13: which pairs the thread-per-pair rescore kernel keeps (min(qLen, tLen) <= RESCORE_SHORT), 12 iterations each
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_call13; mkdir -p $O
B="python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall"
for kv in RESCORE_SHORT=96 RESCORE_SHORT=128 RESCORE_SHORT=192 RESCORE_SHORT=256; do
  ( export PLASSHIP_TUNE_$kv; timeout 600 $B ) > $O/bench_$kv.log 2> $O/bench_$kv.err
done
ls $O
