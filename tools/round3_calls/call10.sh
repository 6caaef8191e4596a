#!/bin/bash
# GPU call 10: rescore with pipelined metadata / early first blocks, writeOut with 4 sequences in flight, coalesced compactAln:
# parity, then A/B of the knobs on the 12 iterations
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_call10; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py tests/test_gpu_large_nucl.py tests/test_gpu_chain_cli.py -m gpu -q -x --timeout 1200 --durations=5 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" > $O/summary.txt
B="python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall"
( timeout 600 $B ) > $O/bench_default.log 2> $O/bench_default.err
for kv in RESCORE_WPE=4 RESCORE_WPE=6 WRITEOUT_U=1 WRITEOUT_U=2; do
  ( export PLASSHIP_TUNE_$kv; timeout 600 $B --no-verify ) > $O/bench_$kv.log 2> $O/bench_$kv.err
done
cat $O/summary.txt; tail -4 $O/pytest.log
