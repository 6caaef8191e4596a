#!/bin/bash
# GPU call 18: word-wise tails of the copies in the extension kernels / writeOut: parity, then wavefront counts and grids again
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_call18; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py tests/test_gpu_large_nucl.py tests/test_gpu_sharded.py -m gpu -q -x --timeout 1200 --durations=5 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" > $O/summary.txt
B="python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall"
( timeout 600 $B ) > $O/bench_default.log 2> $O/bench_default.err
for kv in ASM16=5 ASM16=4 ASM64=3 ASM64=5 SHORT=72 WRITEOUT=32; do
  ( export PLASSHIP_TUNE_$kv; timeout 600 $B --no-verify ) > $O/bench_$kv.log 2> $O/bench_$kv.err
done
cat $O/summary.txt; tail -4 $O/pytest.log
