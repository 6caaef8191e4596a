#!/bin/bash
# GPU call 9: unconditional score lookups in rescore / assemble / aln2nucl — parity, then the 12-step bench and c5
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_call9; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py tests/test_gpu_large_nucl.py tests/test_gpu_chain_cli.py -m gpu -q -x --timeout 1200 --durations=5 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" > $O/summary.txt
( time timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall ) > $O/bench12.log 2> $O/bench12.err
echo "bench rc=$?" >> $O/summary.txt
( time timeout 600 python bench.py --config c5 --no-cpu-baseline ) > $O/bench_c5.log 2> $O/bench_c5.err
echo "c5 rc=$?" >> $O/summary.txt
cat $O/summary.txt; tail -5 $O/pytest.log
