#!/bin/bash
# GPU call 23 (last of the round): next residues requested ahead in the thread-per-sequence extraction — parity, then the evidence of the
# driver's command again (kernel trace, the two PMC traffic passes, the bench line with the regenerated traffic file in place)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_call23; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -q -x --timeout 500 > $O/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc" > $O/summary.txt; tail -2 $O/pytest.log
[ $rc -ne 0 ] && exit 0
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o driver -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-wall > $R/$O/bench_driver_cmd_rocprof.log 2> $R/$O/bench_driver_cmd_rocprof.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d $R/$O/pmc_$c -o pmc -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-wall > $R/$O/pmc_$c.log 2> $R/$O/pmc_$c.err
done
cd $R
python tools/rocpd_summary.py $O/prof/driver_results.db --timeline 1 > $O/kernel_stats_driver_cmd.txt 2>&1
python tools/rocpd_traffic.py $O/pmc_FETCH_SIZE/pmc_results.db $O/pmc_WRITE_SIZE/pmc_results.db 50 $O/pmc_traffic.json 37 "python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-wall" > $O/pmc_hbm_traffic.txt 2>&1
cp $O/pmc_traffic.json profiles/r03_pmc_traffic.json
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2> $O/bench_driver_cmd.err
tail -c 200 $O/bench_driver_cmd.log; echo
find $O -name "*.db" -delete
