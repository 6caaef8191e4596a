#!/bin/bash
# round 2, evidence run: full GPU suite, command-line check, the driver's bench command plain and under rocprofv3 --kernel-trace, PMC traffic
# passes, probes.  Everything lands under gpurun_out/final/ and is copied into profiles/ by hand.
O=gpurun_out/final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 900 bash tests/gpu_cli_check.sh > $O/cli_check.log 2>&1; tail -2 $O/cli_check.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2> $O/bench_driver_cmd.err; tail -c 400 $O/bench_driver_cmd.log; echo
timeout 600 python bench.py --steps 12 --warmup 0 --no-cpu-baseline > $O/bench_c3_12steps.log 2>/dev/null
timeout 300 python bench.py --config c2 > $O/bench_c2.log 2>/dev/null
PROBE_LENGTHS=48,60,80,100,150,250,400,700,1000,1500,2500 timeout 200 python tools/extract_probe.py 3e8 > $O/extract_probe.log 2>&1
timeout 300 python tools/bench_nucl.py 500000 3 > $O/bench_nucl.log 2>&1
timeout 300 python tools/io_probe.py 2500000 > $O/io_probe.log 2>&1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o driver -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/$O/bench_driver_cmd_rocprof.log 2> $R/$O/bench_driver_cmd_rocprof.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $R/$O/pmc_$c -o pmc -- python $R/bench.py --steps 12 --warmup 0 --no-cpu-baseline > $R/$O/pmc_$c.log 2> $R/$O/pmc_$c.err
done
cd $R
python tools/rocpd_summary.py $O/prof/driver_results.db > $O/kernel_stats_driver_cmd.txt 2>&1
python tools/rocpd_traffic.py $O/pmc_FETCH_SIZE/pmc_results.db $O/pmc_WRITE_SIZE/pmc_results.db 40 $O/pmc_traffic.json 12 > $O/pmc_hbm_traffic.txt 2>&1
head -12 $O/kernel_stats_driver_cmd.txt | cut -c1-140
find $O -name "*.db" -size +30M -delete
ls $O
