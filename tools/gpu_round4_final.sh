#!/bin/bash
# round 4, evidence run: full GPU suite, command-line check, the driver's bench command plain and under rocprofv3 --kernel-trace, the two PMC
# traffic passes of the SAME command, the other configs (c2, c5 incl. its PMC passes), the sharded path in a 1-rank group against the
# single-GPU path at 12.5 M reads.  Everything lands under gpurun_out/r04_final/ and is copied into profiles/ by hand.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_final; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q --timeout 1400 --durations=10 > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 bash tests/gpu_cli_check.sh > $O/cli_check.log 2>&1; tail -2 $O/cli_check.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $R/$O/pmc_$c -o pmc -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-wall > $R/$O/pmc_$c.log 2> $R/$O/pmc_$c.err
done
cd $R
# 20 timed + 5 warm-up + 12 verification iterations
python tools/rocpd_traffic.py $O/pmc_FETCH_SIZE/pmc_results.db $O/pmc_WRITE_SIZE/pmc_results.db 50 profiles/r04_pmc_traffic.json 37 "python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-wall" > $O/pmc_hbm_traffic.txt 2>&1
cp profiles/r04_pmc_traffic.json $O/pmc_traffic.json
# the driver's command AFTER the PMC passes, so that roofline.traffic is quoted from them (bench.PMC_FILES)
PLASS_BENCH_VERBOSE=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2> $O/bench_driver_cmd.err; tail -c 300 $O/bench_driver_cmd.log; echo
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o driver -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-wall > $R/$O/bench_driver_cmd_rocprof.log 2> $R/$O/bench_driver_cmd_rocprof.err
cd $R
python tools/rocpd_summary.py $O/prof/driver_results.db --timeline 1 > $O/kernel_stats_driver_cmd.txt 2>&1
head -14 $O/kernel_stats_driver_cmd.txt | cut -c1-150
# c5: PMC passes (10 timed + 10 verification steps), then the bench line
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $R/$O/pmc5_$c -o pmc -- python $R/bench.py --config c5 --steps 10 --warmup 0 --no-cpu-baseline > $R/$O/pmc5_$c.log 2> $R/$O/pmc5_$c.err
done
cd $R
python tools/rocpd_traffic.py $O/pmc5_FETCH_SIZE/pmc_results.db $O/pmc5_WRITE_SIZE/pmc_results.db 30 profiles/r04_pmc_traffic_c5.json 20 "python bench.py --config c5 --steps 10 --warmup 0 --no-cpu-baseline" > $O/pmc_hbm_traffic_c5.txt 2>&1
cp profiles/r04_pmc_traffic_c5.json $O/pmc_traffic_c5.json
timeout 900 python bench.py --config c5 > $O/bench_c5.log 2> $O/bench_c5.err; tail -c 300 $O/bench_c5.log; echo
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $R/$O/prof5 -o c5 -- python $R/bench.py --config c5 --steps 10 --warmup 0 --no-cpu-baseline --no-verify > $R/$O/bench_c5_rocprof.log 2> $R/$O/bench_c5_rocprof.err
cd $R
python tools/rocpd_summary.py $O/prof5/c5_results.db > $O/kernel_stats_c5.txt 2>&1
timeout 300 python bench.py --config c2 --no-wall > $O/bench_c2.log 2>/dev/null
# sharded orchestration in a 1-rank native-RCCL group against the single-GPU path, 12.5 M reads
timeout 400 python bench.py --pairs 6250000 --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify > $O/bench_12M_single.log 2>/dev/null
MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 PLASS_BENCH_FORCE_DIST=1 timeout 400 python bench.py --pairs 6250000 --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify > $O/bench_12M_sharded_1rank.log 2> $O/bench_12M_sharded_1rank.err
find $O -name "*.db" -size +30M -delete
ls $O
