#!/bin/bash
# round 6, evidence run: full GPU suite, command-line check, smoke, the two PMC traffic passes of the driver's command, the driver's command itself
# (after the PMC passes, so that roofline.traffic is quoted from them), the same command under rocprofv3 --kernel-trace, the SQ lane pass, the other
# configs (c5 incl. its PMC passes, c2), the sharded path in a 1-rank group against the single-GPU path at 12.5 M reads.  Everything lands under
# gpurun_out/r06_final/ and is copied into profiles/ by hand.  Sections can be skipped: SKIP="tests c5" bash tools/gpu_round6_final.sh
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_final; mkdir -p $O
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
skip() { case " $SKIP " in *" $1 "*) return 0;; esac; return 1; }
if ! skip tests; then
  timeout 2400 python -m pytest tests -m gpu -q --timeout 1500 --durations=10 > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
  timeout 600 bash tests/gpu_cli_check.sh > $O/cli_check.log 2>&1; tail -2 $O/cli_check.log
  timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
fi
if ! skip pmc; then
  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/$O/pmc_$c -o pmc -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-wall --no-verify > $R/$O/pmc_$c.log 2> $R/$O/pmc_$c.err
  done
  cd $R
  python tools/rocpd_traffic.py $O/pmc_FETCH_SIZE/pmc_results.db $O/pmc_WRITE_SIZE/pmc_results.db 60 profiles/r06_pmc_traffic.json 25 "python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-wall --no-verify" > $O/pmc_hbm_traffic.txt 2>&1
  cp profiles/r06_pmc_traffic.json $O/pmc_traffic.json
fi
if ! skip bench; then
  PLASS_BENCH_VERBOSE=1 timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2> $O/bench_driver_cmd.err; tail -c 300 $O/bench_driver_cmd.log; echo
fi
if ! skip trace; then
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o driver -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-wall > $R/$O/bench_driver_cmd_rocprof.log 2> $R/$O/bench_driver_cmd_rocprof.err
  cd $R
  python tools/rocpd_summary.py $O/prof/driver_results.db --timeline 1 > $O/kernel_stats_driver_cmd.txt 2>&1
  head -16 $O/kernel_stats_driver_cmd.txt | cut -c1-150
fi
if ! skip lanes; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU -d $R/$O/pmc_lanes -o pmc -- python $R/bench.py --gpus 1 --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify > $R/$O/pmc_lanes.log 2> $R/$O/pmc_lanes.err
  cd $R
  python tools/rocpd_pmc.py $O/pmc_lanes/pmc_results.db > $O/pmc_lanes_per_kernel.txt 2>&1
fi
if ! skip c5; then
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 700 rocprofv3 --kernel-trace --pmc $c -d $R/$O/pmc5_$c -o pmc -- python $R/bench.py --config c5 --steps 10 --warmup 0 --no-cpu-baseline > $R/$O/pmc5_$c.log 2> $R/$O/pmc5_$c.err
  done
  cd $R
  python tools/rocpd_traffic.py $O/pmc5_FETCH_SIZE/pmc_results.db $O/pmc5_WRITE_SIZE/pmc_results.db 40 profiles/r06_pmc_traffic_c5.json 20 "python bench.py --config c5 --steps 10 --warmup 0 --no-cpu-baseline" > $O/pmc_hbm_traffic_c5.txt 2>&1
  cp profiles/r06_pmc_traffic_c5.json $O/pmc_traffic_c5.json
  timeout 900 python bench.py --config c5 > $O/bench_c5.log 2> $O/bench_c5.err; tail -c 300 $O/bench_c5.log; echo
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace -d $R/$O/prof5 -o c5 -- python $R/bench.py --config c5 --steps 10 --warmup 0 --no-cpu-baseline --no-verify > $R/$O/bench_c5_rocprof.log 2> $R/$O/bench_c5_rocprof.err
  cd $R
  python tools/rocpd_summary.py $O/prof5/c5_results.db > $O/kernel_stats_c5.txt 2>&1
fi
if ! skip small; then
  timeout 300 python bench.py --config c2 --no-wall > $O/bench_c2.log 2>/dev/null
  timeout 400 python bench.py --pairs 6250000 --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify > $O/bench_12M_single.log 2>/dev/null
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 PLASS_BENCH_FORCE_DIST=1 timeout 400 python bench.py --pairs 6250000 --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify > $O/bench_12M_sharded_1rank.log 2> $O/bench_12M_sharded_1rank.err
fi
find $O -name "*.db" -size +30M -delete
ls $O
