#!/bin/bash
# round 4, GPU call 35 (last): the two PMC traffic passes and the driver's command on the final sources (rescoring with per-lane wraps)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call35; mkdir -p $O
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d $R/$O/pmc_$c -o pmc -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-wall --no-verify > $R/$O/pmc_$c.log 2> $R/$O/pmc_$c.err
done
cd $R
# 20 timed + 5 warm-up iterations (no verification traversal in these passes)
python tools/rocpd_traffic.py $O/pmc_FETCH_SIZE/pmc_results.db $O/pmc_WRITE_SIZE/pmc_results.db 50 profiles/r04_pmc_traffic.json 25 "python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-wall --no-verify" > $O/pmc_hbm_traffic.txt 2>&1
cp profiles/r04_pmc_traffic.json $O/pmc_traffic.json
PLASS_BENCH_VERBOSE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2> $O/bench_driver_cmd.err; tail -c 200 $O/bench_driver_cmd.log; echo
find $O -name "*.db" -size +30M -delete
