#!/bin/bash
# smoke() of __graft_entry__.py and the sharded bench mode in a 1-rank RCCL group (what the driver runs at round end, minus the 8-GPU node)
mkdir -p gpurun_out/smoke
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
PLASS_BENCH_FORCE_DIST=1 timeout 900 python bench.py --config c2 --no-cpu-baseline > gpurun_out/smoke/bench_sharded_c2.log 2> gpurun_out/smoke/bench_sharded_c2.err
tail -n 1 gpurun_out/smoke/bench_sharded_c2.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('sharded 1-rank c2:', round(d['value']/1e6,1), round(d['ms_per_step'],2), d.get('exchange',{}).get('communicator'), d.get('sharded_mode_error'))"
