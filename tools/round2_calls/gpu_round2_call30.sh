#!/bin/bash
# round 2, GPU call 30: grid sweeps of the persistent kernels on the 50 M-read set (6 steps each)
mkdir -p gpurun_out/c30
run() {
  env "$@" timeout 600 python bench.py --no-cpu-baseline --steps 6 --warmup 0 > gpurun_out/c30/b.log 2> gpurun_out/c30/b.err
  python - "$*" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/c30/b.log").read().strip().splitlines()[-1])
s = d["roofline"]["stage_ms_per_step"]
print(sys.argv[1], "| ms/step", round(d["ms_per_step"], 1), "group", s["groupKernel"], "repsort", s["rep_sort(partition+aggSortKernel)"], "rescore", s["rescoreKernel"], "short", s["extractShortKernel"], "asm", [round(x, 1) for x in d["roofline"]["module_wall_ms_per_step"]])
PY
}
run X=0
run PLASSHIP_TUNE_GROUP=4 PLASSHIP_TUNE_AGGSORT=8 PLASSHIP_TUNE_RESCORE=8 PLASSHIP_TUNE_WRITEOUT=8
run PLASSHIP_TUNE_GROUP=8 PLASSHIP_TUNE_AGGSORT=32 PLASSHIP_TUNE_RESCORE=16 PLASSHIP_TUNE_WRITEOUT=32
run PLASSHIP_TUNE_GROUP=12 PLASSHIP_TUNE_AGGSORT=4 PLASSHIP_TUNE_RESCORE=24 PLASSHIP_TUNE_WRITEOUT=64
run PLASSHIP_TUNE_GROUP=2 PLASSHIP_TUNE_AGGSORT=64 PLASSHIP_TUNE_RESCORE=32 PLASSHIP_TUNE_SHORT=36
