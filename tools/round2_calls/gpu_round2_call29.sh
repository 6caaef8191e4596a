#!/bin/bash
# round 2, GPU call 29: wavefronts per SIMD of the extension kernels (PLASSHIP_TUNE_ASM16 / ASM64 / ASMBIG), 6 steps each
mkdir -p gpurun_out/c29
run() {
  env "$@" timeout 600 python bench.py --no-cpu-baseline --steps 6 --warmup 0 > gpurun_out/c29/b.log 2> gpurun_out/c29/b.err
  python - "$*" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/c29/b.log").read().strip().splitlines()[-1])
s = d["roofline"]["stage_ms_per_step"]
print(sys.argv[1], "| ms/step", round(d["ms_per_step"], 1), "asm16", s["assembleGroupKernel<16>"], "asm32+64", s["assembleGroupKernel<32>+<64>"], "big", s["assembleBigKernel"])
PY
}
run X=0
run PLASSHIP_TUNE_ASM16=5 PLASSHIP_TUNE_ASM64=4 PLASSHIP_TUNE_ASMBIG=5
run PLASSHIP_TUNE_ASM16=6 PLASSHIP_TUNE_ASM64=5 PLASSHIP_TUNE_ASMBIG=4
run PLASSHIP_TUNE_ASM16=4 PLASSHIP_TUNE_ASM64=3 PLASSHIP_TUNE_ASMBIG=8
