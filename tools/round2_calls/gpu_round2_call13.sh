#!/bin/bash
# round 2, GPU call 13: leaner one-thread-per-sequence extraction kernel against the previous one (parity + 50 M-read bench)
mkdir -p gpurun_out/c13
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -x -q > gpurun_out/c13/pytest.log 2>&1
tail -3 gpurun_out/c13/pytest.log
for v in new old new9 new12 new18; do
  case $v in new) export PLASSHIP_TUNE_SHORT2=9; unset PLASSHIP_SHORT_V1;; old) export PLASSHIP_SHORT_V1=1;; new9) unset PLASSHIP_SHORT_V1; export PLASSHIP_TUNE_SHORT2=8;; new12) export PLASSHIP_TUNE_SHORT2=12;; new18) export PLASSHIP_TUNE_SHORT2=18;; esac
  if [ $v = new ] || [ $v = old ]; then
    timeout 600 python bench.py --no-cpu-baseline --steps 6 --warmup 0 > gpurun_out/c13/bench_$v.log 2> gpurun_out/c13/bench_$v.err
    python - $v <<'PY'
import json, sys
d = json.loads(open("gpurun_out/c13/bench_%s.log" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "value", round(d["value"] / 1e6, 1), "ms/step", round(d["ms_per_step"], 1), "extract_ms", [r["extract_ms"] for r in d["iterations"]], "short", d["roofline"]["stage_ms_per_step"].get("extractShortKernel"))
PY
  fi
  PROBE_LENGTHS=40,48,60 timeout 120 python tools/extract_probe.py 3e8 2>&1 | tail -3 | sed "s/^/$v /"
done
