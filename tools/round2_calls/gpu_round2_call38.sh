#!/bin/bash
# round 2, GPU call 38: 12.5 M reads of the C3 model through the single-GPU path (line store) and through the sharded path (dense
# partition, 1-rank RCCL group): the per-iteration counts and the final DB must agree
mkdir -p gpurun_out/c38
timeout 900 python bench.py --no-cpu-baseline --pairs 6250000 --steps 6 --warmup 0 > gpurun_out/c38/single.log 2> gpurun_out/c38/single.err
PLASS_BENCH_FORCE_DIST=1 timeout 900 python bench.py --no-cpu-baseline --pairs 6250000 --steps 6 --warmup 0 > gpurun_out/c38/sharded.log 2> gpurun_out/c38/sharded.err
python - <<'PY'
import json
a = json.loads(open("gpurun_out/c38/single.log").read().strip().splitlines()[-1])
b = json.loads(open("gpurun_out/c38/sharded.log").read().strip().splitlines()[-1])
keys = ("N_k", "N_m", "N_c", "verified", "extended", "residues")
ok = True
for x, y in zip(a["iterations"], b["iterations"]):
    same = all(x[k] == y[k] for k in keys)
    ok &= same
    print(x["iteration"], "same" if same else "DIFFERENT", [x[k] for k in keys], round(x["ms"], 1), round(y["ms"], 1))
print("final DB", a["config"]["final_db"], b["config"]["final_db"], "AGREE" if ok and a["config"]["final_db"] == b["config"]["final_db"] else "DISAGREE")
PY
