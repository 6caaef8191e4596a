#!/bin/bash
# round 6, GPU call 30 (probe, timing only): a second copy of the thread-per-sequence kernel on a side stream beside the wave tiers — does a write-bound kernel fit under the hashing-bound ones?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call30; mkdir -p $O
export PYTHONUNBUFFERED=1
run() { env "$@" timeout 400 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-wall --no-verify 2>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); s=d['roofline']['stage_ms_per_step']
print('%-44s %.1f | short %.1f wave tiers %.1f' % (sys.argv[1], d['ms_per_step'], s['extractShortKernel'], s['extractKernel']))
print('      extraction per iteration: ' + ' '.join('%.1f' % r.get('extract_ms', -1) for r in d['iterations']))" "$*" | tee -a $O/sweep.txt; tail -2 $O/err.txt | grep -v amdgpu.ids; }
run X=0
run PLASSHIP_TUNE_EXT_OVERLAP_PROBE=1
