#!/bin/bash
# round 6, GPU call 35: the position cache's kernel with every link of its chain a sequence earlier than the next (list entry 4 ahead ... positions 1 ahead)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call35; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deep.py tests/test_gpu_large_nucl.py tests/test_gpu_chain_cli.py -m gpu -x -q --timeout 1200 -k "nucl or circular or chain or cycl" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
run5() { env "$@" timeout 900 python bench.py --config c5 --steps 10 --warmup 0 --no-cpu-baseline 2>$O/err.txt | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['roofline']['stage_ms_per_step']
print('c5 %-40s %.1f verify %s | ext %.1f+%.1f km %.1f' % (sys.argv[1], d['ms_per_step'], d['verify'].get('match'), s['extractShortKernel'], s['extractKernel'], s['kmermatcher_stage']))
print('      extraction per step: ' + ' '.join('%.1f' % r.get('extract_ms', -1) for r in d['iterations']))" "$*" | tee -a $O/sweep.txt; }
run5 X=0
run5 PLASSHIP_TUNE_CACHEDPOS=64
run5 PLASSHIP_TUNE_KMCACHE=2
