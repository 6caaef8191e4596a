#!/bin/bash
# round 6, GPU call 10: the position cache of nucleotide runs (kmermatch_extract.hpp section 2d) — the nucleotide chain tests (every DB against the
# reference-pinned fixtures, cached sequences asserted), then configs[4] at full size with and without it
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call10; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_deep.py tests/test_gpu_large_nucl.py tests/test_gpu_chain_cli.py -m gpu -x -q --timeout 1200 > $O/pytest_nucl.log 2>&1; tail -5 $O/pytest_nucl.log
run() { env "$@" timeout 900 python bench.py --config c5 --steps 10 --warmup 0 --no-cpu-baseline 2>$O/err.txt | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(sys.argv[1], 'ms_per_step %.1f' % d['ms_per_step'], 'verify', d['verify'].get('match'))
for r in d['iterations']: print('   it %d %-10s %.1f ms  kmermatcher %.1f  extract %.1f  rescore %.1f  assemble %.1f  other %.1f' % (r['iteration'], r['kind'], r['ms'], r['kmermatcher_ms'], r['extract_ms'], r['rescore_ms'], r['assemble_ms'], r['aln2nucl_or_cyclecheck_ms']))
" "$*" | tee -a $O/c5_ab.txt; tail -3 $O/err.txt | grep -v amdgpu.ids; }
run X=0
run PLASSHIP_TUNE_KMCACHE=2
