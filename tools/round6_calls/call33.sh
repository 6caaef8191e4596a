#!/bin/bash
# round 6, GPU call 33: cyclecheck's table kernels with look-ahead and 16-byte staging loads (configs[4])
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_call33; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deep.py tests/test_gpu_large_nucl.py tests/test_gpu_chain_cli.py -m gpu -x -q --timeout 1200 -k "cycl or nucl or circular or chain" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
run5() { env "$@" timeout 900 python bench.py --config c5 --steps 10 --warmup 0 --no-cpu-baseline 2>$O/err.txt | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['roofline']['stage_ms_per_step']
print('c5 %-40s %.1f verify %s | km %.1f asm %.1f aln2nucl/cyclecheck %.1f' % (sys.argv[1], d['ms_per_step'], d['verify'].get('match'), s['kmermatcher_stage'], s['assemble_stage'], s['proteinaln2nucl / cyclecheck']))
print('      aln2nucl / cyclecheck per step: ' + ' '.join('%.1f' % r.get('aln2nucl_or_cyclecheck_ms', -1) for r in d['iterations']))" "$*" | tee -a $O/sweep.txt; }
run5 X=0
run5 X=0
