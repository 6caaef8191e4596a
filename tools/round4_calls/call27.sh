#!/bin/bash
# round 4, GPU call 27: the whole GPU suite on the round's final sources and tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call27; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 1400 --durations=10 > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
