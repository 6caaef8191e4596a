#!/bin/bash
# round 4, GPU call 26: the one test of the evidence run that failed on a bug of the TEST (numpy array from bytes), fixed
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call26; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 500 -x -k "upload_refuses or adversarial or golden_aa" > $O/pytest_upload.log 2>&1; tail -3 $O/pytest_upload.log
