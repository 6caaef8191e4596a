#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call21; mkdir -p $O
timeout 600 python tools/cli_dropin_probe.py > $O/cli_dropin_probe.log 2>&1
tail -45 $O/cli_dropin_probe.log | cut -c1-260
