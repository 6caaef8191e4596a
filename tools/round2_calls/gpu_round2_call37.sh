#!/bin/bash
# round 2, GPU call 37: further scheduling-region splits in the extraction kernel (variants X1: before the duplicate check, X2: before the emission, X3: both)
for v in X0 X1 X2 X3; do
  echo "== variant $v"
  PLASSHIP_LIB=$PWD/plass_amd/variants/lib$v.so PROBE_LENGTHS=100,250,400,1000,2500 timeout 150 python tools/extract_probe.py 3e8 2>&1 | tail -5
done
