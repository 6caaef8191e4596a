#!/bin/bash
# round 2, GPU call 23: extraction kernel code-generation variants (plass_amd/variants/lib?.so), uniform-length probe
for v in A B C D A D; do
  echo "== variant $v"
  PLASSHIP_LIB=$PWD/plass_amd/variants/lib$v.so PROBE_LENGTHS=250,400,700,1000,1500,2500 timeout 150 python tools/extract_probe.py 3e8 2>&1 | tail -6
done
