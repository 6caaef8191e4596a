#!/bin/bash
# round 2, GPU call 26: wavefronts per SIMD of the extraction tiers (variants of the in-tree build H)
for v in H V1 V2 V3 H; do
  echo "== variant $v"
  PLASSHIP_LIB=$PWD/plass_amd/variants/lib$v.so PROBE_LENGTHS=100,250,400,700,1000,1500,2500 timeout 150 python tools/extract_probe.py 3e8 2>&1 | tail -7
done
