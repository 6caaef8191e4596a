#!/bin/bash
# round 2, GPU call 24: extraction code-generation variants E, F against A and D; LDS-scores tier for 1024..2035 windows (G)
for v in A D E F; do
  echo "== variant $v"
  PLASSHIP_LIB=$PWD/plass_amd/variants/lib$v.so PROBE_LENGTHS=250,400,700,1000 timeout 150 python tools/extract_probe.py 3e8 2>&1 | tail -4
done
for t in 0 8 13; do
  echo "== variant G tier2a=$t"
  PLASSHIP_TIER2A=$t PLASSHIP_LIB=$PWD/plass_amd/variants/libG.so PROBE_LENGTHS=1200,1500,2000,2500 timeout 150 python tools/extract_probe.py 3e8 2>&1 | tail -4
done
