#!/bin/bash
# Same-box A/B of library builds on BASELINE configs[2] at full size (scratch/full_step.py, 2 passes each; H10X_WL=genome3g-300M for the 3 Gb set):
#   scratch/ab_full.sh [build ...]      builds = names under build/ab/ without .so; default: base var base var
V=${@:-base var base var}
for v in $V; do
  cp build/ab/$v.so hash10x_amd/libh10x_hip.so
  python3 scratch/full_step.py 2 2>/dev/null | grep -A2 "^pass 1" | grep "cluster_kernel\|pass 1" | sed "s/^/$v /" | cut -c1-200
done
