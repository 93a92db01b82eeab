#!/bin/bash
# Same-box A/B of two builds of libh10x_hip.so: build/ab/base.so and build/ab/var.so take turns as hash10x_amd/libh10x_hip.so, the yeast bench runs on each
# (boxes of the pool differ by a few per cent: a 3 % effect cannot be judged across calls).   scratch/ab.sh [rounds] [extra bench args]
R=${1:-3}; shift
for r in $(seq $R); do
  for v in base var; do
    cp build/ab/$v.so hash10x_amd/libh10x_hip.so
    python3 bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); d=b['device_ms_per_step']; print('$v', round(b['ms_per_step'],3), 'main', d['cluster_main'], 'cluster', d['cluster'], 'mosh', d['mosh_extract'], b['build_id'])"
  done
done
