#!/bin/bash
# Builds A/B variants of libh10x_hip.so (EXTRA=-D... on stage_c.hip only) into scratch/bin/var_<name>/ — run here, before gpurun:
#   scratch/r3_variants.sh build name1="-DX=1 -DY=0" name2="..."
# and times them on the GPU box (run there, through gpurun):   scratch/r3_variants.sh run "yeast c3:0" name1 name2 ...
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  shift
  for spec in "$@"; do
    name=${spec%%=*}; flags=${spec#*=}
    mkdir -p scratch/bin/var_$name
    ( cd hash10x_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -c stage_c.hip -o ../../scratch/bin/var_$name/stage_c.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/bin/var_$name/libh10x_hip.so prim.o stage_a.o stage_b.o ../../scratch/bin/var_$name/stage_c.o stage_d.o stage_e.o shard.o comm.o h10x_api.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib )
    echo "built $name ($flags)"
  done
else
  shift; what=$1; shift
  cp hash10x_amd/libh10x_hip.so /tmp/libh10x_hip.orig.so
  for name in "$@"; do
    cp scratch/bin/var_$name/libh10x_hip.so hash10x_amd/libh10x_hip.so
    echo "=== variant $name"
    if [ "$what" = full ]; then timeout -k 10 300 python scratch/r3_full.py 2 2>&1 | grep -E "^pass 1|cluster_kernel" | tail -2
    else timeout -k 10 300 python scratch/r3_time.py $what 2>&1 | grep -v "^  mode"; fi
  done
  cp /tmp/libh10x_hip.orig.so hash10x_amd/libh10x_hip.so
fi
