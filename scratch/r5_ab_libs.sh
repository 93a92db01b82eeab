#!/bin/bash
# same-box A/B of library builds (build/ab/<name>.so) with scratch/r5_opts.py:   scratch/r5_ab_libs.sh "<workload> <passes> <option sets...>" name1 name2 ...   (DICT=1: keep the per-kernel times)
ARGS=$1; shift
for v in "$@"; do
  cp build/ab/$v.so hash10x_amd/libh10x_hip.so
  echo "=== $v"
  if [ -n "$DICT" ]; then python3 scratch/r5_opts.py $ARGS 2>&1 | grep -v "^generated"; else python3 scratch/r5_opts.py $ARGS 2>&1 | grep -v "^      {\|^generated"; fi
done
