#!/bin/bash
# Profile passes of one build on the GPU box (run through gpurun from the repo root):
#   scratch/prof_all.sh <tag> [script.py]       default script: scratch/one_step.py (2 steps of the bench workload)
# Writes gpurun_out/<tag>_{bench.json,stats,fetch,write,sqA,sqB,sqC,tcc}/…; fold with scratch/pmc_fold.py into profiles/.
# Counters are collected in passes of their own, never together with a trace flag (the pool's gpurun refuses that).
set -e -o pipefail
TAG=$1; SCRIPT=${2:-scratch/one_step.py}
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
python3 -c "import json, os, sys; sys.path.insert(0, '.'); import hash10x_amd; json.dump({'head': os.environ.get('H10X_HEAD'), 'build_id': hash10x_amd.build_id(), 'script': '$SCRIPT'}, open('$O/${TAG}_meta.json', 'w'))"
if [ -n "$ONLY_SQ" ]; then :                      # counters only (no bench, no kernel trace): quick look at one script
elif [ -z "$SKIP_BENCH" ]; then
  if [ -z "$NO_FIRST_BENCH" ]; then                 # (NO_FIRST_BENCH=1: the bench line is taken afterwards, once the folds are in profiles/ — `traffic_stale: false`)
    python3 bench.py --steps 10 --warmup 2 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
    echo "bench done"
  fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o runc -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_rocprof.err
else                                            # another workload than the bench's: trace the script itself
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o runc -- python3 $SCRIPT > $O/${TAG}_trace.log 2> $O/${TAG}_rocprof.err
fi
echo "kernel trace done"
pass() { rocprofv3 --pmc $2 --output-format csv -d $O/${TAG}_$1 -o runc -- python3 $SCRIPT > $O/${TAG}_$1.log 2>&1; echo "pass $1 done"; }
if [ -z "$ONLY_SQ" ]; then
pass fetch "FETCH_SIZE"
pass write "WRITE_SIZE"
fi
pass sqA "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
pass sqB "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"
pass sqC "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_INSTS_FLAT_FLATSEG SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_LDS_ATOMIC_RETURN SQ_THREAD_CYCLES_VALU"
pass sqD "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INST_CYCLES_SALU SQ_INSTS_SMEM_NORM SQ_BUSY_CU_CYCLES"
pass tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
echo "all passes done"
