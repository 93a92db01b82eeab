#!/bin/bash
# instruction counts of the mode-4 cluster kernels on the 1/10 config-3 set for several cluster_dbg_skip values: scratch/r4_inst.sh <tag> <skip> ...
export TMPDIR=/tmp
TAG=$1; shift
for SK in "$@"; do
  rm -rf gpurun_out/${TAG}_i$SK
  H10X_FIRST_GLOBAL=4 H10X_DBG_SKIP_OPT=$SK STEPS=1 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${TAG}_i$SK -o runc -- python3 scratch/r4_c3step.py > gpurun_out/${TAG}_i$SK.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/${TAG}_i$SK/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if "cluster_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
ch = 137e6
print("skip $SK: per 64-entry chunk: VALU %.1f SALU %.1f branch %.1f LDS %.1f VMEM rd %.2f wr %.2f; GUI_ACTIVE/8 %.1f Mcycles" % (acc["SQ_INSTS_VALU"]/ch, acc["SQ_INSTS_SALU"]/ch, acc["SQ_INSTS_BRANCH"]/ch, acc["SQ_INSTS_LDS"]/ch, acc["SQ_INSTS_VMEM_RD"]/ch, acc["SQ_INSTS_VMEM_WR"]/ch, acc["GRBM_GUI_ACTIVE"]/8e6))
PY
done
