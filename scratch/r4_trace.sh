#!/bin/bash
# kernel trace of scratch/r4_whatif.py <args>: per-kernel durations and the start/end of the two cluster launches
export TMPDIR=/tmp
TAG=$1; shift
rm -rf gpurun_out/${TAG}_trace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -o runc -- python3 scratch/r4_whatif.py "$@" > gpurun_out/${TAG}_trace.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/${TAG}_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "cluster_kernel" in r["Kernel_Name"] or "replay" in r["Kernel_Name"] or "read_merge" in r["Kernel_Name"] or "point_sum" in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows[-14:]:
    print("%-60s start %9.3f ms  dur %8.3f ms" % (r["Kernel_Name"].split("(")[0][-60:], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
