#!/bin/bash
# kernel trace of the full-size config-3 run (scratch/full_step.py <passes>): durations of everything --cluster launches
export TMPDIR=/tmp
TAG=$1; shift
rm -rf gpurun_out/${TAG}_trace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -o runc -- python3 scratch/full_step.py "$@" > gpurun_out/${TAG}_trace.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/${TAG}_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
last = max(i for i, r in enumerate(rows) if "cluster_classify" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    print("%-62s start %9.3f ms  dur %9.3f ms" % (r["Kernel_Name"].split("(")[0][-62:], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
