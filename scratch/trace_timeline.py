"""Print the kernel timeline of the last --cluster of a rocprofv3 --kernel-trace csv (start/end in ms relative to the classify kernel)."""
import csv, sys, glob
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/**/*_kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "cluster_classify" in r["Kernel_Name"]]
start = idx[-1]
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    name = r["Kernel_Name"].replace("void h10x::", "").replace("h10x::", "")[:70]
    print("%8.3f %8.3f  %-70s grid %s wg %s lds %s vgpr %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, name,
          r["Grid_Size_X"], r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"]))
