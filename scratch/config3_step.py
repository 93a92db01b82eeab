"""Two (STEPS) steps of the 1/10 config-3 set for the counter passes of scratch/prof_all.sh; H10X_FIRST_GLOBAL forces the placement."""
import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import bench, hash10x_amd
wl = bench.WORKLOADS["config3-tenth-20M"]
recs = bench.generate(wl, wl["seed"])
d = hash10x_amd.DeviceRecords(recs); del recs
h = hash10x_amd.Hash10x(B=wl["B"])
if os.environ.get("H10X_FIRST_GLOBAL"): h.set_option("cluster_first_global", int(os.environ["H10X_FIRST_GLOBAL"]))
for it in range(int(os.environ.get("STEPS", "2"))):
    h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"]); h.cluster(1, 0, wl["ct"])
print("done", h.sizes(), h.counters()["cluster_first_mode"], h.counters()["cluster_class_counts"])
