import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, hash10x_amd
wl = dict(bench.WORKLOADS["config3-tenth-20M"])
recs = bench.generate(wl, wl["seed"])
d = hash10x_amd.DeviceRecords(recs); hash10x_amd.synchronize(0)
del recs
h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True); h.set_option("cluster_first_global", 3)
for it in range(2):
    h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"]); h.cluster(1, 0, wl["ct"])
c = h.counters()
print({k: c[k] for k in ("sum_good", "sum_good_depth", "cluster_main", "cluster_class_counts", "cluster_overflow_blocks", "cluster_first_mode")})
print({k: round(v[0], 2) for k, v in h.timings().items() if v[0] > 0})
