import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, hash10x_amd
wl = bench.WORKLOADS["yeast-like-2.5M"]
recs = bench.generate(wl, 1)
d = hash10x_amd.DeviceRecords(recs); hash10x_amd.synchronize(0)
h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True)
for it in range(3):
    h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"]); h.cluster(1, 0, wl["ct"])
    c = h.counters(); print({k: c[k] for k in ("sum_good", "sum_good_depth", "sum_hash_clustered", "clustered_codes", "cluster_main", "cluster_class_counts", "entries")})
print({k: round(v[0] / 3, 3) for k, v in h.timings().items() if v[0] > 0})
