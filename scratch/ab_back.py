import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, hash10x_amd
wl = bench.WORKLOADS["yeast-like-2.5M"]
recs = bench.generate(wl, 1)
d = hash10x_amd.DeviceRecords(recs)
for rep in range(3):
    for name, big in (("back list", 0), ("no back list", 1936)):
        h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True); h.set_option("cluster_big_ranks", big)
        h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"])
        for it in range(5): h.cluster(1, 0, wl["ct"])
        t = h.timings(); print("%-13s cluster %.3f ms  main launch %.3f ms  classes %s" % (name, t["cluster"][0] / 5, t["cluster_main"][0] / 5, h.counters()["cluster_class_counts"]), flush=True)
        h.close()
