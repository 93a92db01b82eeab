import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, hash10x_amd
wl = bench.WORKLOADS["yeast-like-2.5M"]
recs = bench.generate(wl, 1)
d = hash10x_amd.DeviceRecords(recs)
for big in [int(x) for x in sys.argv[1:]] or [100000, 0, 3000, 2500, 2000, 1700, 1500]:
    h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True); h.set_option("cluster_big_ranks", big)
    h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"])
    best = 1e9; bk = 1e9
    for it in range(4):
        t = h.timings(); t0 = t["cluster_main"][0]; k0 = t["cluster_kernel"][0]; h.cluster(1, 0, wl["ct"]); t = h.timings()
        best = min(best, t["cluster_main"][0] - t0); bk = min(bk, t["cluster_kernel"][0] - k0)
    print("big_ranks %6d: cluster_main %.3f ms, all cluster kernels %.3f ms, classes %s" % (big, best, bk, h.counters()["cluster_class_counts"]), flush=True)
    h.close()
