import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, hash10x_amd
wl = bench.WORKLOADS["yeast-like-2.5M"]
recs = bench.generate(wl, 1)
d = hash10x_amd.DeviceRecords(recs); hash10x_amd.synchronize(0)
for thr, bud in [(1024, 0), (512, 53 * 1024), (512, 40 * 1024 - 512), (512, 79 * 1024), (1024, 0)]:
    h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True)
    h.set_option("cluster_threads0", thr); h.set_option("cluster_budget0", bud)
    h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"])
    best = 1e9; bk = 1e9
    for it in range(4):
        t0 = h.timings(); h.cluster(1, 0, wl["ct"]); t1 = h.timings()
        best = min(best, t1["cluster_main"][0] - t0["cluster_main"][0]); bk = min(bk, t1["cluster_kernel"][0] - t0["cluster_kernel"][0])
    print("threads0 %4d budget0 %6d : cluster_main %.3f ms cluster_kernel %.3f classes %s" % (thr, bud, best, bk, h.counters()["cluster_class_counts"]), flush=True)
    h.close()
