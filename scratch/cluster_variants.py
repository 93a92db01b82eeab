"""Time --cluster on the bench workload under the class-0 tuning knobs (threads, LDS budget)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, hash10x_amd
wl = bench.WORKLOADS["yeast-like-2.5M"]
recs = bench.generate(wl, 1)
d = hash10x_amd.DeviceRecords(recs)
variants = [(1024, 79 * 1024), (1024, 159 * 1024), (512, 79 * 1024), (512, 52 * 1024), (512, 39 * 1024), (1024, 105 * 1024)]
if len(sys.argv) > 1:
    variants = [tuple(int(x) for x in v.split(":")) for v in sys.argv[1:]]
for thr, bud in variants:
    h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True)
    h.set_option("cluster_threads0", thr); h.set_option("cluster_budget0", bud)
    h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"])
    best = 1e9
    for it in range(4):
        h.reset_timings() if hasattr(h, "reset_timings") else None
        t0 = h.timings()["cluster_kernel"][0]
        h.cluster(1, 0, wl["ct"])
        best = min(best, h.timings()["cluster_kernel"][0] - t0)
    c = h.counters()
    print("threads0 %4d budget0 %6d : cluster_kernel %.3f ms  classes %s" % (thr, bud, best, c["cluster_class_counts"]), flush=True)
    del h
