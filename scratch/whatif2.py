import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, hash10x_amd
wl = bench.WORKLOADS["yeast-like-2.5M"]
recs = bench.generate(wl, 1)
d = hash10x_amd.DeviceRecords(recs)
for skip in [int(x) for x in sys.argv[1:]] or [0, 16, 15, 31]:
    h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True); h.set_option("cluster_dbg_skip", skip)
    h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"])
    best = 1e9
    for it in range(3):
        t0 = h.timings()["cluster_main"][0]; h.cluster(1, 0, wl["ct"]); best = min(best, h.timings()["cluster_main"][0] - t0)
    print("skip mask %2d (1 update, 2 mode, 4 phase d, 8 barrier, 16 no warming touches): cluster_main %.3f ms" % (skip, best), flush=True)
    h.close()
