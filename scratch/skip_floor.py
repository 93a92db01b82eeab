"""What-if timings of the main cluster launch with phases switched off (results wrong, timing only) and the phase stamps."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, hash10x_amd
wl = bench.WORKLOADS["yeast-like-2.5M"]
recs = bench.generate(wl, 1)
d = hash10x_amd.DeviceRecords(recs)
for skip in [int(x) for x in (sys.argv[1:] or ["0", "1", "2", "3", "7", "15", "0"])]:
    h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True)
    h.set_option("cluster_dbg_skip", skip)
    h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"])
    best = 1e9
    for it in range(4):
        t0 = h.timings()["cluster_main"][0]
        h.cluster(1, 0, wl["ct"])
        best = min(best, h.timings()["cluster_main"][0] - t0)
    print("dbg_skip %2d : cluster_main %.3f ms" % (skip, best), flush=True)
    h.close()
h = hash10x_amd.Hash10x(B=wl["B"]); h.set_option("cluster_stamps", 1)
h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"]); h.cluster(1, 0, wl["ct"])
t = h.counters()["cluster_phase_ticks"]; s = float(sum(t)) or 1.0
print("phase shares: init %.3f list %.3f barrier %.3f replay %.3f phase_d %.3f output %.3f" % tuple(x / s for x in t[:6]))
