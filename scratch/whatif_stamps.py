import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, hash10x_amd
wl = bench.WORKLOADS["yeast-like-2.5M"]
recs = bench.generate(wl, 1)
d = hash10x_amd.DeviceRecords(recs)
names = ["init", "first", "mode", "replay", "quotient", "out", "-", "x"]
for skip in (0, 15):
    h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True); h.set_option("cluster_dbg_skip", skip); h.set_option("cluster_stamps", 1)
    h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"])
    for it in range(2): h.cluster(1, 0, wl["ct"])
    c = h.counters(); tot = sum(c["cluster_phase_ticks"]) or 1
    print("skip", skip, {n: round(v / 1e8 * 1e3, 1) for n, v in zip(names, c["cluster_phase_ticks"])}, "WG-ms total %.1f" % (tot / 1e8 * 1e3), flush=True)
    h.close()
