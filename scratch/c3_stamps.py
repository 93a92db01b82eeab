"""config3-tenth: phase shares of the cluster launches (option cluster_stamps); argv[1] = cluster_first_global (0 auto, 2 ranked, 3 hashed)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, hash10x_amd
wl = bench.WORKLOADS["config3-tenth-20M"]
recs = bench.generate(wl, wl.get("seed", 1))
d = hash10x_amd.DeviceRecords(recs)
for mode in [int(x) for x in (sys.argv[1:] or ["0"])]:
    h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True); h.set_option("cluster_stamps", 1)
    if mode: h.set_option("cluster_first_global", mode)
    h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"]); h.cluster(1, 0, wl["ct"])
    c = h.counters(); t = c["cluster_phase_ticks"]; s = float(sum(t)) or 1.0
    print("mode %d: phase shares: init+bitmap %.3f list %.3f barrier %.3f replay %.3f phase_d %.3f output %.3f" % ((mode,) + tuple(x / s for x in t[:6])), flush=True)
    print("   cluster %.2f ms, classes %s overflow %s" % (h.timings()["cluster"][0], c["cluster_class_counts"], c["cluster_overflow_blocks"]), flush=True)
    h.close()
