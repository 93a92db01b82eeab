"""Phase stamps of the cluster kernel (wall_clock64 ticks, summed over workgroups) under each first[] placement."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, hash10x_amd
wl = bench.WORKLOADS["yeast-like-2.5M"]
recs = bench.generate(wl, 1)
d = hash10x_amd.DeviceRecords(recs)
names = ["init(+bitmap)", "lists(a+b)", "sync", "replay", "quotient(d)", "out", "x", "y"]
for mode in (0, 2, 3):
    h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True); h.set_option("cluster_stamps", 1); h.set_option("cluster_first_global", mode)
    h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"]); h.cluster(1, 0, wl["ct"])
    c = h.counters(); t = h.timings()
    tk = c["cluster_phase_ticks"]
    print("knob", mode, "classes", c["cluster_class_counts"], "cluster ms %.2f" % t["cluster"][0], {n: round(v / 1e5 / 512, 3) for n, v in zip(names, tk)}, "(ms per resident workgroup if spread over 512)", flush=True)
    h.close()
