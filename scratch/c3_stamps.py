"""config3-tenth: phase shares of the cluster launches (option cluster_stamps)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, hash10x_amd
wl = bench.WORKLOADS["config3-tenth-20M"]
recs = bench.generate(wl, wl.get("seed", 1))
d = hash10x_amd.DeviceRecords(recs)
h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True); h.set_option("cluster_stamps", 1)
h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"]); h.cluster(1, 0, wl["ct"])
t = h.counters()["cluster_phase_ticks"]; s = float(sum(t)) or 1.0
print("phase shares: init+bitmap %.3f list %.3f barrier %.3f replay %.3f phase_d %.3f output %.3f" % tuple(x / s for x in t[:6]), flush=True)
print({k: round(v[0], 2) for k, v in h.timings().items() if v[0] > 0.01})
