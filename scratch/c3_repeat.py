"""config3-tenth: cluster_main of every --cluster call over several sessions of one process (looking for run-to-run spread)."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, hash10x_amd
wl = bench.WORKLOADS["config3-tenth-20M"]
recs = bench.generate(wl, wl.get("seed", 1))
d = hash10x_amd.DeviceRecords(recs)
for inst in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True)
    t0 = time.perf_counter()
    h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"])
    out = []
    for it in range(4):
        a = h.timings()["cluster_main"][0]
        h.cluster(1, 0, wl["ct"])
        out.append(h.timings()["cluster_main"][0] - a)
    c = h.counters()
    print("session %d: cluster_main %s ms  classes %s overflow %s  (%.1f s)" % (inst, " ".join("%.2f" % x for x in out), c["cluster_class_counts"], c.get("cluster_overflow_blocks"), time.perf_counter() - t0), flush=True)
    h.close()
