"""A/B of the first[] width in the cluster launch: option cluster_narrow_first = 1 (2 bytes everywhere), 0 (4 bytes where no wave is lost), w >= 2 (4 bytes down to w list-loop waves)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, hash10x_amd
wl = bench.WORKLOADS[os.environ.get("H10X_WL", "yeast-like-2.5M")]
recs = bench.generate(wl, 1)
d = hash10x_amd.DeviceRecords(recs)
for knob in [int(x) for x in (sys.argv[1:] or ["1", "0", "12", "8", "5", "1"])]:
    h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True)
    h.set_option("cluster_narrow_first", knob)
    h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"])
    best = 1e9
    for it in range(4):
        t0 = h.timings()["cluster_main"][0]
        h.cluster(1, 0, wl["ct"])
        best = min(best, h.timings()["cluster_main"][0] - t0)
    print("narrow_first %2d : cluster_main %.3f ms" % (knob, best), flush=True)
    h.close()
