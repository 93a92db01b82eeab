"""What-if timings of the clusHash build (build with EXTRA=-DH10X_DBG_SKIP; results wrong): 1024 no table look-up, 2048 no block sort."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, hash10x_amd
wl = bench.WORKLOADS["yeast-like-2.5M"]
recs = bench.generate(wl, 1)
d = hash10x_amd.DeviceRecords(recs)
for skip in [int(x) for x in (sys.argv[1:] or ["0", "1024", "2048", "3072", "0"])]:
    h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True)
    h.set_option("cluster_dbg_skip", skip)
    best = 1e9
    for it in range(4):
        h.read_fqb_device(d.ptr, d.n_records)
        best = min(best, h.timings()["clushash_build"][0])
    print("dbg %4d : clushash_build %.3f ms" % (skip, best), flush=True)
    h.close()
