"""What-if timings of the k-mer hashing launch (build with EXTRA=-DH10X_DBG_SKIP; results are wrong, timing only):
16 no set inserts, 32 no hashing (no survivors either), 64 no compaction of the table."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, hash10x_amd
wl = bench.WORKLOADS["yeast-like-2.5M"]
recs = bench.generate(wl, 1)
d = hash10x_amd.DeviceRecords(recs)
for skip in [int(x) for x in (sys.argv[1:] or ["0", "16", "32", "48", "64", "112", "0"])]:
    h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True)
    h.set_option("cluster_dbg_skip", skip)
    best = 1e9
    for it in range(4):
        try:
            h.read_fqb_device(d.ptr, d.n_records)
        except Exception as e:                                # noqa: BLE001   wrong results may trip a later check
            pass
        best = min(best, h.timings()["mosh_extract"][0])
    print("dbg %3d : mosh_extract %.3f ms" % (skip, best), flush=True)
    h.close()
