import sys, os, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import numpy as np, bench, hash10x_amd
wl = bench.WORKLOADS["yeast-like-2.5M"]
recs = bench.generate(wl, 1)
d = hash10x_amd.DeviceRecords(recs)
h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True)
for it in range(4):
    t0 = time.perf_counter(); h.read_fqb_device(d.ptr, d.n_records); t1 = time.perf_counter()
    h.depth_range(wl["lo"], wl["hi"]); t2 = time.perf_counter(); h.cluster(1, 0, wl["ct"]); t3 = time.perf_counter()
    t = h.timings()
    dev_read = sum(t[k][0] for k in ("block_runs","mosh_extract","mosh_fallback","compact_entries","sort_by_hash","index_rank","probe_table","clushash_build"))
    print("it %d: read_fqb wall %.2f ms (device %.2f) | depth_range wall %.2f (device %.2f) | cluster wall %.2f (device %.2f) | total %.2f" % (
        it, 1e3*(t1-t0), dev_read, 1e3*(t2-t1), t["good_hashes"][0], 1e3*(t3-t2), t["cluster"][0], 1e3*(t3-t0)))
