import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, bench, hash10x_amd
wl = bench.WORKLOADS["yeast-like-2.5M"]
recs = bench.generate(wl, 1)
d = hash10x_amd.DeviceRecords(recs)
h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True); h.set_option("cluster_first_global", 2)
for it in range(2):
    h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"]); h.cluster(1, 0, wl["ct"])
c = h.counters(); t = h.timings()
names = ["init", "first", "mode", "replay", "quotient", "sum+labels", "merge", "x"]
tot = sum(c["cluster_phase_ticks"]) or 1
print({n: round(100.0 * v / tot, 1) for n, v in zip(names, c["cluster_phase_ticks"])})
print("ticks total (100MHz) per WG-sum:", tot, "cluster_kernel ms", t["cluster_kernel"], "cluster ms", t["cluster"])
print({k: c[k] for k in c if k != "cluster_phase_ticks"})
print({k: round(v[0],3) for k,v in t.items() if v[0] > 0})
b = h.export_blocks()
import numpy as np
print("nGood stats via counters: mean good/block", c["sum_good"]/10000, "mean depth", c["sum_good_depth"]/max(c["sum_good"],1), "nSub mean", b["nSubCluster"].mean())
