"""Depth histogram of the 1/10-scale proxy of the genome3g workload (same coverage: 36 M pairs, 200 k barcodes, 300 Mb x 2, e = 0.05 %): picks --hashDepthRange."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, bench, hash10x_amd
wl = dict(pairs=36000000, barcodes=200000, genome=300000000, err=0.0005, mol=10.0, snp=150, mol_len=50000.0)
t = time.time(); recs, _f, _t = bench.generate_v2(wl, 3); print("generated in %.1f s" % (time.time() - t), flush=True)
d = hash10x_amd.DeviceRecords(recs); del recs
h = hash10x_amd.Hash10x(B=27); h.enable_timing(True)
h.read_fqb_device(d.ptr, d.n_records)
dep = h.export_depth()
print("hashNumber", h.sizes()["hashNumber"], "entries", h.counters()["entries"])
hist = np.bincount(np.minimum(dep[1:], 200))
for lo in range(0, 80, 4): print(lo, hist[lo:lo + 4].tolist())
print("sum d^2 for ranges:", {(a, b): int((dep[(dep >= a) & (dep < b)].astype(np.int64) ** 2).sum()) for a, b in ((8, 40), (8, 50), (10, 50), (12, 60), (30, 100))})
for (a, b) in ((8, 50), (10, 50)):
    h.depth_range(a, b); t = time.time(); h.cluster(1, 0, 5); hash10x_amd.synchronize(0)
    c = h.counters(); print((a, b), "cluster %.1f ms" % (1e3 * (time.time() - t)), "sum_good", c["sum_good"], "sum_good_depth", c["sum_good_depth"], "mode", c["cluster_first_mode"], c["cluster_class_counts"], flush=True)
    b_ = h.export_blocks(); print("   nSubCluster mean %.2f max %d" % (b_["nSubCluster"][1:].mean(), b_["nSubCluster"].max()))
    h.read_fqb_device(d.ptr, d.n_records)
