"""Phase breakdown of cluster_kernel on the config-3-proportioned 20 M-pair set for a forced first[] placement (argv[1]: 0 auto, 2 ranked, 3 hashed, 1 hybrid)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, bench, hash10x_amd
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
wl = dict(pairs=20000000, barcodes=100000, genome=50000000, err=0.001, mol=10.0, snp=150, mol_len=50000.0, B=26, lo=30, hi=100, ct=5)
recs = bench.generate(wl, 2)
d = hash10x_amd.DeviceRecords(recs)
h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True); h.set_option("cluster_stamps", 1)
if mode: h.set_option("cluster_first_global", mode)
h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"])
for it in range(2): h.cluster(1, 0, wl["ct"])
c = h.counters(); t = h.timings()
names = ["init", "first", "mode", "replay", "quotient", "out", "-", "x"]
tot = sum(c["cluster_phase_ticks"]) or 1
print("mode", mode, {n: round(100.0 * v / tot, 1) for n, v in zip(names, c["cluster_phase_ticks"])}, "WG-seconds %.2f" % (tot / 1e8))
print("cluster_kernel ms", round(t["cluster_kernel"][0] / t["cluster_kernel"][1], 2), "classes", c["cluster_class_counts"], "overflow", c["cluster_overflow_blocks"])
