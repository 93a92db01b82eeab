"""BASELINE configs[2] stand-in on ONE GPU: synthetic 500 Mb diploid, 200 M read pairs, 1 M barcodes, -B 28 (SURVEY §8d table,
e = 0.1 % keeps the distinct hashes under the 2^(B-2) - 2 cap). Prints per-command wall time, per-kernel device time and
the clustering roofline figure; the size-independent checks are re-run determinism (two passes give the same digest of
blocks + clusHash) and the invariants of the .hash state (clusHash sorted by index inside every block, depth = number of
barcodes per hash). usage: config3.py [pairs_millions [barcodes_thousands [genome_mb [B]]]]"""
import sys, os, time, hashlib
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, bench, hash10x_amd
a = [int(x) for x in sys.argv[1:]] + [None] * 4
wl = dict(pairs=(a[0] or 200) * 1000000, barcodes=(a[1] or 1000) * 1000, genome=(a[2] or 500) * 1000000, err=0.001, mol=10.0, snp=150, mol_len=50000.0,
          B=a[3] or 28, lo=30, hi=100, ct=5)
import threading
_stop = threading.Event()
def _beat():                                                  # gpurun kills a run that stays silent for 7 minutes
    t0 = time.time()
    while not _stop.wait(45): print("  ... %.0f s" % (time.time() - t0), flush=True)
threading.Thread(target=_beat, daemon=True).start()
t = time.time(); recs = bench.generate(wl, 2); print("generated %d pairs in %.1fs" % (recs.size // 30, time.time() - t), flush=True)
h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True)
if os.environ.get("H10X_FIRST_GLOBAL"): h.set_option("cluster_first_global", int(os.environ["H10X_FIRST_GLOBAL"]))   # placement override (2 ranked, 3 hashed)
t = time.time(); dr = hash10x_amd.DeviceRecords(recs); print("uploaded %.1f GB in %.1fs" % (recs.nbytes / 1e9, time.time() - t), flush=True)
del recs
digests = []
for it in range(2):
    t0 = time.time(); h.read_fqb_device(dr.ptr, dr.n_records); t1 = time.time()
    h.depth_range(wl["lo"], wl["hi"]); t2 = time.time(); h.cluster(1, 0, wl["ct"]); t3 = time.time()
    print("pass %d wall s: readFQB %.3f hashDepthRange %.3f cluster %.3f  -> %.1f M read pairs/s, %.0f barcodes/s clustered" %
          (it, t1 - t0, t2 - t1, t3 - t2, wl["pairs"] / (t3 - t0) / 1e6, wl["barcodes"] / (t3 - t2)), flush=True)
    b = h.export_blocks(); ch = h.export_clushash()
    digests.append(hashlib.sha256(b.tobytes() + ch.tobytes()).hexdigest())
c = h.counters(); tm = h.timings(); z = h.sizes()
print("sizes", z)
print("counters", {k: c[k] for k in ("entries", "distinct", "sum_good", "sum_good_depth", "cluster_first_mode", "cluster_class_counts", "cluster_overflow_blocks", "fallback_blocks")})
print("device ms (last pass)", {k: round(v[0], 2) for k, v in tm.items() if v[0] > 0})
alg = 4.0 * c["sum_good_depth"] + 14.0 * c["sum_good"] + 16.0 * c["sum_hash_clustered"]
print("cluster_kernel: %.2f GB algorithmic in %.1f ms -> %.0f GB/s = %.1f %% of 8 TB/s" % (alg / 1e9, tm["cluster_kernel"][0], alg / (tm["cluster_kernel"][0] * 1e-3) / 1e9,
      alg / (tm["cluster_kernel"][0] * 1e-3) / 8e12 * 100))
_stop.set()
print("determinism: two passes", "identical" if digests[0] == digests[1] else "DIFFER", digests[0][:16])
# invariants of the state (size-independent properties)
off = np.concatenate([[0], np.cumsum(b["nHash"][1:].astype(np.int64))])
ix = ch["hash"].astype(np.int64)
brk = np.zeros(ix.size, bool); brk[off[:-1][off[:-1] < ix.size]] = True
ok_sorted = bool(np.all((np.diff(ix) > 0) | brk[1:]))
depth = np.bincount(ix, minlength=z["hashNumber"])
dd = h.export_depth()
print("invariants: clusHash strictly ascending inside every block:", ok_sorted, "| depth == barcodes per hash:", bool(np.array_equal(depth[1:], dd[1:z["hashNumber"]])),
      "| labels <= nSubCluster:", bool(np.all(ch["subCluster"] <= np.repeat(b["nSubCluster"][1:], b["nHash"][1:]))))
