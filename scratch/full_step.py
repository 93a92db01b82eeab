"""BASELINE configs[2] at its own size on ONE GPU (gen_fqb v2 set: 200 M pairs, 1 M barcodes, 500 Mb x 2, -B 29 — the set whose reference digest is in
tests/golden/manifest.json "full_digest_cases"): per-command wall time, per-kernel device time, cluster classes, phase shares.
   python scratch/full_step.py [passes] [stamps]          env H10X_FIRST_GLOBAL = 2 ranked / 3 hashed placement override"""
import sys, os, time, json, threading
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import numpy as np, bench, hash10x_amd
man = json.load(open(os.path.join(R, "tests", "golden", "manifest.json")))["full_digest_cases"][0]
g = man["gen2"]
wl = dict(pairs=g["pairs"], barcodes=g["barcodes"], genome=g["genome"], err=g["err"], mol=g["mol"], snp=g["snp"], mol_len=g["mol_len"])
LO, HI = 30, 100
if os.environ.get("H10X_WL"):                                  # another generator-v2 workload of bench.py, e.g. genome3g-300M
    w = bench.WORKLOADS[os.environ["H10X_WL"]]
    wl = {k: w[k] for k in ("pairs", "barcodes", "genome", "err", "mol", "snp", "mol_len")}; g = dict(g, seed=w.get("seed", 1)); man = dict(man, B=w["B"], hash_number=0, blocks_max=0, sum_nHash=0)
    LO, HI = w["lo"], w["hi"]
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
stamps = len(sys.argv) > 2
_stop = threading.Event()
def _beat():
    t0 = time.time()
    while not _stop.wait(45): print("  ... %.0f s" % (time.time() - t0), flush=True)
threading.Thread(target=_beat, daemon=True).start()
t = time.time(); recs, _f, _t = bench.generate_v2(wl, g["seed"]); print("generated %d pairs in %.1f s (%d threads)" % (recs.size // 30, time.time() - t, os.cpu_count()), flush=True)
t = time.time(); dr = hash10x_amd.DeviceRecords(recs); print("uploaded %.1f GB in %.1f s" % (recs.nbytes / 1e9, time.time() - t), flush=True)
del recs
h = hash10x_amd.Hash10x(B=man["B"]); h.enable_timing(True)
if os.environ.get("H10X_FIRST_GLOBAL"): h.set_option("cluster_first_global", int(os.environ["H10X_FIRST_GLOBAL"]))
if stamps: h.set_option("cluster_stamps", 1)
for kv in filter(None, os.environ.get("H10X_OPTS", "").split(",")):
    k, v = kv.split("="); h.set_option(k, int(v))
for it in range(passes):
    hash10x_amd.synchronize(0)
    t0 = time.time(); h.read_fqb_device(dr.ptr, dr.n_records); hash10x_amd.synchronize(0); t1 = time.time()
    h.depth_range(LO, HI); hash10x_amd.synchronize(0); t2 = time.time(); h.cluster(1, 0, 5); hash10x_amd.synchronize(0); t3 = time.time()
    print("pass %d wall s: readFQB %.3f hashDepthRange %.3f cluster %.3f -> %.1f M read pairs/s" % (it, t1 - t0, t2 - t1, t3 - t2, wl["pairs"] / (t3 - t0) / 1e6), flush=True)
    tm = h.timings(); c = h.counters()
    print("   device ms", {k: round(v[0], 1) for k, v in tm.items() if v[0] > 0}, flush=True)
    alg = 4.0 * c["sum_good_depth"] + 14.0 * c["sum_good"] + 16.0 * c["sum_hash_clustered"]
    print("   cluster_kernel %.1f ms: %.1f GB algorithmic = %.0f GB/s = %.2f %% of 8 TB/s; placement %s classes %s overflow %s" % (
        tm["cluster_kernel"][0], alg / 1e9, alg / tm["cluster_kernel"][0] / 1e6, alg / tm["cluster_kernel"][0] / 1e6 / 80, c["cluster_first_mode"], c["cluster_class_counts"], c["cluster_overflow_blocks"]), flush=True)
    if stamps:
        tk = c["cluster_phase_ticks"]; s = float(sum(tk)) or 1.0
        print("   phase shares: init %.3f list %.3f barrier %.3f settle %.3f regather %.3f output %.3f" % tuple(x / s for x in tk[:6]), flush=True)
z = h.sizes()
print("sizes", z, "expected", {k: man[k] for k in ("hash_number", "blocks_max", "sum_nHash")})
_stop.set()
