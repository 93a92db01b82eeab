"""N x the yeast-scale set on ONE GPU (single context, no sharding): timings per command and per kernel, the cluster
classes / first[] placement that were used, and — when REF=1 and oracle/_ref exists — byte parity of the .hash with the
reference binary. usage: scale_1gpu.py N [REF]"""
import sys, os, time, tempfile
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, bench, hash10x_amd, orc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
REF = len(sys.argv) > 2 and sys.argv[2] == "1"
wl = dict(bench.WORKLOADS["yeast-like-2.5M"])
wl["pairs"] *= N; wl["barcodes"] *= N; wl["genome"] *= N; wl["B"] = min(30, wl["B"] + (N - 1).bit_length())
t = time.time(); recs = bench.generate(wl, 1); print("generated", recs.size // 30, "pairs in %.1fs" % (time.time() - t), flush=True)
d = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True)
dr = hash10x_amd.DeviceRecords(recs)
for it in range(2):
    t0 = time.time(); h.read_fqb_device(dr.ptr, dr.n_records); t1 = time.time()
    h.depth_range(wl["lo"], wl["hi"]); t2 = time.time(); h.cluster(1, 0, wl["ct"]); t3 = time.time()
    print("pass %d wall s: readFQB %.3f hashDepthRange %.3f cluster %.3f  -> %.1f M read pairs/s" % (it, t1 - t0, t2 - t1, t3 - t2, wl["pairs"] / (t3 - t0) / 1e6), flush=True)
c = h.counters(); tm = h.timings()
print("sizes", h.sizes())
print("counters", {k: c[k] for k in ("entries", "distinct", "sum_good", "sum_good_depth", "cluster_first_mode", "cluster_class_counts", "cluster_overflow_blocks", "fallback_blocks")})
print("device ms (both passes)", {k: round(v[0], 2) for k, v in tm.items() if v[0] > 0})
alg = 4.0 * c["sum_good_depth"] + 14.0 * c["sum_good"] + 16.0 * c["sum_hash_clustered"]
print("cluster_kernel: %.3f GB algorithmic, %.2f ms per launch -> %.0f GB/s" % (alg / 1e9, tm["cluster_kernel"][0] / 2, alg / (tm["cluster_kernel"][0] / 2 * 1e-3) / 1e9))
if REF and orc.have_ref():
    h.write_hash(d + "/hip.hash"); got = open(d + "/hip.hash", "rb").read()
    recs.tofile(d + "/x.fqb")
    t = time.time()
    r = orc.run_ref(["-B", wl["B"], "--readFQB", "x.fqb", "--hashDepthRange", wl["lo"], wl["hi"], "--cluster", 1, 0, "--writeHash", "ref.hash"], d, timeout=3000)
    print("reference took %.1fs rc %d" % (time.time() - t, r.returncode), flush=True)
    exp = orc.canonical_hash_bytes(open(d + "/ref.hash", "rb").read())
    print("PARITY vs reference binary:", "identical" if exp == got else orc.describe_diff(got, exp))
    for f in ("x.fqb", "hip.hash", "ref.hash"): os.remove(d + "/" + f)
