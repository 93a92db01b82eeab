import sys, os, time, threading, tempfile, hashlib
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, bench, hash10x_amd, orc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
wl = dict(bench.WORKLOADS["yeast-like-2.5M"])
wl["pairs"] *= N; wl["barcodes"] *= N; wl["genome"] *= N; wl["B"] += (N - 1).bit_length()
t = time.time(); recs = bench.generate(wl, 1); print("generated", recs.size // 30, "pairs in %.1fs" % (time.time() - t), flush=True)
cut = hash10x_amd.partition(recs, N)
comms = hash10x_amd.Comm.local(N)
d = tempfile.mkdtemp()
errs = [None] * N; times = [None] * N; ctrs = [None] * N
def work(r):
    try:
        h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True)
        shard = recs[30 * cut[r]: 30 * cut[r + 1]]
        dr = hash10x_amd.DeviceRecords(shard)
        for it in range(2):
            t0 = time.time(); h.shard_read_fqb_device(comms[r], dr.ptr, dr.n_records); t1 = time.time()
            h.depth_range(wl["lo"], wl["hi"]); t2 = time.time(); h.cluster(1, 0, wl["ct"]); t3 = time.time()
        times[r] = (t1 - t0, t2 - t1, t3 - t2); ctrs[r] = (h.counters(), {k: round(v[0], 2) for k, v in h.timings().items() if v[0] > 0})
        h.shard_gather()
        if r == 0: h.write_hash(d + "/hip.hash")
        h.close()
    except Exception as e:
        errs[r] = e
th = [threading.Thread(target=work, args=(r,)) for r in range(N)]
[x.start() for x in th]; [x.join() for x in th]
for e in errs:
    if e: raise e
for r in range(N): print("rank", r, "wall s read/range/cluster", ["%.3f" % x for x in times[r]], "first_mode", ctrs[r][0]["cluster_first_mode"], "classes", ctrs[r][0]["cluster_class_counts"], "overflow", ctrs[r][0]["cluster_overflow_blocks"], ctrs[r][1])
got = open(d + "/hip.hash", "rb").read(); print("gathered .hash bytes", len(got), flush=True)
if orc.have_ref():
    recs.tofile(d + "/x.fqb")
    t = time.time()
    r = orc.run_ref(["-B", wl["B"], "--readFQB", "x.fqb", "--hashDepthRange", wl["lo"], wl["hi"], "--cluster", 1, 0, "--writeHash", "ref.hash"], d, timeout=3000)
    print("reference took %.1fs rc %d" % (time.time() - t, r.returncode))
    exp = orc.canonical_hash_bytes(open(d + "/ref.hash", "rb").read())
    print("PARITY vs reference binary:", "identical" if exp == got else orc.describe_diff(got, exp))
    if exp != got:
        G, E = orc.HashFile(got), orc.HashFile(exp)
        bad = np.nonzero(G.blocks["nSubCluster"] != E.blocks["nSubCluster"])[0]
        print("bad nSub codes", bad.tolist()[:10], "labels 255 in reference:", int((E.clushash["subCluster"] == 255).sum()))
