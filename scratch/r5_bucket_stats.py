"""What a one-pass bucket sort of a block's hash indices would meet: per workload, for a sample of blocks, the length of the run of consecutive indices that ends at the
   block's largest index (its new hashes), and the occupancy of NB buckets cut linearly (by a shift) over the rest: mean comparison-loop length (sum s^2 / n), largest bucket.
   python scratch/r5_bucket_stats.py <workload> [<workload> ...]"""
import sys, os, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import numpy as np, bench, hash10x_amd

def stats(keys, NB):
    keys = np.sort(keys.astype(np.int64)); n = keys.size
    kmax = keys[-1]
    # top run
    t = 0
    while t < n and keys[n - 1 - t] == kmax - t: t += 1
    rest = keys[: n - t]
    if rest.size == 0: return n, t, 0.0, 0, 0.0, 0
    lim = int(rest[-1]) + 1
    sh = max(0, lim.bit_length() - NB.bit_length() + 1)
    b = rest >> sh
    cnt = np.bincount(b, minlength=NB)
    lin = (float((cnt.astype(np.float64) ** 2).sum()) / rest.size, int(cnt.max()))
    # log-like: float top bits (8 mantissa bits per octave)
    f = rest.astype(np.float32).view(np.uint32) >> 15
    _, c2 = np.unique(f, return_counts=True)
    lg = (float((c2.astype(np.float64) ** 2).sum()) / rest.size, int(c2.max()))
    return n, t, lin[0], lin[1], lg[0], lg[1]

for name in sys.argv[1:]:
    wl = dict(bench.WORKLOADS[name])
    recs = bench.generate_v2(wl, wl["seed"])[0] if wl.get("gen") == 2 else bench.generate(wl, wl.get("seed", 1))
    dr = hash10x_amd.DeviceRecords(recs); del recs
    h = hash10x_amd.Hash10x(B=wl["B"])
    h.read_fqb_device(dr.ptr, dr.n_records)
    bl = h.export_blocks(); ch = h.export_clushash()
    off = np.zeros(bl.size + 1, dtype=np.int64); off[2:] = np.cumsum(bl["nHash"][1:].astype(np.int64))
    rng = np.random.default_rng(1); pick = np.sort(rng.choice(np.arange(1, bl.size), size=min(600, bl.size - 1), replace=False))
    # also the first 8 blocks (mostly new hashes)
    rows = []
    for b in list(range(1, 9)) + list(pick):
        k = ch["hash"][off[b]: off[b + 1]]
        if k.size < 2: continue
        NB = 4096 if k.size <= 3072 else 8192
        rows.append((b,) + stats(k, NB))
    a = np.array([r[1:] for r in rows], dtype=np.float64)
    print("%s: %d blocks sampled, entries/block mean %.0f max %.0f; top run mean %.0f (%.1f %% of entries)" % (name, len(rows), a[:, 0].mean(), a[:, 0].max(), a[:, 1].mean(), 100 * a[:, 1].sum() / a[:, 0].sum()))
    print("   linear buckets: loop length mean %.2f p90 %.2f max %.2f; largest bucket mean %.1f p99 %.0f max %.0f" % (a[:, 2].mean(), np.percentile(a[:, 2], 90), a[:, 2].max(), a[:, 3].mean(), np.percentile(a[:, 3], 99), a[:, 3].max()))
    print("   float buckets : loop length mean %.2f p90 %.2f max %.2f; largest bucket mean %.1f p99 %.0f max %.0f" % (a[:, 4].mean(), np.percentile(a[:, 4], 90), a[:, 4].max(), a[:, 5].mean(), np.percentile(a[:, 5], 99), a[:, 5].max()))
    for r in rows[:8]: print("   block %d: n %d run %d lin %.2f/%d float %.2f/%d" % r)
    sys.stdout.flush()
    h.close(); dr.free()
