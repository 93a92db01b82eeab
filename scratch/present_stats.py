"""For a sample of barcodes: E = entries of its good-hash lists, P = distinct other barcodes in them, P2 = those that occur
at least twice (only they can contribute to a count: a barcode met once has first[] == the rank that met it)."""
import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import numpy as np, bench, hash10x_amd
a = [int(x) for x in sys.argv[1:]] + [None] * 5
wl = dict(pairs=(a[0] or 20) * 1000000 // 8, barcodes=(a[1] or 100) * 1000 // 8, genome=(a[2] or 50) * 1000000 // 8, err=(a[4] or 1) / 1000.0, mol=10.0, snp=150, mol_len=50000.0,
          B=a[3] or 24, lo=30, hi=100)
recs = bench.generate(wl, 2)
h = hash10x_amd.Hash10x(B=wl["B"]); h.read_fqb(recs)
b = h.export_blocks(); ch = h.export_clushash(); d = h.export_depth()
nB = b.size
code = np.repeat(np.arange(1, nB), b["nHash"][1:])
ix = ch["hash"].astype(np.int64)
good = (d[ix] >= wl["lo"]) & (d[ix] < wl["hi"])
order = np.argsort(ix, kind="stable"); sx = ix[order]; sc = code[order]
start = np.searchsorted(sx, np.arange(d.size + 1))
off = np.concatenate([[0], np.cumsum(b["nHash"][1:].astype(np.int64))])
rng = np.random.default_rng(1)
rows = []
for c in rng.choice(np.arange(1, nB - 1), 40, replace=False):
    g = ix[off[c - 1]:off[c]][good[off[c - 1]:off[c]]]
    ent = np.concatenate([sc[start[x]:start[x + 1]] for x in g]) if g.size else np.zeros(0, int)
    ent = ent[ent != c]
    u, cnt = np.unique(ent, return_counts=True)
    rows.append((g.size, ent.size, u.size, int((cnt >= 2).sum()), int(cnt[cnt >= 2].sum())))
r = np.array(rows)
print("workload", wl)
print("mean over 40 barcodes: n good %.0f, entries E %.0f, present P %.0f, P2 (>=2 occurrences) %.0f, entries of P2 %.0f" % tuple(r.mean(0)))
print("max: n %d E %d P %d P2 %d" % tuple(r.max(0)[:4]))
