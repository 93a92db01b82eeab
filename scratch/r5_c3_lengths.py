"""List lengths the cluster kernel sees, by class of the packed translated placement:  python scratch/r5_c3_lengths.py <workload> [lo hi]
A hash of depth d inside the range sits in the good lists of d blocks, each time with a list of d entries: ranks(d) = count(d) * d, entries(d) = ranks(d) * d.
Prints the share of ranks per class (Q <= 16, H <= 32, F <= 64, T 65..96, D 97..128, X beyond), the 64-lane chunks pass A / pass B run with and without class T
(T: three chunks to two lists; as class D: two chunks each) and the share of lanes that hold an entry.
Input: the depth histogram from the GPU library (needs a GPU), or a file of `HASH_COUNT_HIST d count ...` lines as the reference's --hashStats prints them
(H10X_HASHSTATS=<file>: that is how the figures in stage_c.hip's comment were made, on config3-tenth-20M in the build container)."""
import os, re, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import numpy as np


def hist_from_stats(path):
    h = {}
    for line in open(path):
        m = re.match(r"HASH_COUNT_HIST\s+(\d+)\s+(\d+)", line)
        if m:
            h[int(m.group(1))] = int(m.group(2))
    a = np.zeros(max(h) + 1, dtype=np.int64)
    for d, c in h.items():
        a[d] = c
    return a


def hist_from_gpu(name):
    import bench, hash10x_amd
    wl = dict(bench.WORKLOADS[name])
    recs = bench.generate_v2(wl, wl["seed"])[0] if wl.get("gen") == 2 else bench.generate(wl, wl.get("seed", 1))
    dr = hash10x_amd.DeviceRecords(recs); del recs
    h = hash10x_amd.Hash10x(B=wl["B"])
    h.read_fqb_device(dr.ptr, dr.n_records)
    depth = h.export_depth()[1:]
    h.close()
    return np.bincount(depth.astype(np.int64)), wl


name = sys.argv[1]
if os.environ.get("H10X_HASHSTATS"):
    import bench
    cnt, wl = hist_from_stats(os.environ["H10X_HASHSTATS"]), dict(bench.WORKLOADS[name])
else:
    cnt, wl = hist_from_gpu(name)
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (wl["lo"], wl["hi"])
d = np.arange(cnt.size)
inr = (d >= lo) & (d < hi)
ranks = np.where(inr, cnt * d, 0)
entries = ranks * d
classes = [("Q", 1, 16, 0.25), ("H", 17, 32, 0.5), ("F", 33, 64, 1.0), ("T", 65, 96, 1.5), ("D", 97, 128, 2.0)]
tot = ranks.sum()
chunks_t = chunks_d = 0.0
for nm, a, b, ch in classes:
    r = ranks[a:b + 1].sum()
    chunks_t += r * ch
    chunks_d += r * (2.0 if nm == "T" else ch)
    print("class %s (%3d .. %3d entries): %6.2f %% of the ranks, mean length %.1f" % (nm, a, b, 100.0 * r / max(tot, 1), entries[a:b + 1].sum() / max(r, 1)))
rx = ranks[129:].sum()
xch = (ranks[129:] * np.ceil(d[129:] / 64.0)).sum()
print("class X (beyond 128): %.2f %% of the ranks" % (100.0 * rx / max(tot, 1)))
print("%s --hashDepthRange %d %d: %d ranks, %d entries; chunks with class T %.0f (lanes in use %.1f %%), without %.0f (%.1f %%): %.1f %% fewer" %
      (name, lo, hi, tot, entries.sum(), chunks_t + xch, 100.0 * entries.sum() / (64.0 * (chunks_t + xch)), chunks_d + xch, 100.0 * entries.sum() / (64.0 * (chunks_d + xch)),
       100.0 * (1.0 - (chunks_t + xch) / (chunks_d + xch))))
