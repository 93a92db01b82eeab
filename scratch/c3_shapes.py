"""config3-tenth: distribution of the cluster kernel's per-block working set (ranks n, entries in lists, barcodes present is
estimated as in the classification) and how many blocks fit which LDS budget in the ranked placement."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench, hash10x_amd
wl = bench.WORKLOADS[os.environ.get("H10X_WL", "config3-tenth-20M")]
recs = bench.generate(wl, wl.get("seed", 1))
d = hash10x_amd.DeviceRecords(recs)
h = hash10x_amd.Hash10x(B=wl["B"])
h.read_fqb_device(d.ptr, d.n_records)
b = h.export_blocks(); ch = h.export_clushash(); dep = h.export_depth()
good = (dep[ch["hash"]] >= wl["lo"]) & (dep[ch["hash"]] < wl["hi"])
off = np.concatenate([[0], np.cumsum(b["nHash"][1:].astype(np.int64))])
cs = np.concatenate([[0], np.cumsum(good.astype(np.int64))])
n = cs[off[1:]] - cs[off[:-1]]                                   # ranks per block (blocks 1..)
dsum = np.concatenate([[0], np.cumsum(np.where(good, dep[ch["hash"]], 0).astype(np.int64))])
ent = dsum[off[1:]] - dsum[off[:-1]]
nB = len(b)
print("blocks", nB - 1, "n: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %d" % (n.mean(), np.percentile(n, 50), np.percentile(n, 90), np.percentile(n, 99), n.max()))
bm = ((nB + 31) // 32) * 6                                       # bitmap + prefix bytes
est = np.minimum(np.maximum(6 * n, ent // 7), nB)                # first[] entries budgeted (rankedFirstEstimateE)
for waves in (5, 8, 16):
    need = bm + 2 * est + 15 * n + waves * n + 64
    print("hist waves %2d: need KB p50 %.0f p90 %.0f p99 %.0f; fit 80 KB: %.1f %%  96 KB: %.1f %%  104 KB: %.1f %% 160 KB: %.1f %%" % (
        waves, np.percentile(need, 50) / 1024, np.percentile(need, 90) / 1024, np.percentile(need, 99) / 1024,
        100 * (need <= 80 * 1024).mean(), 100 * (need <= 96 * 1024).mean(), 100 * (need <= 104 * 1024).mean(), 100 * (need <= 160 * 1024).mean()))
print("entries per rank: mean %.1f; est first entries per rank: mean %.1f" % ((ent / np.maximum(n, 1)).mean(), (est / np.maximum(n, 1)).mean()))
