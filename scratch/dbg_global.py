import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, orc, hash10x_amd, tempfile
d = tempfile.mkdtemp()
recs = orc.gen_fqb(d + "/x.fqb", 30000, 150, 300000, 0.003, 41, 4.0, 150, 6000)
o = orc.Oracle(B=20); o.read_fqb(recs); o.depth_range(4, 30); o.cluster(1, 0, 3); o.write_hash(d + "/o.hash")
exp = orc.HashFile(open(d + "/o.hash", "rb").read())
# warm the process like the digest tests do
for gen, rng, ct in ((dict(pairs=60000, barcodes=200, genome=400000, err=0.002, seed=5, mol=4.0), (8, 40), 5),
                     (dict(pairs=36000, barcodes=3, genome=3000000, err=0.01, seed=3, mol=12.0), (1, 3), 5)):
    r2 = orc.gen_fqb(d + "/w.fqb", **gen)
    hw = hash10x_amd.Hash10x(B=20); hw.read_fqb(r2); hw.depth_range(*rng); hw.cluster(1, 0, ct); print("warm", hw.counters()["cluster_class_counts"], hw.sizes())
    del hw
for budget in (2048, 2048, 0):
    h = hash10x_amd.Hash10x(B=20)
    if budget: h.set_option("cluster_lds_budget", budget)
    h.read_fqb(recs); h.depth_range(4, 30); h.cluster(1, 0, 3); h.write_hash(d + "/g.hash")
    got = orc.HashFile(open(d + "/g.hash", "rb").read())
    bad_ns = np.nonzero(got.blocks["nSubCluster"] != exp.blocks["nSubCluster"])[0]
    bad_pm = np.nonzero(got.blocks["pointToMin"].view(np.uint64) != exp.blocks["pointToMin"].view(np.uint64))[0]
    bad_lab = np.nonzero(got.clushash["subCluster"] != exp.clushash["subCluster"])[0]
    print("budget", budget, "classes", h.counters()["cluster_class_counts"], "bad nSub", len(bad_ns), "bad pointToMin", len(bad_pm), "bad labels", len(bad_lab))
    for c in bad_ns[:3]:
        g = got.block_clushash(c)["subCluster"]; e = exp.block_clushash(c)["subCluster"]
        print("  code", c, "nSub got/exp", got.blocks["nSubCluster"][c], exp.blocks["nSubCluster"][c], "nRead", got.blocks["nRead"][c], "nHash", got.blocks["nHash"][c],
              "labelled got/exp", (g > 0).sum(), (e > 0).sum(), "max label got/exp", g.max(), e.max())
