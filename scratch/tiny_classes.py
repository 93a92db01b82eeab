import sys, os, tempfile
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, hash10x_amd, orc
d = tempfile.mkdtemp()
recs = orc.gen_fqb(d + "/x.fqb", 60, 15, 40000, 0.001, 1507, 4.0, 150, 2500)
for lo, hi, ct in ((3, 4, 5), (2, 4, 1)):
    h = hash10x_amd.Hash10x(k=24, w=32, r=9, B=21); h.set_option("cluster_first_global", 1)
    h.read_fqb(np.ascontiguousarray(recs).reshape(-1), 0, 100000); h.depth_range(lo, hi); h.cluster(1, 0, ct)
    c = h.counters(); print(lo, hi, "classes", c["cluster_class_counts"], "first_mode", c["cluster_first_mode"], "sum_good", c["sum_good"])
