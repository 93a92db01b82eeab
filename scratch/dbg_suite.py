import sys, os, json, tempfile, types
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, orc, hash10x_amd
from driver import run_commands
MAN = json.load(open(orc.GOLDEN + "/manifest.json"))
def fac(**o):
    def mk(k, w, r, B):
        h = hash10x_amd.Hash10x(k=k, w=w, r=r, B=B)
        for n, v in o.items(): h.set_option(n, v)
        return h
    return mk
d = tempfile.mkdtemp()
for case in MAN["digest_cases"]:
    orc.gen_fqb(d + "/" + case["name"] + ".fqb", **case["gen"])
    extra = list(case["args"]); pre = extra[:2] if extra and extra[0] == "-ct" else []
    args = ["-B", case["B"]] + pre + ["--readFQB", case["name"] + ".fqb"] + extra[len(pre):] + ["--writeHash", "out.hash"]
    run_commands(fac(), args, d)
    print(case["name"], orc.sha256(open(d + "/out.hash", "rb").read()) == case["sha256"])
orc.gen_fqb(d + "/x.fqb", 30000, 150, 300000, 0.003, 41, 4.0, 150, 6000)
base = ["-k", 21, "-w", 31, "-r", 17, "-B", 20, "-ct", 3, "--readFQB", "x.fqb", "--hashDepthRange", 4, 30, "--cluster", 1, 0]
run_commands(lambda k, w, r, B: orc.Oracle(k, w, r, B), base + ["--writeHash", "orc.hash"], d)
exp = orc.HashFile(open(d + "/orc.hash", "rb").read())
for trial in range(3):
    h = run_commands(fac(cluster_lds_budget=2048), base + ["--writeHash", "hip.hash"], d)
    got = orc.HashFile(open(d + "/hip.hash", "rb").read())
    bad_ns = np.nonzero(got.blocks["nSubCluster"] != exp.blocks["nSubCluster"])[0]
    bad_pm = np.nonzero(got.blocks["pointToMin"].view(np.uint64) != exp.blocks["pointToMin"].view(np.uint64))[0]
    bad_lab = np.nonzero(got.clushash["subCluster"] != exp.clushash["subCluster"])[0]
    print("trial", trial, "classes", h.counters()["cluster_class_counts"], "bad nSub", len(bad_ns), "bad pointToMin", len(bad_pm), "bad labels", len(bad_lab))
    for c in bad_ns[:4]:
        g = got.block_clushash(c)["subCluster"]; e = exp.block_clushash(c)["subCluster"]
        print("  code", c, "nSub got/exp", got.blocks["nSubCluster"][c], exp.blocks["nSubCluster"][c], "nRead", got.blocks["nRead"][c], "nHash", got.blocks["nHash"][c],
              "labelled got/exp", (g > 0).sum(), (e > 0).sum(), "max label got/exp", g.max(), e.max(), "same labelled set", ((g>0)==(e>0)).all())
