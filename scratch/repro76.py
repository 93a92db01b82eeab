import os, sys, tempfile
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import orc, hash10x_amd
from soak import run_commands
d = tempfile.mkdtemp()
k, w, r, B = 31, 5, 17, 23
orc.gen_fqb(os.path.join(d, "x.fqb"), 60, 15, 100000, 0.003, 1076, 3.0, 150, 8000)
tails = {
 "full": ['-ct', 2, '--readFQB', 'x.fqb', '--hashDepthRange', 3, 33, '--cluster', 1, 0, '--writeHash', 'mid.hash', '--readHash', 'mid.hash', '--hashDepthRange', 3, 35, '--cluster', 1, 0, '--clusterSplit', '--hashDepthRange', 3, 33, '--cluster', 1, 0],
 "noreadhash": ['-ct', 2, '--readFQB', 'x.fqb', '--hashDepthRange', 3, 33, '--cluster', 1, 0, '--clusterSplit', '--hashDepthRange', 3, 33, '--cluster', 1, 0],
 "readonly": ['-ct', 2, '--readFQB', 'x.fqb'],
 "cluster": ['-ct', 2, '--readFQB', 'x.fqb', '--hashDepthRange', 3, 33, '--cluster', 1, 0],
}
for name, tail in tails.items():
    for opts in ({}, {"index_no_pack": 1}):
        base = ["-k", k, "-w", w, "-r", r, "-B", B] + tail
        def make(k_, w_, r_, B_):
            h = hash10x_amd.Hash10x(k=k_, w=w_, r=r_, B=B_)
            for n, v in opts.items(): h.set_option(n, v)
            return h
        run_commands(make, base + ["--writeHash", "hip.hash"], d)
        run_commands(lambda k_, w_, r_, B_: orc.Oracle(k_, w_, r_, B_), base + ["--writeHash", "orc.hash"], d)
        got = open(os.path.join(d, "hip.hash"), "rb").read(); exp = open(os.path.join(d, "orc.hash"), "rb").read()
        print(name, opts, "OK" if got == exp else "DIFF " + orc.describe_diff(got, exp)[:300], flush=True)
