import os, sys, subprocess, tempfile
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R + "/tests"); sys.path.insert(0, R)
import orc, numpy as np
d = tempfile.mkdtemp()
orc.gen_fqb(os.path.join(d, "x.fqb"), 3000, 150, 12000, 0.003, 2043, 4.0, 150, 2500, fa=os.path.join(d, "x"))
args = "-k 19 -w 32 -r 5 -B 21 -ct 1 --readFQB x.fqb --hashStats --hashDepthRange 4 7 --cluster 1 0 --tables --cribBuild x.A.fa x.B.fa --clusterReport 0 0 --clusterReport 1 75 --clusterSplit --cribSummary --hashDepthRange 4 7 --cluster 1 0 --writeHash OUT".split()
for variant in (sys.argv[1:] or ["full"]):
    a = list(args)
    if variant == "nocrib": a = [x for x in " ".join(a).replace("--tables --cribBuild x.A.fa x.B.fa --clusterReport 0 0 --clusterReport 1 75 ", "").replace("--cribSummary ", "").split()]
    ref = orc.run_ref([x if x != "OUT" else "ref.hash" for x in a], d)
    exp = orc.HashFile(orc.canonical_hash_bytes(open(os.path.join(d, "ref.hash"), "rb").read()))
    for gpus in (1, 2, 3):
        hip = subprocess.run([R + "/bin/hash10x-amd"] + (["--gpus", str(gpus)] if gpus > 1 else []) + [x if x != "OUT" else "hip.hash" for x in a], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        print(variant, "gpus", gpus, "rc", hip.returncode, hip.stderr.decode()[-300:])
        raw = open(os.path.join(d, "hip.hash"), "rb").read()
        try:
            got = orc.HashFile(raw)
            print("  hash_number", got.hash_number, exp.hash_number, "blocks_max", got.blocks_max, exp.blocks_max, "sum nHash", int(got.blocks["nHash"].sum()), int(exp.blocks["nHash"].sum()), "size", len(raw), len(exp.data) if hasattr(exp, "data") else None)
            for f in ("nRead", "nHash", "nSubCluster", "clusterParent"):
                a1, b1 = got.blocks[f], exp.blocks[f]
                n = min(len(a1), len(b1)); bad = np.nonzero(a1[:n] != b1[:n])[0]
                print("   ", f, "len", len(a1), len(b1), "first diffs", bad[:8], a1[bad[:4]], b1[bad[:4]])
        except Exception as e:
            print("  parse failed", e, len(raw))
