"""Oracle restatement vs the REAL reference binary on fresh seeded inputs. Runs only where
oracle/_ref exists (the build container; the binaries also travel to the GPU box)."""
import os

import numpy as np
import pytest

import orc
from driver import run_commands

pytestmark = pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.mark.parametrize("seed,pairs,barcodes,genome,mol,mol_len,lo,hi,ct", [
    (21, 6000, 50, 50000, 3.0, 3000, 3, 14, 2),
    (22, 9000, 30, 200000, 5.0, 8000, 2, 10, 1),
    (23, 5000, 25, 30000, 2.0, 2500, 4, 20, 3),
])
def test_random_sets(tmp_path, seed, pairs, barcodes, genome, mol, mol_len, lo, hi, ct):
    cwd = str(tmp_path)
    orc.gen_fqb(os.path.join(cwd, "x.fqb"), pairs, barcodes, genome, 0.004, seed, mol, 150, mol_len)
    args = ["-B", 20, "-ct", ct, "--readFQB", "x.fqb", "--writeHash", "a.hash", "--hashDepthRange", lo, hi,
            "--cluster", 1, 0, "--writeHash", "b.hash", "--clusterSplit", "--writeHash", "c.hash"]
    r = orc.run_ref(args, cwd)
    assert r.returncode == 0, r.stderr.decode()
    ref = {n: orc.canonical_hash_bytes(open(os.path.join(cwd, n), "rb").read()) for n in ("a.hash", "b.hash", "c.hash")}
    for n in ref:
        os.remove(os.path.join(cwd, n))
    run_commands(lambda k, w, r_, B: orc.Oracle(k, w, r_, B), args, cwd)
    for n in ("a.hash", "b.hash", "c.hash"):
        got = open(os.path.join(cwd, n), "rb").read()
        assert got == ref[n], n + ": " + orc.describe_diff(got, ref[n])
    assert int(orc.HashFile(ref["b.hash"]).blocks["nSubCluster"].sum()) > 0


def test_omp_cluster_matches_serial(tmp_path):
    cwd = str(tmp_path)
    recs = orc.gen_fqb(os.path.join(cwd, "x.fqb"), 6000, 50, 50000, 0.004, 31, 3.0, 150, 3000)
    outs = []
    for threads in (1, 4):
        o = orc.Oracle(B=20)
        o.read_fqb(recs)
        o.depth_range(3, 14)
        o.cluster(1, 0, 2, threads)
        o.write_hash(os.path.join(cwd, "t%d.hash" % threads))
        outs.append(open(os.path.join(cwd, "t%d.hash" % threads), "rb").read())
    assert outs[0] == outs[1]
