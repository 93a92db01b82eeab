"""Replays a reference command line (hash10x.c:1158-1279 argv grammar) against an engine object.

An engine is created by `factory(k, w, r, B)` when --readFQB / --readHash is met (that is where the
reference calls initialise(), hash10x.c:1202,1208) and must offer:
    read_fqb(records, N, chunk) / read_hash(path) / depth_range(lo, hi) /
    cluster(code_min, code_max, threshold) / cluster_split() / write_hash(path)
Both the CPU oracle (tests/orc.Oracle) and the product binding (hash10x_amd.Hash10x) do, so the same
golden command lines drive either.
"""
import os

import numpy as np


def run_commands(factory, args, cwd):
    p = {"k": 21, "w": 31, "r": 17, "B": 28, "N": 0, "c": 100000, "ct": 5}
    eng = None
    a = [str(x) for x in args]
    i = 0
    while i < len(a):
        t = a[i]
        if t in ("-k", "-w", "-r", "-B", "-N", "-c"):
            p[t[1:]] = int(a[i + 1]); i += 2
        elif t in ("-ct", "--clusterThreshold"):
            p["ct"] = int(a[i + 1]); i += 2
        elif t == "--readFQB":
            eng = factory(p["k"], p["w"], p["r"], p["B"])
            recs = np.fromfile(os.path.join(cwd, a[i + 1]), dtype=np.uint32)
            eng.read_fqb(recs, p["N"], p["c"]); i += 2
        elif t == "--readHash":
            eng = factory(p["k"], p["w"], p["r"], p["B"])
            eng.read_hash(os.path.join(cwd, a[i + 1])); i += 2
        elif t == "--writeHash":
            eng.write_hash(os.path.join(cwd, a[i + 1])); i += 2
        elif t == "--hashDepthRange":
            eng.depth_range(int(a[i + 1]), int(a[i + 2])); i += 3
        elif t == "--cluster":
            eng.cluster(int(a[i + 1]), int(a[i + 2]), p["ct"]); i += 3
        elif t == "--clusterSplit":
            eng.cluster_split(); i += 1
        else:
            raise ValueError("driver: unsupported token %s" % t)
    return eng
