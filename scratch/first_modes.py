"""Time --cluster of the N x yeast-scale set on one GPU with the first[] placements that apply: default, ranked (2), hashed (3). usage: first_modes.py N"""
import sys, os, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, bench, hash10x_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
wl = dict(bench.WORKLOADS["yeast-like-2.5M"])
wl["pairs"] *= N; wl["barcodes"] *= N; wl["genome"] *= N; wl["B"] = min(30, wl["B"] + (N - 1).bit_length())
t = time.time(); recs = bench.generate(wl, 1); print("generated", recs.size // 30, "pairs in %.1fs" % (time.time() - t), flush=True)
dr = hash10x_amd.DeviceRecords(recs)
for mode in (0, 2, 3):
    h = hash10x_amd.Hash10x(B=wl["B"]); h.set_option("cluster_first_global", mode); h.enable_timing(True)
    h.read_fqb_device(dr.ptr, dr.n_records); h.depth_range(wl["lo"], wl["hi"])
    for it in range(3): h.cluster(1, 0, wl["ct"])
    c = h.counters(); tm = h.timings()
    print("knob %d -> first_mode %d classes %s overflow %d: cluster %.2f ms, main launch %.2f ms" % (mode, c["cluster_first_mode"], c["cluster_class_counts"], c["cluster_overflow_blocks"],
          tm["cluster"][0] / 3, tm["cluster_main"][0] / 3), flush=True)
    h.close()
