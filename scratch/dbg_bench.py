import sys, os, time, hashlib
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import numpy as np, bench, hash10x_amd
wl = bench.WORKLOADS["yeast-like-2.5M"]
t=time.time(); recs = bench.generate(wl, 1); print("gen s", time.time()-t, "sha", hashlib.sha256(recs.tobytes()).hexdigest()[:16], flush=True)
d = hash10x_amd.DeviceRecords(recs)
h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True)
for it in range(3):
    t0=time.time()
    try:
        h.read_fqb_device(d.ptr, d.n_records); print("ok", h.sizes(), time.time()-t0, flush=True)
    except Exception as e:
        print("ERR", e, time.time()-t0, h.timings(), flush=True)
