"""--readFQB alone on a generator-v2 workload, per-kernel device times (for timing experiments that leave a state unusable for the later commands)"""
import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import bench, hash10x_amd
wl = dict(bench.WORKLOADS[sys.argv[1]])
recs = bench.generate_v2(wl, wl["seed"])[0]
dr = hash10x_amd.DeviceRecords(recs); del recs
h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True)
for it in range(3):
    h.read_fqb_device(dr.ptr, dr.n_records); hash10x_amd.synchronize(0)
    print({k: round(v[0], 2) for k, v in h.timings().items() if v[0] > 0}, flush=True)
