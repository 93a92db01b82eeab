"""two passes of BASELINE configs[2] at its own size (the set of tests/golden/manifest.json "full_digest_cases") for rocprofv3: kernel trace / counters"""
import sys, os, json
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import bench, hash10x_amd
man = json.load(open(os.path.join(R, "tests", "golden", "manifest.json")))["full_digest_cases"][0]
g = man["gen2"]
wl = dict(pairs=g["pairs"], barcodes=g["barcodes"], genome=g["genome"], err=g["err"], mol=g["mol"], snp=g["snp"], mol_len=g["mol_len"])
recs, _f, _t = bench.generate_v2(wl, g["seed"])
d = hash10x_amd.DeviceRecords(recs); hash10x_amd.synchronize(0)
del recs
h = hash10x_amd.Hash10x(B=man["B"]); h.enable_timing(True)
for it in range(2):
    h.read_fqb_device(d.ptr, d.n_records); h.depth_range(30, 100); h.cluster(1, 0, 5)
c = h.counters()
print({k: c[k] for k in ("sum_good", "sum_good_depth", "cluster_main", "cluster_class_counts", "cluster_overflow_blocks", "cluster_first_mode")})
print({k: round(v[0], 2) for k, v in h.timings().items() if v[0] > 0})
