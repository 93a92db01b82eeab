"""The 1/4-scale BASELINE configs[2] set (50 M pairs, 300 k barcodes: between the ranked placement's sure range and its limit): ranked (auto) against translated (forced).
   python scratch/r4_c3quarter.py"""
import sys, os, time, json
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import bench, hash10x_amd
case = [c for c in json.load(open(os.path.join(R, "tests", "golden", "manifest.json")))["big_digest_cases"] if c["gen"]["barcodes"] > 262144][0]
g = case["gen"]
wl = dict(pairs=g["pairs"], barcodes=g["barcodes"], genome=g["genome"], err=g["err"], mol=g.get("mol", 10.0), snp=g.get("snp", 150), mol_len=g.get("mol_len", 50000.0))
t = time.time(); recs = bench.generate(wl, g["seed"]); print("generated %d pairs in %.1f s" % (recs.size // 30, time.time() - t), flush=True)
d = hash10x_amd.DeviceRecords(recs); del recs
for mode in (0, 4):
    h = hash10x_amd.Hash10x(B=case["B"]); h.enable_timing(True)
    if mode: h.set_option("cluster_first_global", mode)
    for it in range(3):
        h.read_fqb_device(d.ptr, d.n_records); h.depth_range(30, 100); h.cluster(1, 0, 5); hash10x_amd.synchronize(0)
    tm = h.timings(); c = h.counters()
    print("mode %d: placement %s classes %s overflow %s cluster_kernel %.1f ms cluster_main %.1f ms" % (mode, c["cluster_first_mode"], c["cluster_class_counts"], c["cluster_overflow_blocks"], tm["cluster_kernel"][0], tm["cluster_main"][0]), flush=True)
    h.close()
