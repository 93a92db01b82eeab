"""Same-process A/B of tuning knobs: one workload generated once, then for every option set on the command line (comma-separated name=value pairs, "base" = none)
   `passes` passes of --readFQB + --hashDepthRange + --cluster with the per-kernel device times.   python scratch/r5_opts.py <workload|config3-full> <passes> base cluster_threads0=768 ...
   (the knobs are reset to 0 between sets; a set is run again at the end if the list ends with "again": boxes drift)"""
import sys, os, time, json
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import numpy as np, bench, hash10x_amd
name, passes, sets = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
if name == "config3-full":
    man = json.load(open(os.path.join(R, "tests", "golden", "manifest.json")))["full_digest_cases"][0]; g = man["gen2"]
    wl = dict(pairs=g["pairs"], barcodes=g["barcodes"], genome=g["genome"], err=g["err"], mol=g["mol"], snp=g["snp"], mol_len=g["mol_len"], B=man["B"], lo=30, hi=100, ct=5, seed=g["seed"], gen=2)
else:
    wl = dict(bench.WORKLOADS[name])
t = time.time()
recs = bench.generate_v2(wl, wl["seed"])[0] if wl.get("gen") == 2 else bench.generate(wl, wl.get("seed", 1))
print("generated %d pairs in %.1f s" % (recs.size // 30, time.time() - t), flush=True)
dr = hash10x_amd.DeviceRecords(recs); del recs
h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True)
if os.environ.get("STAMPS"): h.set_option("cluster_stamps", 1)
used = set()
for s in sets:
    for k in used: h.set_option(k, -1 if k in ("cluster_tr_packed", "cluster_tr_class_t", "shard_delta_lists", "shard_row_shift") else 0)
    if s not in ("base", "again"):
        for kv in s.split(","):
            k, v = kv.split("="); h.set_option(k, int(v)); used.add(k)
    best = None
    for it in range(passes + 1):
        hash10x_amd.synchronize(0); t0 = time.time()
        h.read_fqb_device(dr.ptr, dr.n_records); h.depth_range(wl["lo"], wl["hi"]); h.cluster(1, 0, wl["ct"]); hash10x_amd.synchronize(0)
        dt = time.time() - t0
        tm = h.timings()
        if it and (best is None or tm["cluster_kernel"][0] < best[0]): best = (tm["cluster_kernel"][0], tm.get("cluster_main", (0, 0))[0], dt, {k: round(v[0], 2) for k, v in tm.items() if v[0] > 0})
    c = h.counters()
    if os.environ.get("STAMPS"):
        tk = c["cluster_phase_ticks"]; tot = float(sum(tk[:6])) or 1.0
        print("      phase shares: init %.3f passA %.3f barrier %.3f compact %.3f passB %.3f settle %.3f; ranks settled behind the loop %.3f" % (tuple(x / tot for x in tk[:6]) + (tk[6] / float(tk[7] or 1),)))
    print("%-40s cluster_kernel %.2f ms main %.2f step %.1f ms classes %s ovf %s mode %s\n      %s" % (s, best[0], best[1], 1e3 * best[2], c["cluster_class_counts"], c["cluster_overflow_blocks"], c["cluster_first_mode"], best[3]), flush=True)
