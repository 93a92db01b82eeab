"""Round 3 timing probe: per-step device times of the yeast-scale set and of the 1/10 config-3 set (placements given as argv: 0 auto, 2 ranked, 3 hashed).
   python scratch/r3_time.py [yeast] [c3:0] [c3:3] ..."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, hash10x_amd

def run(wl, d, mode, steps=3, stamps=False, opts=None):
    h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True)
    if mode: h.set_option("cluster_first_global", mode)
    if stamps: h.set_option("cluster_stamps", 1)
    for k, v in (opts or {}).items(): h.set_option(k, v)
    tm = {}
    for it in range(steps + 1):
        h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"]); h.cluster(1, 0, wl["ct"])
        hash10x_amd.synchronize(0)
        if it:
            for k, (ms, n) in h.timings().items():
                a = tm.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += n
    c = h.counters()
    alg = 4.0 * c["cluster_main"][1] + 14.0 * c["cluster_main"][0] + 16.0 * c["cluster_main"][2]
    algAll = 4.0 * c["sum_good_depth"] + 14.0 * c["sum_good"] + 16.0 * c["sum_hash_clustered"]
    main = tm["cluster_main"][0] / steps; allk = tm["cluster_kernel"][0] / steps
    print("  mode %d: %s" % (mode, {k: round(v[0] / steps, 3) for k, v in tm.items() if v[0] > 0}), flush=True)
    print("     cluster_main %.3f ms = %.1f GB/s (%.1f %%), all cluster launches %.3f ms = %.1f GB/s (%.1f %%); placement %s classes %s overflow %s" % (
        main, alg / main / 1e6 if main else 0, alg / main / 1e6 / 80 if main else 0, allk, algAll / allk / 1e6, algAll / allk / 1e6 / 80,
        c["cluster_first_mode"], c["cluster_class_counts"], c["cluster_overflow_blocks"]), flush=True)
    if stamps:
        t = c["cluster_phase_ticks"]; s = float(sum(t)) or 1.0
        print("     phase shares: init %.3f list %.3f barrier %.3f settle %.3f regather %.3f output %.3f" % tuple(x / s for x in t[:6]), flush=True)
    h.close()

args = sys.argv[1:] or ["yeast", "c3:0"]
if any(a.startswith("yeast") for a in args):
    wl = bench.WORKLOADS["yeast-like-2.5M"]; recs = bench.generate(wl, 1); d = hash10x_amd.DeviceRecords(recs); del recs
    print("yeast-like-2.5M", flush=True)
    for a in args:
        if a.startswith("yeast"):
            run(wl, d, 0, steps=5, stamps=a.endswith("+s"))
    d.free()
c3 = [a for a in args if a.startswith("c3:")]
if c3:
    wl = bench.WORKLOADS["config3-tenth-20M"]; t0 = time.time(); recs = bench.generate(wl, wl["seed"]); print("config3-tenth-20M (generated in %.1f s)" % (time.time() - t0), flush=True)
    d = hash10x_amd.DeviceRecords(recs); del recs
    for a in c3:
        run(wl, d, int(a[3:4]), steps=2, stamps=a.endswith("+s"))
    d.free()
