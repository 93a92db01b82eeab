"""Round 4: what-if timings of the translated placement on the 1/10 config-3 set (results wrong with a skip bit set).
   python scratch/r4_whatif.py <dbg_skip bits> ..."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, hash10x_amd
wl = bench.WORKLOADS["config3-tenth-20M"]; recs = bench.generate(wl, wl["seed"])
d = hash10x_amd.DeviceRecords(recs); del recs
h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True)
h.set_option("cluster_first_global", 4); h.set_option("cluster_stamps", 1)
h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"])
for a in sys.argv[1:]:
    kv = dict(x.split("=") for x in a.split(",")) if "=" in a else {"cluster_dbg_skip": a}
    for k, v in kv.items(): h.set_option(k, int(v))
    for it in range(2):
        h.timing_reset() if hasattr(h, "timing_reset") else None
        t0 = time.perf_counter(); h.cluster(1, 0, wl["ct"]); hash10x_amd.synchronize(0); t1 = time.perf_counter()
    c = h.counters(); t = c["cluster_phase_ticks"]; s = float(sum(t[:6])) or 1.0
    print("%s: cluster wall %.2f ms classes %s overflow %s; phases clear %.3f A(w0) %.3f Atail %.3f compact+init %.3f B(w0) %.3f rest %.3f; queued %d spilled %d ovf1 %d ovf2 %d" % (
        a, 1e3 * (t1 - t0), c["cluster_class_counts"], c["cluster_overflow_blocks"], *[x / s for x in t[:6]], t[6], t[7] & 0xFFFFF, (t[7] >> 20) & 0xFFFFF, t[7] >> 40), flush=True)
    for k in kv: h.set_option(k, 0)
