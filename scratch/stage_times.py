"""Best-of-N device time of every timed stage of the yeast-scale step (readFQB + hashDepthRange + cluster).
usage: stage_times.py [reps=5]   — environment knobs of experimental builds are read by the library itself."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, hash10x_amd
wl = bench.WORKLOADS[os.environ.get("H10X_WL", "yeast-like-2.5M")]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
recs = bench.generate(wl, 1)
d = hash10x_amd.DeviceRecords(recs)
best = {}
h = hash10x_amd.Hash10x(B=wl["B"]); h.enable_timing(True)
prev = {}
for it in range(reps + 1):                                    # one session, as the bench: the first pass (allocations) is dropped
    h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"]); h.cluster(1, 0, wl["ct"])
    now = {k: v[0] for k, v in h.timings().items()}
    if it:
        for k, v in now.items():
            best[k] = min(best.get(k, 1e9), v)           # the timers restart at every readFQB
    prev = now
h.close()
print(os.environ.get("H10X_TAG", ""), " ".join("%s %.3f" % (k, v) for k, v in best.items() if v > 0.0005), "| sum %.3f" % sum(best.values()), flush=True)
