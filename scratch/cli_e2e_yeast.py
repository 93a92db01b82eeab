"""The bench's end-to-end command on the yeast-scale file, a few times, with and without the warm-up thread (H10X_NOWARM=1): python scratch/cli_e2e_yeast.py"""
import sys, os, time, subprocess
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import bench
wl = bench.WORKLOADS["yeast-like-2.5M"]
recs = bench.generate(wl, wl.get("seed", 1)); path = "/tmp/y.fqb"; recs.tofile(path)
exe = os.path.join(R, "bin", "hash10x-amd")
cmd = [exe, "-B", str(wl["B"]), "--readFQB", path, "--hashDepthRange", str(wl["lo"]), str(wl["hi"]), "--cluster", "1", "0", "--writeHash", "/tmp/y.hash"]
for env in ({}, {"H10X_NOWARM": "1"}, {}, {"H10X_NOWARM": "1"}):
    for it in range(3):
        t = time.perf_counter(); r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, H10X_INGEST_TIMING="1", H10X_HOSTPROF="1", **env)); dt = time.perf_counter() - t
        assert r.returncode == 0, r.stderr.decode()[-400:]
        walls = [l.split()[-1] for l in r.stdout.decode().splitlines() if l.strip().startswith("user")]
        print(env, "%.3f s" % dt, walls, " | ".join(l.strip() for l in r.stderr.decode().splitlines() if "hostprof" in l or "ingest of" in l), flush=True)
