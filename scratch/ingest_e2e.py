"""End to end through the command line on a 2.4 GB file (the 1/10 BASELINE configs[2] set): bin/hash10x-amd --readFQB (pipelined ingest: pinned slabs, uploads
beside the file reads) against the time of just reading the file and of just copying it to the device.   python scratch/ingest_e2e.py [workload]"""
import sys, os, time, subprocess, threading
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import numpy as np, bench, hash10x_amd
name = sys.argv[1] if len(sys.argv) > 1 else "config3-tenth-20M"
wl = bench.WORKLOADS[name]
recs = bench.generate(wl, wl.get("seed", 1))
path = "/tmp/e2e.fqb"; recs.tofile(path); nbytes = recs.nbytes
print("%s: %.2f GB written to %s" % (name, nbytes / 1e9, path), flush=True)
def read_all(threads):
    fd = os.open(path, os.O_RDONLY); buf = bytearray(nbytes); mv = memoryview(buf)
    def part(a, b):
        at = a
        while at < b:
            at += os.preadv(fd, [mv[at:min(b, at + (64 << 20))]], at)
    t = time.perf_counter()
    ths = [threading.Thread(target=part, args=(nbytes * i // threads, nbytes * (i + 1) // threads)) for i in range(threads)]
    [x.start() for x in ths]; [x.join() for x in ths]
    dt = time.perf_counter() - t; os.close(fd); return dt
for th in (1, 4): print("file read, %d thread(s): %.3f s = %.1f GB/s" % (th, read_all(th), nbytes / read_all(th) / 1e9), flush=True)
t = time.perf_counter(); d = hash10x_amd.DeviceRecords(recs); hash10x_amd.synchronize(0); dt = time.perf_counter() - t
print("host -> device (pageable numpy array): %.3f s = %.1f GB/s" % (dt, nbytes / dt / 1e9), flush=True); d.free(); del recs
exe = os.path.join(R, "bin", "hash10x-amd")
for args in (["-B", str(wl["B"]), "--readFQB", path], ["-B", str(wl["B"]), "--readFQB", path, "--hashDepthRange", str(wl["lo"]), str(wl["hi"]), "--cluster", "1", "0", "--writeHash", "/tmp/e2e.hash"]):
    for it in range(2):
        if os.path.exists("/tmp/e2e.hash"): os.remove("/tmp/e2e.hash")   # (a NEW file, as in the bench: truncating 1.5 GB of an old one and the flush ext4 then does at close cost 0.3 s)
        t = time.perf_counter(); r = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, H10X_INGEST_TIMING="1")); dt = time.perf_counter() - t
        assert r.returncode == 0, r.stderr.decode()[-500:]
    lines = [l for l in r.stdout.decode().splitlines() if "wall" in l or l.startswith("COMMAND")]
    print("CLI %s: %.3f s wall (process)\n   %s\n   %s" % (" ".join(args[2:6]), dt, "\n   ".join(lines[:12]), r.stderr.decode().strip()), flush=True)
