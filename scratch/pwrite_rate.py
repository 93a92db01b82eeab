"""How fast a file can be written into the page cache of this box: pwrite from T threads, chunk sizes 8 / 32 MiB, fresh file each time: python scratch/pwrite_rate.py [GB]"""
import os, sys, time, threading
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
n = int(gb * (1 << 30)); buf = bytearray(os.urandom(1 << 20)) * 64; mv = memoryview(buf)        # 64 MiB source
for T in (1, 2, 4, 8, 16, 32):
    for fresh in (True, False):
        path = "/tmp/pw.bin"
        if fresh and os.path.exists(path): os.remove(path)
        fd = os.open(path, os.O_WRONLY | os.O_CREAT, 0o666); os.ftruncate(fd, n)
        def job(a, b):
            at = a
            while at < b:
                k = min(b - at, len(buf)); os.pwrite(fd, mv[:k], at); at += k
        t = time.perf_counter()
        ths = [threading.Thread(target=job, args=(n * i // T, n * (i + 1) // T)) for i in range(T)]
        [x.start() for x in ths]; [x.join() for x in ths]
        dt = time.perf_counter() - t; os.close(fd)
        print("%2d threads, %s file: %.3f s = %.1f GB/s" % (T, "new" if fresh else "rewritten", dt, n / dt / 1e9), flush=True)
os.remove("/tmp/pw.bin")
