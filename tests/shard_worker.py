"""One rank of a multi-PROCESS sharded run (test infrastructure): `python shard_worker.py <rank> <nranks> <port | rccl:idfile> <fqb> <B> <lo> <hi> <ct> <out.hash>`.
A port number selects the host-staged socket communicator, so several ranks can share the one GPU of a test box (RCCL refuses that);
`rccl:<file>` selects RCCL with one GPU per rank (device = rank): rank 0 writes the 128-byte ncclUniqueId to <file>, the others wait for it."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hash10x_amd

rank, n, where, fqb, B, lo, hi, ct, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]), sys.argv[9]
recs = np.fromfile(fqb, dtype=np.uint32)
cut = hash10x_amd.partition(recs, n)
device = 0
if where.startswith("rccl:"):
    path = where[5:]
    if rank == 0:
        uid = hash10x_amd.Comm.unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(path + ".tmp", path)
    else:
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > 300:
                raise SystemExit("no ncclUniqueId from rank 0")
            time.sleep(0.05)
        uid = open(path, "rb").read()
    device = rank
    comm = hash10x_amd.Comm.rccl(rank, n, uid, device)
else:
    comm = hash10x_amd.Comm.socket(rank, n, "127.0.0.1", int(where))
h = hash10x_amd.Hash10x(B=B, device=device)
h.shard_read_fqb(comm, recs[30 * cut[rank]: 30 * cut[rank + 1]])
h.depth_range(lo, hi)
h.cluster(1, 0, ct)
h.cluster_split()
h.depth_range(lo, hi)
h.cluster(1, 0, ct)
assert h.shard_allreduce_max(float(rank)) == float(n - 1)
h.write_hash(out)
h.close()
comm.destroy()
print("rank %d done" % rank)
