"""One rank of a multi-PROCESS sharded run (test infrastructure): `python shard_worker.py <rank> <nranks> <port> <fqb> <B> <lo> <hi> <ct> <out.hash>`.
Uses the host-staged socket communicator, so several ranks can share the one GPU of a test box (RCCL refuses that)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hash10x_amd

rank, n, port, fqb, B, lo, hi, ct, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]), sys.argv[9]
recs = np.fromfile(fqb, dtype=np.uint32)
cut = hash10x_amd.partition(recs, n)
comm = hash10x_amd.Comm.socket(rank, n, "127.0.0.1", port)
h = hash10x_amd.Hash10x(B=B)
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
