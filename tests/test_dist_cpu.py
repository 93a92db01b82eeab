"""world_size-2 gloo test of the multi-process plumbing bench.py relies on at N > 1 (one process per
GPU, barrier + max-over-ranks timing, per-rank seeded samples) and of the barcode-range partition the
sharded path will use. CPU only; the compute in it is the oracle (test infrastructure)."""
import os
import socket
import subprocess
import sys

import orc

WORKER = r'''
import ctypes, os, sys, json
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["H10X_REPO"]); sys.path.insert(0, os.path.join(os.environ["H10X_REPO"], "tests"))
import orc, hash10x_amd
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
recs = np.frombuffer(orc.read_maybe_gz(os.path.join(orc.GOLDEN, "small.fqb.gz")), dtype=np.uint32).copy()
n = recs.size // 30
host = hash10x_amd.load_native()[1]
cut = (ctypes.c_uint64 * (world + 1))()
assert host.h10x_host_partition(recs.ctypes.data, n, world, cut) == 0
lo, hi = cut[rank], cut[rank + 1]
# every rank hashes its own contiguous barcode range (+ the first record of the next shard, which
# plays the "next barcode" that closes the last block: hash10x.c:213-220)
end = hi + 1 if rank + 1 < world else hi
o = orc.Oracle(B=20); o.read_fqb(recs[30 * lo: 30 * end])
b, _, mx = o.blocks()
nblk = mx - 1 - (1 if rank + 1 < world else 0)          # blocks this rank owns (the extra closing block is not ours)
own = torch.tensor([int(nblk), int(b["nHash"][1:1 + nblk].sum()), int(hi - lo)], dtype=torch.int64)
allv = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
dist.all_gather(allv, own)
t = torch.tensor([0.5 + rank], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
if rank == 0:
    full = orc.Oracle(B=20); full.read_fqb(recs)
    fb, _, fmx = full.blocks()
    print(json.dumps({"blocks": int(sum(int(v[0]) for v in allv)), "full_blocks": int(fmx - 1),
                      "entries": int(sum(int(v[1]) for v in allv)), "full_entries": int(fb["nHash"].sum()),
                      "records": int(sum(int(v[2]) for v in allv)), "n": int(n), "tmax": float(t.item())}))
dist.destroy_process_group()
'''


def test_two_rank_gloo_shards_cover_the_file(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    wf = tmp_path / "worker.py"
    wf.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   H10X_REPO=orc.REPO, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(wf)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se.decode()[-2000:]
    import json
    res = json.loads(outs[0][0].decode().strip().splitlines()[-1])
    assert res["records"] == res["n"]
    assert res["blocks"] == res["full_blocks"]                 # shards own disjoint, complete barcode ranges
    # per-barcode unique hash sets are shard-local, so entry counts add up (the file's last barcode is unhashed in both)
    assert res["entries"] == res["full_entries"]
    assert res["tmax"] == 1.5


RDV = r'''
import os, sys
sys.path.insert(0, os.environ["H10X_REPO"])
import bench
class Fake:                      # stands in for hash10x_amd.Comm.unique_id() on a box without RCCL devices
    class Comm:
        @staticmethod
        def unique_id():
            return bytes(range(128))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
uid = bench.rendezvous_unique_id(rank, world, Fake)
assert uid == bytes(range(128)), uid
print("ok", rank)
'''


def test_rccl_unique_id_rendezvous_three_ranks(tmp_path):
    """bench.py's torch-free bootstrap: rank 0 serves the 128-byte ncclUniqueId to the other ranks over TCP."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    wf = tmp_path / "rdv.py"
    wf.write_text(RDV)
    procs = []
    for r in (2, 1, 0):                               # clients first: they must wait for the server
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), H10X_REPO=orc.REPO)
        procs.append(subprocess.Popen([sys.executable, str(wf)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    for p in procs:
        so, se = p.communicate(timeout=120)
        assert p.returncode == 0, se.decode()[-1500:]
        assert so.decode().startswith("ok")


STRONG_WORKER = r"""
import os, sys, json
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["H10X_REPO"]); sys.path.insert(0, os.path.join(os.environ["H10X_REPO"], "tests"))
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
wl = dict(pairs=6000, barcodes=40, genome=60000, err=0.001, mol=3.0, snp=150, mol_len=4000.0)
mine, first, total = bench.generate_v2(wl, 3, rank, world)          # strong scaling: the SAME set, every rank builds only its shard
n = mine.size // 30
# a composable checksum over the shard's records at their place in the whole set: what bench.py sums over the ranks for its parity gate
cs = bench.checksum_words(mine.view(np.uint64), first * 15, 7, slice_words=1000)
t = torch.tensor([n, first] + [c >> 32 for c in cs] + [c & 0xFFFFFFFF for c in cs], dtype=torch.int64)
allv = [torch.zeros_like(t) for _ in range(world)]
dist.all_gather(allv, t)
dist.barrier()
if rank == 0:
    whole, f0, tot = bench.generate_v2(wl, 3)
    ref = bench.checksum_words(whole.view(np.uint64), 0, 7)
    summed = [sum(((int(v[2 + k]) << 32) | int(v[4 + k])) for v in allv) & 0xFFFFFFFFFFFFFFFF for k in range(2)]
    starts = [int(v[1]) for v in allv]; counts = [int(v[0]) for v in allv]
    print(json.dumps({"pairs": tot, "whole": whole.size // 30, "counts": counts, "starts": starts, "sum_ok": summed == ref,
                      "serial_ok": bench.checksum_words(whole.view(np.uint64), 0, 7, slice_words=777) == ref}))
dist.destroy_process_group()
"""


def test_strong_scaling_shards_of_one_set_add_up(tmp_path):
    """bench.py --scaling strong: the ranks generate disjoint, contiguous shards of ONE fixed generator-v2 set (not N copies of a set), and the
    composable checksum the parity gate sums over the ranks equals the one of the whole set — here over the records, two ranks on gloo."""
    import json
    orc.build_gen()
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-DH10X_GEN_NO_MAIN", "-o", os.path.join(orc.REPO, "build", "libgen_fqb.so"),
                    os.path.join(orc.REPO, "hash10x_amd", "tools", "gen_fqb.c"), "-lm"], check=True)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    wf = tmp_path / "strong_worker.py"
    wf.write_text(STRONG_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), H10X_REPO=orc.REPO, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(wf)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se.decode()[-2000:]
    res = json.loads(outs[0][0].decode().strip().splitlines()[-1])
    assert res["whole"] == res["pairs"] == sum(res["counts"]) and res["starts"] == [0, res["counts"][0]]
    assert res["sum_ok"] and res["serial_ok"]
