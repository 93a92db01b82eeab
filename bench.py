#!/usr/bin/env python3
"""bench.py — hash10x hot path on MI355X: --readFQB -> --hashDepthRange -> --cluster.

One "step" = one pass of the whole path over one synthetic linked-read set that is already resident
in HBM when the timed region starts: mosh construction + global hash<->barcode index (everything
`--readFQB` leaves in memory), good-hash lists (`--hashDepthRange lo hi`) and per-barcode clustering
(`--cluster 1 0`). Workload at N=1 = BASELINE.json configs[1]: the yeast-scale set (2.5 M read
pairs, 10 k barcodes, -B 24) — LRSIM and the yeast genomes are not available offline, so the seeded
generator of hash10x_amd/tools/gen_fqb.c stands in.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU; the workload is N times the yeast-scale set (weak scaling), its barcodes
sharded over the ranks and the global hash<->barcode index built by RCCL all-to-all over xGMI
(csrc/shard.hip, DESIGN.md "Multi-GPU"). RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* are read from the
environment as torch.distributed.run sets them; torch itself is not imported (see rendezvous_unique_id).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant
kernel and `cpu_baseline` (the reference binary from oracle/_ref, or the oracle port, timed on the
host cores of this box on the same workload).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable

WORKLOADS = {
    # name: generator parameters, table bits, depth range (see DESIGN.md for how the range was picked)
    "yeast-like-2.5M": dict(pairs=2500000, barcodes=10000, genome=12000000, err=0.005, mol=10.0, snp=150, mol_len=50000.0,
                            B=24, lo=30, hi=100, ct=5),
    "small-0.25M": dict(pairs=250000, barcodes=1000, genome=1200000, err=0.005, mol=10.0, snp=150, mol_len=50000.0,
                        B=22, lo=30, hi=100, ct=5),
    # BASELINE configs[2] (500 Mb x 2, 200 M pairs, 1 M barcodes, e = 0.1 %) at 1/10 scale: the `secondary` block of the bench
    # line (ranked placement of first[], several thousand barcodes on the > 255 clusters path). -B 26: the table size that
    # holds its 10.4 M hashes (BASELINE's -B 28 for the full set is too small, see tests "config3_dies...").
    # BASELINE configs[3]/[4] shape — a 3 Gb genome x 2 haplotypes at -B 30 — at the largest read-pair count whose reference run the build
    # container's 62 GB of RAM hold (300 M pairs, 1.6 M barcodes; the reference's own cap, 2^28 - 2 distinct hashes, would allow ~380 M:
    # distinct hashes ~ 103 M of the genome + 0.43 per pair). e = 0.05 %. 15 x coverage per haplotype puts the het /
    # hom depth peaks at ~11 / ~22 (scratch/r4_g3_hist.py on the 1/10 proxy), hence --hashDepthRange 6 45. Generator v2 (shard-local).
    # This is the strong-scaling workload: the SAME set on 1, 2, 4, 8 ranks, pinned by the reference binary's digest (manifest "strong_digests").
    "genome3g-300M": dict(pairs=300000000, barcodes=1600000, genome=3000000000, err=0.0005, mol=10.0, snp=150, mol_len=50000.0,
                          B=30, lo=6, hi=45, ct=5, seed=3, gen=2),
    # the same proportions at 1/10 (same coverage, same depth range): the strong-scaling mode's functional check on test boxes (two ranks on one GPU)
    "genome3g-tenth-30M": dict(pairs=30000000, barcodes=160000, genome=300000000, err=0.0005, mol=10.0, snp=150, mol_len=50000.0,
                               B=27, lo=6, hi=45, ct=5, seed=3, gen=2),
    # the same proportions at 1/4: the virtual-rank model at a second size (how much of the modelled 8-rank step is fixed cost) — no digest, parity is pinned at 1/10 and at full size
    "genome3g-quarter-75M": dict(pairs=75000000, barcodes=400000, genome=750000000, err=0.0005, mol=10.0, snp=150, mol_len=50000.0,
                                 B=28, lo=6, hi=45, ct=5, seed=3, gen=2),
    # ... and at 1/2: the largest set the model is run on (1/10 sets misled three times) — 18 GB of records, one unsharded context and then eight ranks' shards on ONE GPU
    "genome3g-half-150M": dict(pairs=150000000, barcodes=800000, genome=1500000000, err=0.0005, mol=10.0, snp=150, mol_len=50000.0,
                               B=29, lo=6, hi=45, ct=5, seed=3, gen=2),
    "config3-tenth-20M": dict(pairs=20000000, barcodes=100000, genome=50000000, err=0.001, mol=10.0, snp=150, mol_len=50000.0,
                              B=26, lo=30, hi=100, ct=5, seed=2),
}


class GenParams(ctypes.Structure):
    _fields_ = [("pairs", ctypes.c_uint64), ("barcodes", ctypes.c_uint32), ("genome", ctypes.c_uint64), ("err", ctypes.c_double),
                ("seed", ctypes.c_uint64), ("mean_mol", ctypes.c_double), ("snp_spacing", ctypes.c_uint32), ("mean_len", ctypes.c_double)]


def generate(wl, seed):
    so = os.path.join(REPO, "build", "libgen_fqb.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    g = ctypes.CDLL(so)
    g.h10x_gen_fqb.restype = ctypes.c_uint64
    g.h10x_gen_fqb.argtypes = [ctypes.POINTER(GenParams), ctypes.c_void_p]
    p = GenParams(wl["pairs"], wl["barcodes"], wl["genome"], wl["err"], seed, wl["mol"], wl["snp"], wl["mol_len"])
    out = np.empty(wl["pairs"] * 30, dtype=np.uint32)
    n = g.h10x_gen_fqb(ctypes.byref(p), out.ctypes.data)
    g.h10x_gen_free()
    assert n == wl["pairs"]
    return out


def _gen_lib():
    so = os.path.join(REPO, "build", "libgen_fqb.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    g = ctypes.CDLL(so)
    g.h10x_gen2_plan_new.restype = ctypes.c_void_p
    g.h10x_gen2_plan_new.argtypes = [ctypes.POINTER(GenParams)]
    g.h10x_gen2_offsets.restype = ctypes.POINTER(ctypes.c_uint64)
    g.h10x_gen2_offsets.argtypes = [ctypes.c_void_p]
    g.h10x_gen2_cut.restype = ctypes.c_uint32
    g.h10x_gen2_cut.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    g.h10x_gen2_fill.restype = ctypes.c_uint64
    g.h10x_gen2_fill.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    g.h10x_gen2_free.argtypes = [ctypes.c_void_p]
    return g


def generate_v2(wl, seed, part=0, parts=1):
    """Generator v2 (gen_fqb.c: counter-based streams, OpenMP): the records of part `part` of `parts` of the data set — the barcodes
    whose records start in [P part / parts, P (part + 1) / parts) — WITHOUT generating the rest: a rank of a sharded run builds only
    its own shard. Returns (records, first record of the part in the whole set, pairs of the whole set)."""
    g = _gen_lib()
    p = GenParams(wl["pairs"], wl["barcodes"], wl["genome"], wl["err"], seed, wl["mol"], wl["snp"], wl["mol_len"])
    plan = g.h10x_gen2_plan_new(ctypes.byref(p))
    if not plan:
        raise MemoryError("gen_fqb v2: plan")
    try:
        P, C = wl["pairs"], wl["barcodes"]
        b0 = g.h10x_gen2_cut(plan, P * part // parts) if part else 0
        b1 = g.h10x_gen2_cut(plan, P * (part + 1) // parts) if part + 1 < parts else C
        off = g.h10x_gen2_offsets(plan)
        first, n = int(off[b0]), int(off[b1]) - int(off[b0])
        out = np.empty(max(n, 1) * 30, dtype=np.uint32)
        got = g.h10x_gen2_fill(plan, b0, b1, out.ctypes.data)
        assert got == n
        return out[: n * 30], first, P
    finally:
        g.h10x_gen2_free(plan)


def haplotypes_v2(wl, seed):
    """the two truth haplotypes of a generator-v2 workload as base codes 0..3 (what the FASTAs of `gen_fqb -fa` hold): the input of --cribBuild"""
    g = _gen_lib()
    g.h10x_gen2_haplotype.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    p = GenParams(wl["pairs"], wl["barcodes"], wl["genome"], wl["err"], seed, wl["mol"], wl["snp"], wl["mol_len"])
    plan = g.h10x_gen2_plan_new(ctypes.byref(p))
    if not plan:
        raise MemoryError("gen_fqb v2: plan")
    try:
        out = []
        for hap in (0, 1):
            a = np.empty(wl["genome"], dtype=np.uint8)
            g.h10x_gen2_haplotype(plan, hap, a.ctypes.data)
            out.append(a)
        return out
    finally:
        g.h10x_gen2_free(plan)


# ---- a checksum of a clustered state that ranks can add up: every 8-byte word of the ClusterBlock array (heap-pointer word zeroed) and of
# the ClusterHash records, in the FILE's numbering, goes through a 64-bit mixer together with its position, and the results are summed
# mod 2^64 under two different salts. A rank sums over its own segments; the sum over ranks is what the same function gives for the
# reference binary's .hash (tests/golden/make_golden.py --scale commits those under "bench_scale_digests").
_SALTS = (0x243F6A8885A308D3, 0x13198A2E03707344)


def _mix64(x):
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def checksum_words(words, first_index, domain, slice_words=1 << 22):
    """[sum over i of mix(words[i] ^ mix(first_index + i + salt + domain))] for the two salts, mod 2^64 (slices of 2^22 words on a few
    threads: numpy releases the GIL, and the 3 Gb workload has 2 x 10^9 records to go through)"""
    w = np.ascontiguousarray(words, dtype=np.uint64)

    def one(a):
        part = w[a: a + slice_words]
        res = []
        with np.errstate(over="ignore"):
            for salt in _SALTS:
                idx = np.arange(part.size, dtype=np.uint64) + np.uint64((first_index + a + salt + domain * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
                res.append(int(_mix64(part ^ _mix64(idx)).sum(dtype=np.uint64)))
        return res
    starts = list(range(0, w.size, slice_words))
    if len(starts) > 4:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
            parts = list(ex.map(one, starts))
    else:
        parts = [one(a) for a in starts]
    return [sum(p[k] for p in parts) & 0xFFFFFFFFFFFFFFFF for k in range(2)]


def checksum_state(block_bytes, first_block, clushash_bytes, first_entry):
    b = np.frombuffer(bytes(block_bytes), dtype=np.uint64).copy() if not isinstance(block_bytes, np.ndarray) else block_bytes.view(np.uint64).copy()
    b[2::4] = 0                                             # ClusterBlock.clusHash: a heap pointer in the reference's file
    c = np.frombuffer(bytes(clushash_bytes), dtype=np.uint64) if not isinstance(clushash_bytes, np.ndarray) else clushash_bytes.view(np.uint64)
    x, y = checksum_words(b, 4 * first_block, 1), checksum_words(c, first_entry, 2)
    return [(x[0] + y[0]) & 0xFFFFFFFFFFFFFFFF, (x[1] + y[1]) & 0xFFFFFFFFFFFFFFFF]


def sharded_state_checksum(h, rank):
    """this rank's share of checksum_state over the whole data set, summed over the ranks (collective)"""
    mine = [0, 0]
    for sg in h.shard_segments():
        if sg["rank"] != rank:
            continue
        ls, cnt, gb = sg["localStart"], sg["count"], sg["globalBase"]
        if gb == 0:                                          # slot 0 of the block array is nobody's block (hash10x.c:200): left out on both sides
            ls, cnt, gb = ls + 1, cnt - 1, 1
        blocks = h.export_slice(3, ls, cnt) if cnt > 0 else np.zeros(0, dtype=np.uint8)
        for a in range(0, max(sg["entries"], 1), 1 << 25):   # ClusterHash records 256 MB at a time
            n = min(1 << 25, sg["entries"] - a)
            ch = h.export_slice(4, sg["localEntryStart"] + a, n) if n > 0 else np.zeros(0, dtype=np.uint8)
            part = checksum_state(blocks if a == 0 else np.zeros(0, dtype=np.uint8), gb, ch, sg["globalEntryStart"] + a)
            mine = [(mine[0] + part[0]) & 0xFFFFFFFFFFFFFFFF, (mine[1] + part[1]) & 0xFFFFFFFFFFFFFFFF]
    return h.shard_allreduce_sum_u64(mine)


def scaled_workload(wl, world):
    """weak scaling: `world` times the yeast-scale set (pairs, barcodes, genome; table bits grow with log2)"""
    w = dict(wl)
    w["pairs"] *= world; w["barcodes"] *= world; w["genome"] *= world
    w["B"] += (world - 1).bit_length()
    return w


def rendezvous_unique_id(rank, world, hash10x_amd):
    """RCCL bootstrap without torch in the process (torch ships its own HIP/RCCL runtimes; a second runtime next to
    libh10x_hip's cost ~3 ms per step at N = 1): rank 0 draws the ncclUniqueId and serves "H10X" + its 128 bytes on
    the first free port of MASTER_PORT + 29 .. + 44 at MASTER_ADDR; the other ranks walk the same ports until one
    answers with the magic."""
    import socket
    if world == 1:
        return hash10x_amd.Comm.unique_id()
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    if addr == "localhost":
        addr = "127.0.0.1"
    ports = [int(os.environ.get("MASTER_PORT", "29500")) + 29 + i for i in range(16)]
    if rank == 0:
        uid = hash10x_amd.Comm.unique_id()
        srv = None
        for port in ports:
            try:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                srv.bind((addr, port))
                break
            except OSError:
                srv.close()
                srv = None
        if srv is None:
            raise RuntimeError("rendezvous: no free port in %r" % ports)
        srv.listen(world)
        srv.settimeout(900)
        for _ in range(world - 1):
            c, _a = srv.accept()
            c.sendall(b"H10X" + uid)
            c.close()
        srv.close()
        return uid
    deadline = time.time() + 900
    while time.time() < deadline:
        for port in ports:
            try:
                c = socket.create_connection((addr, port), timeout=5)
            except OSError:
                continue
            try:
                c.settimeout(20)
                buf = b""
                while len(buf) < 132:
                    chunk = c.recv(132 - len(buf))
                    if not chunk:
                        break
                    buf += chunk
            except OSError:
                buf = b""
            finally:
                c.close()
            if len(buf) == 132 and buf[:4] == b"H10X":
                return buf[4:]
        time.sleep(0.2)
    raise RuntimeError("rendezvous: rank 0 did not answer on %s:%r" % (addr, ports))


# K1's ceiling is the integer multiplier, not HBM: hashFunc (seqhash.c:58-59) is a 64 x 64 -> low 64 multiply = 1 v_mad_u64_u32 + 2 v_mul_lo_u32
# per k-mer and strand. scratch/valu_rate64.hip measured their issue cost on this chip at 4.9 and 4.2 SIMD cycles per wave instruction
# (plain 32-bit ops: 2), so a multiply-ONLY kernel on 1024 SIMDs at 2.4 GHz would do 64 lanes x 1024 x 2.4e9 / (4.9 + 2 x 4.2) per second.
INT_PEAK_U64_MUL_PER_S = 64 * 1024 * 2.4e9 / (4.9 + 2 * 4.2)


def mosh_int_ops(kmers, ms_per_step):
    if not ms_per_step:
        return {"u64_multiplies_per_s": None, "kmers_per_step": kmers}
    rate = 2.0 * kmers / (ms_per_step * 1e-3)
    return {"u64_multiplies_per_s": rate, "kmers_per_step": kmers, "int_peak_u64_multiplies_per_s": INT_PEAK_U64_MUL_PER_S, "frac_of_int_peak": rate / INT_PEAK_U64_MUL_PER_S,
            "what_is_counted": "hashFunc evaluations (seqhash.c:58-59: one 64-bit multiply each, two per k-mer) per second. Since round 3 the kernel does not "
                               "perform them as multiplies: the products of consecutive k-mers roll (f(j+1) F = 4 f(j) F + a term of the two bases that "
                               "change, from a 16-entry LDS table; two real multiplies per lane and read pair) - same values mod 2^64, so this is the rate "
                               "an implementation that multiplies would need",
            "int_peak_derivation": "64 lanes x 1024 SIMDs x 2.4 GHz / (4.9 + 2 x 4.2) SIMD cycles per u64 multiply (v_mad_u64_u32 + 2 v_mul_lo_u32, rates measured by scratch/valu_rate64.hip); "
                                   "the kernel issues ~35 vector + 23 scalar instructions per 64 k-mer slots all told (profiles/r3d_pmc_sq.json; DESIGN 3, K1); its vector units are busy 61 % of the time by the 4-cycle accounting of SQ_ACTIVE_INST_VALU"}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(wl, recs, workdir, gpu_hash_path):
    """The reference binary (oracle/_ref, compiled -O3 from /root/reference) — or the oracle port where it is absent — on
    this box's host cores, same input, same commands, timed by WALL CLOCK (the reference's own lines are getrusage CPU
    seconds): SURVEY 8d's two invocations, `--readFQB .. --writeHash` and `--readHash .. --hashDepthRange ..
    --cluster 1 0 --writeHash`, each bracketed with perf_counter; once single-threaded and once with the OpenMP build on all
    cores. Returns (cpu_baseline, cpu_baseline_omp, parity string)."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import orc
    fqb = os.path.join(workdir, "bench.fqb")
    recs.tofile(fqb)
    pairs = recs.size // 30
    gpu_canon = open(gpu_hash_path, "rb").read()
    cores = os.cpu_count() or 1
    model = cpu_model()
    if orc.have_ref():
        def two_runs(binary, threads):
            t0 = time.perf_counter()
            r = orc.run_ref(["-B", wl["B"], "--readFQB", "bench.fqb", "--writeHash", "a.hash"], workdir, binary=binary, timeout=3000)
            t1 = time.perf_counter()
            if r.returncode != 0:
                raise RuntimeError("reference failed: " + r.stderr.decode())
            r = orc.run_ref((["-t", threads] if threads > 1 else []) + ["-B", wl["B"], "-ct", wl["ct"], "--readHash", "a.hash", "--hashDepthRange", wl["lo"], wl["hi"],
                            "--cluster", 1, 0, "--writeHash", "ref.hash"], workdir, binary=binary, timeout=3000)
            t2 = time.perf_counter()
            if r.returncode != 0:
                raise RuntimeError("reference failed: " + r.stderr.decode())
            # the reference's own per-command CPU seconds, to split the second process into load / range / cluster
            cmd, tm = None, {}
            for line in r.stdout.decode().splitlines():
                if line.startswith("COMMAND "):
                    cmd = line.split()[1]
                elif line.strip().startswith("user") and cmd:
                    f = line.split()
                    tm[cmd] = tm.get(cmd, 0.0) + float(f[1]) + float(f[3])
            os.remove(os.path.join(workdir, "a.hash"))
            return t1 - t0, t2 - t1, tm
        w_read, w_clu, tm = two_runs("hash10x", 1)
        ref_canon = orc.canonical_hash_bytes(open(os.path.join(workdir, "ref.hash"), "rb").read())
        parity = "identical" if ref_canon == gpu_canon else "DIFFERENT: " + orc.describe_diff(gpu_canon, ref_canon)
        os.remove(os.path.join(workdir, "ref.hash"))
        cpu_clu = tm.get("--cluster", 0.0)                    # CPU seconds = wall for one thread
        one = {"value": pairs / (w_read + w_clu), "unit": "read-pairs/s", "cores": 1, "kind": "reference", "cpu_model": model, "timer": "wall clock (perf_counter around each process)",
               "sample": "full workload (%d read pairs): reference hash10x -O3, 1 thread; process 1 (--readFQB --writeHash) %.2fs, process 2 (--readHash --hashDepthRange --cluster 1 0 "
                         "--writeHash) %.2fs of which --cluster %.2fs by the reference's own timer" % (pairs, w_read, w_clu, cpu_clu),
               "read_pairs_per_s_hashed": pairs / w_read, "barcodes_per_s_clustered": (wl["barcodes"] / cpu_clu) if cpu_clu else None,
               "cluster_seconds": cpu_clu, "second_process_seconds": w_clu}
        omp = None
        if os.path.exists(os.path.join(orc.REF_DIR, "hash10x_omp")):
            o_read, o_clu, otm = two_runs("hash10x_omp", cores)
            omp_canon = orc.canonical_hash_bytes(open(os.path.join(workdir, "ref.hash"), "rb").read())
            os.remove(os.path.join(workdir, "ref.hash"))
            # wall seconds of --cluster alone = second process minus what its other commands took single-threaded above
            other = w_clu - cpu_clu
            omp = {"value": pairs / (o_read + o_clu), "unit": "read-pairs/s", "cores": cores, "kind": "reference", "cpu_model": model, "timer": "wall clock (perf_counter around each process)",
                   "sample": "full workload: reference hash10x -O3 -DOMP -fopenmp, -t %d; process 1 %.2fs (readFQB is serial in the reference), process 2 %.2fs "
                             "(of which ~%.2fs are --readHash / fillHashTable / goodHashes / --writeHash, serial)" % (cores, o_read, o_clu, other),
                   "read_pairs_per_s_hashed": pairs / o_read, "second_process_seconds": o_clu,
                   "cluster_seconds_estimate": max(o_clu - other, 1e-9), "parity_with_1_thread": "identical" if omp_canon == ref_canon else "DIFFERENT"}
        # end to end as a user runs it: the C program (bin/hash10x-amd), file in, file out, one process — next to the reference's two
        e2e = None
        exe = os.path.join(REPO, "bin", "hash10x-amd")
        if os.path.exists(exe):
            cmd = [exe, "-B", str(wl["B"]), "-ct", str(wl["ct"]), "--readFQB", "bench.fqb", "--hashDepthRange", str(wl["lo"]), str(wl["hi"]), "--cluster", "1", "0", "--writeHash", "cli.hash"]
            best = None
            for _ in range(4):                               # best of four: from the second on the file is in the page cache, like the reference's runs above (creating the HIP context alone varies 70-260 ms between runs)
                if os.path.exists(os.path.join(workdir, "cli.hash")):
                    os.remove(os.path.join(workdir, "cli.hash"))   # every run writes a NEW file (truncating the previous 250 MB and ext4's flush at close of a replaced file cost 50 ms)
                t0 = time.perf_counter()
                g = subprocess.run(cmd, cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                dt = time.perf_counter() - t0
                if g.returncode != 0:
                    raise RuntimeError("hash10x-amd failed: " + g.stderr.decode()[-300:])
                if best is None or dt < best:
                    best, gbest = dt, g
            walls = {}
            cmdname = None
            for line in gbest.stdout.decode().splitlines():  # the program's own per-command wall seconds (its resource lines), of the best run
                if line.startswith("COMMAND "):
                    cmdname = line.split()[1]
                elif line.strip().startswith("user") and cmdname and "wall" in line:
                    walls[cmdname] = walls.get(cmdname, 0.0) + float(line.split()[-1])
            cli_same = open(os.path.join(workdir, "cli.hash"), "rb").read() == gpu_canon
            os.remove(os.path.join(workdir, "cli.hash"))
            e2e = {"command": "hash10x-amd -B %d --readFQB bench.fqb --hashDepthRange %d %d --cluster 1 0 --writeHash cli.hash" % (wl["B"], wl["lo"], wl["hi"]),
                   "wall_seconds": best, "read_pairs_per_s": pairs / best, "includes": "process start, HIP context, file read (page cache: 16 reader threads on a mapping) beside H2D from six 64 MiB page-locked slabs, all kernels, D2H + file write of the .hash",
                   "per_command_wall_seconds": walls, "reference_two_processes_wall_seconds": w_read + w_clu, "speedup_vs_reference_1thread": (w_read + w_clu) / best,
                   "reference_omp_wall_seconds": (o_read + o_clu) if omp else None, "hash_identical_to_library_path": cli_same}
        one["end_to_end"] = e2e
        return one, omp, parity
    o = orc.Oracle(B=wl["B"])
    t0 = time.perf_counter(); o.read_fqb(recs); t1 = time.perf_counter()
    o.depth_range(wl["lo"], wl["hi"]); t2 = time.perf_counter()
    o.cluster(1, 0, wl["ct"], 1); t3 = time.perf_counter()
    o.write_hash(os.path.join(workdir, "orc.hash"))
    ref_canon = open(os.path.join(workdir, "orc.hash"), "rb").read()
    parity = "identical" if ref_canon == gpu_canon else "DIFFERENT: " + orc.describe_diff(gpu_canon, ref_canon)
    one = {"value": pairs / (t3 - t0), "unit": "read-pairs/s", "cores": 1, "kind": "port", "cpu_model": model, "timer": "wall clock",
           "sample": "full workload (%d read pairs): oracle C restatement, 1 thread; readFQB %.2fs + hashDepthRange %.2fs + cluster %.2fs" % (pairs, t1 - t0, t2 - t1, t3 - t2),
           "read_pairs_per_s_hashed": pairs / (t1 - t0), "barcodes_per_s_clustered": wl["barcodes"] / (t3 - t2), "cluster_seconds": t3 - t2}
    o2 = orc.Oracle(B=wl["B"]); o2.read_fqb(recs); o2.depth_range(wl["lo"], wl["hi"])
    t4 = time.perf_counter(); o2.cluster(1, 0, wl["ct"], cores); t5 = time.perf_counter()
    omp = {"value": pairs / ((t2 - t0) + (t5 - t4)), "unit": "read-pairs/s", "cores": cores, "kind": "port", "cpu_model": model, "timer": "wall clock",
           "sample": "oracle C restatement, OpenMP over barcodes with %d threads in --cluster (%.2fs); readFQB / hashDepthRange serial as above" % (cores, t5 - t4),
           "cluster_seconds_estimate": t5 - t4}
    return one, omp, parity


def full_config3_block(hash10x_amd, local_rank, steps=3):
    """BASELINE configs[2] AT ITS OWN SIZE on this GPU: 200 M read pairs, 1 M barcodes, 500 Mb x 2, e = 0.1 %, -B 29 (the table size the reference
    accepts for it), --hashDepthRange 30 100 --cluster 1 0 — the set whose whole .hash is pinned by the reference binary's sha256 in the GPU tests
    (test_config3_full_size_matches_reference_digest). Generated here by gen_fqb v2 (OpenMP); 24 GB of records resident in HBM."""
    man = json.load(open(os.path.join(REPO, "tests", "golden", "manifest.json"))).get("full_digest_cases", [])
    if not man:
        return {"skipped": "no full_digest_cases in tests/golden/manifest.json"}
    case = man[0]; g = case["gen2"]
    wl = dict(pairs=g["pairs"], barcodes=g["barcodes"], genome=g["genome"], err=g["err"], mol=g["mol"], snp=g["snp"], mol_len=g["mol_len"])
    t0 = time.perf_counter()
    recs, _first, _total = generate_v2(wl, g["seed"])
    gen_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    d = hash10x_amd.DeviceRecords(recs, device=local_rank)
    up_s = time.perf_counter() - t0
    pairs = recs.size // 30
    e2e = None
    try:                                                     # twice: the first run of the program on a box also pages in the libraries and the device code (6.3-7.9 s); the second is what a user sees from then on
        first = cli_end_to_end(recs, case["B"], 30, 100, 5, expect_sha256=case["sha256"], expect_size=case["size"])
        e2e = cli_end_to_end(recs, case["B"], 30, 100, 5, expect_sha256=case["sha256"], expect_size=case["size"])
        if isinstance(first, dict) and isinstance(e2e, dict) and "wall_seconds" in first and "wall_seconds" in e2e:
            e2e["first_run_wall_seconds"] = first["wall_seconds"]
            e2e["hash_identical_to_reference"] = bool(first.get("hash_identical_to_reference")) and bool(e2e.get("hash_identical_to_reference"))
            if first["wall_seconds"] < e2e["wall_seconds"]:
                first["first_run_wall_seconds"] = first["wall_seconds"]; first["hash_identical_to_reference"] = e2e["hash_identical_to_reference"]; e2e = first
    except Exception as e:
        e2e = {"error": str(e)[:300]}
    del recs
    h = hash10x_amd.Hash10x(B=case["B"], device=local_rank)
    h.enable_timing(True)
    wall, tm = [], {}
    for it in range(steps + 1):                             # one warm-up
        hash10x_amd.synchronize(local_rank)
        t = time.perf_counter()
        h.read_fqb_device(d.ptr, pairs); h.depth_range(30, 100); h.cluster(1, 0, 5)
        hash10x_amd.synchronize(local_rank)
        if it:
            wall.append(time.perf_counter() - t)
            for k, (ms, n) in h.timings().items():
                a = tm.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += n
    c = h.counters(); z = h.sizes()
    clu_ms = tm["cluster_kernel"][0] / steps
    alg = 4.0 * c["sum_good_depth"] + 14.0 * c["sum_good"] + 16.0 * c["sum_hash_clustered"]
    try:
        tr = profile_traffic("*config3full_pmc_traffic.json", "cluster_kernel")
    except Exception:
        tr = None
    out = {"workload": "config3-full-200M (BASELINE configs[2] at its own size: 200 M pairs, 1 M barcodes, 500 Mb x 2, e = 0.1 %%, -B 29; gen_fqb v2 seed %d)" % g["seed"],
           "read_pairs": pairs, "B": case["B"], "ms_per_step": 1e3 * sum(wall) / len(wall), "read_pairs_per_s": pairs * len(wall) / sum(wall), "steps": steps,
           "device_ms_per_step": {k: round(v[0] / steps, 2) for k, v in tm.items() if v[0] > 0},
           "entries_H": c["entries"], "distinct_U": c["distinct"], "hashNumber": z["hashNumber"],
           "sizes_match_reference": z["hashNumber"] == case["hash_number"] and z["nBlocks"] == case["blocks_max"] and z["nClusHash"] == case["sum_nHash"],
           "first_placement": {0: "dense", 1: "ranked", 2: "hbm-slot", 3: "hashed", 4: "translated"}.get(c["cluster_first_mode"]),
           "cluster_class_counts": c["cluster_class_counts"], "cluster_overflow_blocks": c["cluster_overflow_blocks"],
           "roofline": {"bound": "hbm", "kernel": "all cluster launches", "achieved": alg / (clu_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg / (clu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": alg, "ms_per_step": clu_ms},
           "barcodes_per_s_clustered": wl["barcodes"] / (tm["cluster"][0] / steps * 1e-3),
           "traffic": tr["traffic"] if tr else None, "traffic_of": "all cluster_kernel launches at full size", "traffic_raw": tr, "traffic_note": TRAFFIC_NOTE if tr else None,
           "traffic_stale": (tr["traffic_build_id"] != hash10x_amd.build_id()) if tr else None,
           "generate_seconds": round(gen_s, 1), "upload_seconds": round(up_s, 2), "host_threads": os.cpu_count(), "end_to_end": e2e,
           "parity": "tests/test_gpu_parity.py::test_config3_full_size_matches_reference_digest: the whole 15 GB .hash of this set has the reference binary's sha256"}
    h.close(); d.free()
    return out


def cli_end_to_end(recs, B, lo, hi, ct, expect_sha256=None, expect_size=None):
    """File in, file out, one process, as a user runs it: bin/hash10x-amd -B b --readFQB x.fqb --hashDepthRange lo hi --cluster 1 0 --writeHash x.hash on a .fqb in memory-backed
    storage (/dev/shm where it has room: the disk of a GPU box is not what is measured), per-command wall seconds from the program's own resource lines, and the sha256 of
    the canonical .hash it wrote against the reference binary's (manifest)."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import orc
    import shutil
    exe = os.path.join(REPO, "bin", "hash10x-amd")
    if not os.path.exists(exe):
        return {"skipped": "bin/hash10x-amd not built"}
    need = recs.nbytes + (expect_size or recs.nbytes) + (2 << 30)
    base = None
    for cand in ("/dev/shm", tempfile.gettempdir()):
        try:
            st = os.statvfs(cand)
            if st.f_bavail * st.f_frsize > need:
                base = cand
                break
        except OSError:
            pass
    if base is None:
        return {"skipped": "no room for %.0f GB of files" % (need / 1e9)}
    d = tempfile.mkdtemp(prefix="h10x_e2e_", dir=base)
    try:
        t0 = time.perf_counter(); recs.tofile(os.path.join(d, "x.fqb")); write_in_s = time.perf_counter() - t0
        cmd = [exe, "-B", str(B), "-ct", str(ct), "--readFQB", "x.fqb", "--hashDepthRange", str(lo), str(hi), "--cluster", "1", "0", "--writeHash", "x.hash"]
        t0 = time.perf_counter()
        g = subprocess.run(cmd, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        wall = time.perf_counter() - t0
        if g.returncode != 0:
            raise RuntimeError("hash10x-amd failed: " + g.stderr.decode()[-300:])
        walls, cmdname = {}, None
        for line in g.stdout.decode().splitlines():
            if line.startswith("COMMAND "):
                cmdname = line.split()[1]
            elif line.strip().startswith("user") and cmdname and "wall" in line:
                walls[cmdname] = walls.get(cmdname, 0.0) + float(line.split()[-1])
        t0 = time.perf_counter()
        digest, info = orc.canonical_file_digest(os.path.join(d, "x.hash"))
        digest_s = time.perf_counter() - t0
        pairs = recs.size // 30
        io = walls.get("--readFQB", 0.0) + walls.get("--writeHash", 0.0)
        return {"command": " ".join(["hash10x-amd"] + cmd[1:]), "storage": base, "fqb_bytes": int(recs.nbytes), "hash_bytes": info["size"], "wall_seconds": wall, "read_pairs_per_s": pairs / wall,
                "per_command_wall_seconds": walls, "io_commands_share": (io / wall) if wall else None,
                "io_commands_note": "--readFQB (file read beside the uploads, hashing and index build) + --writeHash (device -> host beside the file writes) as a share of the process's wall time",
                "hash_sha256": digest, "hash_identical_to_reference": (digest == expect_sha256 and info["size"] == expect_size) if expect_sha256 else None,
                "host_timing_lines": [ln.strip() for ln in g.stderr.decode(errors="replace").splitlines() if ln.strip().startswith(("ingest of", "hostprof:"))][-6:] or None,   # (H10X_INGEST_TIMING / H10X_HOSTPROF in the environment)
                "write_input_seconds": round(write_in_s, 1), "digest_seconds": round(digest_s, 1)}
    finally:
        shutil.rmtree(d, ignore_errors=True)


TRAFFIC_NOTE = ("HBM bytes per step from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in passes of their own), corrected as MI355X_MICROARCH.md prescribes for gfx950: "
                "FETCH_SIZE doubled — calibrated on this kernel's own access widths by scratch/cal_widths.hip (4-byte gathers of 256-byte pieces and contiguous 2-byte loads both read 1/2 of the "
                "bytes fetched; a 256-byte piece at a random 4-byte alignment fetches 1.48 x its bytes) — WRITE_SIZE as it is (2-byte and 8-byte stores read exactly)")


def profile_traffic(pattern, match, exclude=None):
    """(corrected bytes per step, raw FETCH_SIZE, raw WRITE_SIZE, file, head, build id) of the kernels whose name contains `match` in the newest profiles/<pattern>"""
    import glob
    excl = (exclude,) if isinstance(exclude, str) else tuple(exclude or ())
    cands = sorted(f for f in glob.glob(os.path.join(REPO, "profiles", pattern)) if not any(x in os.path.basename(f) for x in excl))
    if not cands:
        return None
    pm = json.load(open(cands[-1]))
    fetch = sum(v.get("FETCH_SIZE", {}).get("bytes_per_step", 0.0) for k, v in pm["kernels"].items() if match in k)
    write = sum(v.get("WRITE_SIZE", {}).get("bytes_per_step", 0.0) for k, v in pm["kernels"].items() if match in k)
    if not fetch and not write:
        return None
    return {"traffic": 2.0 * fetch + write, "raw_fetch_size": fetch, "raw_write_size": write, "traffic_source": os.path.basename(cands[-1]),
            "traffic_head": pm.get("head"), "traffic_build_id": pm.get("build_id")}


def genome3g_block(hash10x_amd, local_rank, steps=2, name="genome3g-300M"):
    """The 3 Gb workload (BASELINE configs[3] shape at the size the reference's run could be pinned: WORKLOADS["genome3g-300M"]) on ONE GPU — the set
    `bench.py --workload genome3g-300M --scaling strong --gpus N` scales over 1 / 2 / 4 / 8 ranks: ms per step, roofline of the cluster launches, and the
    parity gate against the reference binary's digest (composable checksum of all blocks and ClusterHash records)."""
    wl = dict(WORKLOADS[name])
    t0 = time.perf_counter()
    recs, _first, _total = generate_v2(wl, wl["seed"])
    gen_s = time.perf_counter() - t0
    d = hash10x_amd.DeviceRecords(recs, device=local_rank)
    pairs = recs.size // 30
    del recs
    h = hash10x_amd.Hash10x(B=wl["B"], device=local_rank)
    h.enable_timing(True)
    wall, tm = [], {}
    for it in range(steps + 1):                             # one warm-up
        hash10x_amd.synchronize(local_rank)
        t = time.perf_counter()
        h.read_fqb_device(d.ptr, pairs); h.depth_range(wl["lo"], wl["hi"]); h.cluster(1, 0, wl["ct"])
        hash10x_amd.synchronize(local_rank)
        if it:
            wall.append(time.perf_counter() - t)
            for k, (ms, n) in h.timings().items():
                a = tm.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += n
    c = h.counters(); z = h.sizes()
    clu_ms = tm["cluster_kernel"][0] / steps
    alg = 4.0 * c["sum_good_depth"] + 14.0 * c["sum_good"] + 16.0 * c["sum_hash_clustered"]
    out = {"workload": "%s (%.1f Gb x 2 haplotypes, %d M pairs, %d k barcodes, e = 0.05 %%, -B %d, --hashDepthRange %d %d; gen_fqb v2 seed %d)" % (name, wl["genome"] / 1e9, wl["pairs"] // 1000000, wl["barcodes"] // 1000, wl["B"], wl["lo"], wl["hi"], wl["seed"]),
           "read_pairs": pairs, "B": wl["B"], "ms_per_step": 1e3 * sum(wall) / len(wall), "read_pairs_per_s": pairs * len(wall) / sum(wall), "steps": steps,
           "device_ms_per_step": {k: round(v[0] / steps, 2) for k, v in tm.items() if v[0] > 0},
           "entries_H": c["entries"], "distinct_U": c["distinct"], "hashNumber": z["hashNumber"],
           "first_placement": {0: "dense", 1: "ranked", 2: "hbm-slot", 3: "hashed", 4: "translated"}.get(c["cluster_first_mode"]),
           "cluster_class_counts": c["cluster_class_counts"], "cluster_overflow_blocks": c["cluster_overflow_blocks"],
           "roofline": {"bound": "hbm", "kernel": "all cluster launches", "achieved": alg / (clu_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg / (clu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": alg, "ms_per_step": clu_ms},
           "barcodes_per_s_clustered": wl["barcodes"] / (tm["cluster"][0] / steps * 1e-3),
           "generate_seconds": round(gen_s, 1), "host_threads": os.cpu_count()}
    try:
        tr = profile_traffic("*genome3g_pmc_traffic.json", "cluster_kernel")
    except Exception:
        tr = None
    out["traffic"] = tr["traffic"] if tr else None
    out["traffic_of"] = "all cluster_kernel launches of a step"
    out["traffic_raw"] = tr
    out["traffic_stale"] = (tr["traffic_build_id"] != hash10x_amd.build_id()) if tr else None
    man = json.load(open(os.path.join(REPO, "tests", "golden", "manifest.json")))

    def state_checksum():
        zz = h.sizes()
        c2 = checksum_state(h.export_slice(3, 1, zz["nBlocks"] - 1), 1, np.zeros(0, dtype=np.uint8), 0)
        for a in range(0, zz["nClusHash"], 1 << 27):
            part = checksum_state(np.zeros(0, dtype=np.uint8), 0, h.export_slice(4, a, min(1 << 27, zz["nClusHash"] - a)), a)
            c2 = [(c2[0] + part[0]) & 0xFFFFFFFFFFFFFFFF, (c2[1] + part[1]) & 0xFFFFFFFFFFFFFFFF]
        return c2
    try:
        t0 = time.perf_counter()
        cs = state_checksum()
        exp = man.get("strong_digests", {}).get(name)
        out["state_checksum"] = ["0x%016x" % v for v in cs]
        out["parity_seconds"] = round(time.perf_counter() - t0, 1)
        if exp is None:
            out["parity_vs_reference_digest"] = "no reference digest committed for %s" % name
        else:
            same = [int(v, 16) for v in exp["checksum"]] == cs and exp["hash_number"] == z["hashNumber"] and exp["sum_nHash"] == z["nClusHash"]
            out["parity_vs_reference_digest"] = "identical" if same else "DIFFERENT (reference %s, H %d, hashNumber %d)" % (exp["checksum"], exp["sum_nHash"], exp["hash_number"])
    except Exception as e:
        out["parity_vs_reference_digest"] = "check failed: " + str(e)[:200]
    # BASELINE configs[4] on the same state: the crib (both truth haplotypes, 3 Gb each, hashed and looked up on the device), the report's figures over all barcodes
    # (the accuracy check: clusters whose located hashes all lie on one chromosome, mean span), then --clusterSplit; HBM high-water mark along the way
    try:
        total = hash10x_amd.device_mem_info(local_rank)[1]
        low = [hash10x_amd.device_mem_info(local_rank)[0]]
        t0 = time.perf_counter(); hap_a, hap_b = haplotypes_v2(wl, wl["seed"]); hap_s = time.perf_counter() - t0
        t0 = time.perf_counter(); known = h.crib_genomes(hap_a, hap_b); hash10x_amd.synchronize(local_rank); crib_s = time.perf_counter() - t0
        del hap_a, hap_b
        low.append(hash10x_amd.device_mem_info(local_rank)[0])
        t0 = time.perf_counter(); fig = h.cluster_report_figures(); rep_s = time.perf_counter() - t0
        low.append(hash10x_amd.device_mem_info(local_rank)[0])
        t0 = time.perf_counter(); h.cluster_split(); hash10x_amd.synchronize(local_rank); split_s = time.perf_counter() - t0
        low.append(hash10x_amd.device_mem_info(local_rank)[0])
        z2 = h.sizes()
        exp4 = man.get("split_digests", {}).get(name, {})
        c4 = {"commands": "--cribBuild A.fa B.fa (through h10x_crib_genome: no FASTA files) --clusterReport 1 0 (figures, no text) --clusterSplit, on the clustered state of the step above",
              "cribBuild_seconds": round(crib_s, 3), "cribBuild_note": "both haplotypes (2 x %.1f Gb of base codes from host memory: upload + mosh extraction + look-ups + classification)" % (wl["genome"] / 1e9),
              "crib_known_unknown_moshes": known, "haplotype_generation_seconds": round(hap_s, 1),
              "clusterReport_seconds": round(rep_s, 3), "clusterReport_note": "h10x_cluster_report over all %d barcodes: per-barcode and per-cluster figures reduced on the device, 80 B per cluster over PCIe, summed on the host" % (z["nBlocks"] - 1),
              "clusterSplit_seconds": round(split_s, 3), "blocks_after_split": z2["nBlocks"],
              "accuracy": {"clusters": fig["clusters"], "clusters_without_OTHER": fig["clusters_without_OTHER"], "purity": fig["clusters_without_OTHER"] / fig["clusters"] if fig["clusters"] else None,
                           "clusters_located": fig["clusters_located"], "mean_span_kb": fig["sum_span"] / fig["clusters_located"] if fig["clusters_located"] else None,
                           "mean_reads_per_cluster": fig["sum_reads"] / fig["clusters"] if fig["clusters"] else None, "mean_hashes_per_cluster": fig["sum_hashes"] / fig["clusters"] if fig["clusters"] else None,
                           "what": "purity = share of CODE_CLUSTER lines without an OTHER list (every located hash of the cluster on one chromosome, hash10x.c:905-913); span in crib position units (pos >> 10)"},
              "hbm_high_water_GB": round((total - min(low)) / 1e9, 1), "hbm_total_GB": round(total / 1e9, 1)}
        if "report" in exp4:
            same = all(fig[k] == exp4["report"][k] for k in ("clusters", "clusters_without_OTHER", "clusters_located", "sum_span", "sum_reads", "sum_hashes"))
            c4["accuracy_vs_reference_report"] = "identical" if same else "DIFFERENT (reference %r)" % {k: exp4["report"][k] for k in ("clusters", "clusters_without_OTHER", "clusters_located", "sum_span", "sum_reads", "sum_hashes")}
        else:
            c4["accuracy_vs_reference_report"] = exp4.get("failed", "no reference report for this set in the manifest") if exp4 else "no reference report for this set in the manifest"
        if "no_split_digest" in exp4:
            c4["split_state_vs_reference_digest"] = "none: " + exp4["no_split_digest"]
        if "checksum" in exp4:
            cs4 = state_checksum()
            c4["split_state_vs_reference_digest"] = "identical" if [int(v, 16) for v in exp4["checksum"]] == cs4 and exp4["blocks_max"] == z2["nBlocks"] else "DIFFERENT"
        out["config5"] = c4
    except Exception as e:
        out["config5"] = {"error": str(e)[:300]}
    h.close(); d.free()
    return out


XGMI_LINK_GBS = 153.0          # per direction and link, 7 links per GPU (MI355X_MICROARCH.md); the model below charges 80 % of it
STAGE_TIMERS = ("block_runs", "mosh_extract", "mosh_fallback", "compact_entries", "sort_by_hash", "index_rank", "probe_table", "clushash_build", "csr_build", "good_hashes", "cluster")


def rank_figures(h):
    """compute and exchange figures of one rank's last step: stage timers (device ms), the exchanges' bytes and waits, compute = stages less the waits inside them"""
    tm = {k: v[0] for k, v in h.timings().items() if v[0] > 0}
    ex = h.exchanges()
    stages = sum(tm.get(k, 0.0) for k in STAGE_TIMERS)
    waits = sum(e["ms_in_stages"] for e in ex.values())
    sw = h.stage_waits()
    return {"stage_ms": {k: round(v, 3) for k, v in tm.items()}, "stage_compute_ms": {k: round(tm[k] - sw.get(k, 0.0), 3) for k in STAGE_TIMERS if k in tm},
            "exchanges": ex, "compute_ms": stages - waits}


def scaling_model(per_rank, single_ms):
    """A MODEL of the N-GPU step from figures measured on ONE GPU (ranks taking turns): max over ranks of their compute + per kind of exchange the largest per-peer
    share any rank sends, over one xGMI link at 80 % of its rate (the all-to-alls are grouped point-to-point sends: every link of a rank works at once, the busiest
    pair bounds the call), + 30 us per call. An exchange the library queues on its exchange stream BESIDE a stage (h10x_exchange_beside: the in-range lists beside the
    good lists, hashDepth[] beside the ClusterHash records) is charged only with what exceeds that stage's compute on the rank where it is shortest; every other
    exchange in full. Not a measurement."""
    n = len(per_rank)
    comp = [r["compute_ms"] for r in per_rank]
    kinds = sorted({k for r in per_rank for k in r["exchanges"]})
    ex = {}
    xfer_ms = 0.0
    cover = {}                                               # per stage: what is left of its (shortest) compute to hide exchanges behind
    for k in kinds:
        peer = max(r["exchanges"].get(k, {}).get("max_peer_out", 0) for r in per_rank)
        out = [r["exchanges"].get(k, {}).get("bytes_out", 0) for r in per_rank]
        calls = max(r["exchanges"].get(k, {}).get("calls", 0) for r in per_rank)
        ms = peer / (0.8 * XGMI_LINK_GBS * 1e9) * 1e3 + 0.03 * calls
        beside = next((r["exchanges"][k].get("beside") for r in per_rank if k in r["exchanges"] and r["exchanges"][k].get("beside")), None)
        exposed = ms
        if beside:
            if beside not in cover:
                cover[beside] = min(r["stage_compute_ms"].get(beside, 0.0) for r in per_rank)
            hidden = min(ms, cover[beside]); cover[beside] -= hidden; exposed = ms - hidden
        xfer_ms += exposed
        ex[k] = {"calls_per_step": calls, "max_rank_bytes_out": max(out), "sum_bytes_out": sum(out), "busiest_peer_share_bytes": peer, "modelled_ms": round(ms, 3)}
        if beside:
            ex[k]["beside"] = beside; ex[k]["exposed_ms"] = round(exposed, 3)
    step = max(comp) + xfer_ms
    back = [r["exchanges"].get("indices_back (all-to-all)", {}).get("bytes_out", 0) for r in per_rank]      # what an owner answers = the entries it owns
    return {"ranks": n, "max_rank_compute_ms": round(max(comp), 3), "mean_rank_compute_ms": round(sum(comp) / n, 3), "compute_imbalance": round(max(comp) / (sum(comp) / n), 3) if sum(comp) else None,
            "modelled_exchange_ms": round(xfer_ms, 3), "modelled_step_ms": round(step, 3), "single_gpu_step_ms": round(single_ms, 3),
            "modelled_speedup_vs_1_gpu": round(single_ms / step, 2) if step else None, "exchanges": ex,
            "busiest_owner_share": round(max(back) / sum(back), 4) if sum(back) else None,
            "model": "max rank compute (stage timers less the waits inside them, ranks taking turns on one GPU; a rank's fastest timed step) + sum over exchanges of busiest-peer bytes / (0.8 x %.0f GB/s) + 30 us per call; "
                     "an exchange queued beside a stage counts with what exceeds that stage's shortest compute; a MODEL, not a measurement" % XGMI_LINK_GBS}


def virtual_ranks_block(hash10x_amd, name, n, local_rank=0, steps=1):
    """`n` ranks as threads on ONE GPU (in-process communicator, ranks taking turns: h10x_comm_local_serialize), each with its shard of the generator-v2 workload `name`:
    per rank the compute of a step and every exchange's bytes; the same set on one unsharded context for the single-GPU step; scaling_model() on top."""
    import threading
    t_block = time.perf_counter()
    wl = dict(WORKLOADS[name])
    # one GPU, unsharded: the step the speed-up is quoted against
    recs, _f, total = generate_v2(wl, wl["seed"])
    d = hash10x_amd.DeviceRecords(recs, device=local_rank); pairs = recs.size // 30; del recs
    h = hash10x_amd.Hash10x(B=wl["B"], device=local_rank); h.enable_timing(True)
    single = []
    for it in range(steps + 1):
        hash10x_amd.synchronize(local_rank); t = time.perf_counter()
        h.read_fqb_device(d.ptr, pairs); h.depth_range(wl["lo"], wl["hi"]); h.cluster(1, 0, wl["ct"]); hash10x_amd.synchronize(local_rank)
        if it:
            single.append(1e3 * (time.perf_counter() - t))
    single_stage = {k: round(v[0], 3) for k, v in h.timings().items() if v[0] > 0}
    h.close(); d.free()
    comms = hash10x_amd.Comm.local(n)
    comms[0].serialize(True)
    res, err = [None] * n, [None] * n

    def work(r):
        try:
            rr, _first, _tot = generate_v2(wl, wl["seed"], r, n)
            dr = hash10x_amd.DeviceRecords(rr, device=local_rank); np_ = rr.size // 30; del rr
            hh = hash10x_amd.Hash10x(B=wl["B"], device=local_rank); hh.enable_timing(True)
            for kv in filter(None, os.environ.get("H10X_OPTS", "").split(",")):      # tuning knobs for A/B runs of the model (e.g. shard_delta_lists=0)
                hh.set_option(kv.split("=")[0], int(kv.split("=")[1]))
            for it in range(steps + 1):
                comms[r].turn_begin()
                try:
                    hh.shard_read_fqb_device(comms[r], dr.ptr, np_); hh.depth_range(wl["lo"], wl["hi"]); hh.cluster(1, 0, wl["ct"])
                finally:
                    comms[r].turn_end(local_rank)
                if it:                                       # a rank's figure = its FASTEST timed step: eight ranks share ONE GPU's memory here, and a step in which the block cache has to go
                    fig = rank_figures(hh)                   # back to the driver (2 ms in --cluster, another rank every run) is an artefact of that, not of a rank with a GPU to itself
                    if res[r] is None or fig["compute_ms"] < res[r]["compute_ms"]:
                        res[r] = fig
            res[r]["read_pairs"] = np_
            comms[r].turn_begin()
            try:
                hh.shard_barrier()
            finally:
                comms[r].turn_end(local_rank)
            hh.close(); dr.free()
        except Exception as e:                              # a rank that dies leaves the others waiting in a collective: say so and let the caller's timeout end it
            err[r] = str(e)
    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(n)]
    for t in th:
        t.start()
    # a rank that raises outside a collective leaves the others waiting in LocalGroup::wait(): the joins are bounded, the threads daemons, and the caller reports
    # what the dead rank said instead of hanging behind a headline that was already computed
    deadline = time.time() + float(os.environ.get("H10X_VRANK_TIMEOUT", "600"))
    for t in th:
        while t.is_alive() and time.time() < deadline and not any(err):
            t.join(0.5)
    if any(err) or any(t.is_alive() for t in th):
        for t in th:
            t.join(2.0)
        if any(t.is_alive() for t in th):
            globals()["_HUNG_THREADS"] = True
        said = "; ".join("rank %d: %s" % (i, e) for i, e in enumerate(err) if e) or "no rank reported an error"
        return {"error": ("virtual ranks did not finish (%s)" % said)[:400]}
    for c in comms:
        c.destroy()
    out = scaling_model(res, sum(single) / len(single))
    out["workload"] = name
    out["seconds"] = round(time.perf_counter() - t_block, 1)   # what the block cost the bench run (generation of the set and of the eight shards included)
    out["single_gpu_stage_ms"] = single_stage
    out["max_rank_stage_compute_ms"] = {k: max(r["stage_compute_ms"].get(k, 0.0) for r in res) for k in STAGE_TIMERS if any(k in r["stage_compute_ms"] for r in res)}
    out["per_rank"] = [{"read_pairs": r["read_pairs"], "compute_ms": round(r["compute_ms"], 3), "stage_compute_ms": r["stage_compute_ms"],
                        "exchange_wait_ms": round(sum(e["ms"] for e in r["exchanges"].values()), 3)} for r in res]
    return out


def secondary_block(hash10x_amd, local_rank):
    """The 1/10-scale BASELINE configs[2] set in the same process (single GPU): ms per step and the roofline of the main
    cluster launch in the ranked placement — the regime real data sets are in, next to the yeast-scale headline."""
    wl = dict(WORKLOADS["config3-tenth-20M"])
    t0 = time.perf_counter()
    recs = generate(wl, seed=wl["seed"])
    gen_s = time.perf_counter() - t0
    d = hash10x_amd.DeviceRecords(recs, device=local_rank)
    pairs = recs.size // 30
    del recs
    h = hash10x_amd.Hash10x(B=wl["B"], device=local_rank)
    h.enable_timing(True)
    steps, wall, tm = 5, [], {}
    for it in range(steps + 1):                             # one warm-up
        hash10x_amd.synchronize(local_rank)
        t = time.perf_counter()
        h.read_fqb_device(d.ptr, pairs); h.depth_range(wl["lo"], wl["hi"]); h.cluster(1, 0, wl["ct"])
        hash10x_amd.synchronize(local_rank)
        if it:
            wall.append(time.perf_counter() - t)
            for k, (ms, n) in h.timings().items():          # the timers restart with every --readFQB (initialise())
                a = tm.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += n
    c = h.counters(); z = h.sizes()
    main_ms = tm["cluster_main"][0] / max(tm["cluster_main"][1], 1)
    alg = 4.0 * c["cluster_main"][1] + 14.0 * c["cluster_main"][0]        # (what the main launch moves: see the headline's cluster_main)
    clu_all_ms = tm["cluster_kernel"][0] / steps
    alg_all = 4.0 * c["sum_good_depth"] + 14.0 * c["sum_good"] + 16.0 * c["sum_hash_clustered"]
    tr = None                                                # rocprofv3 --pmc passes of this workload, committed under profiles/
    try:
        tr = profile_traffic("*_config3_pmc_traffic.json", "cluster_kernel")
    except Exception:
        tr = None
    out = {"workload": "config3-tenth-20M (BASELINE configs[2] proportions at 1/10: 20 M pairs, 100 k barcodes, 50 Mb x 2, e = 0.1 %, -B 26)",
           "ms_per_step": 1e3 * sum(wall) / len(wall), "read_pairs_per_s": pairs * len(wall) / sum(wall), "steps": steps,
           "device_ms_per_step": {k: round(v[0] / steps, 3) for k, v in tm.items() if v[0] > 0},
           "entries_H": c["entries"], "distinct_U": c["distinct"], "hashNumber": z["hashNumber"],
           "first_placement": {0: "dense", 1: "ranked", 2: "hbm-slot", 3: "hashed", 4: "translated"}.get(c["cluster_first_mode"]),
           "cluster_class_counts": c["cluster_class_counts"], "cluster_overflow_blocks": c["cluster_overflow_blocks"],
           "roofline": {"bound": "hbm", "kernel": "cluster_main", "achieved": alg / (main_ms * 1e-3) / 1e9 if main_ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg / (main_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if main_ms else None, "algorithmic_bytes_per_launch": alg, "avg_launch_ms": main_ms,
                        "barcodes_in_launch": c["cluster_main"][3],
                        "all_cluster_launches": {"GB/s": alg_all / (clu_all_ms * 1e-3) / 1e9 if clu_all_ms else None, "ms_per_step": clu_all_ms, "algorithmic_bytes": alg_all}},
           "generate_seconds": gen_s, "traffic": tr["traffic"] if tr else None, "traffic_of": "all cluster_kernel launches (compare with all_cluster_launches.algorithmic_bytes less 16 H)",
           "traffic_raw": tr, "traffic_note": TRAFFIC_NOTE if tr else None,
           "traffic_stale": (tr["traffic_build_id"] != hash10x_amd.build_id()) if tr else None,
           "parity": "tests/test_gpu_parity.py::test_config3_proportions_match_reference_digests pins this exact set (sha256 of the reference binary's .hash)"}
    h.close(); d.free()
    return out


_HUNG_THREADS = False        # set when virtual-rank threads were left behind in a collective: main() then leaves through os._exit
HEADLINE_MAX_BYTES = 6000      # the driver keeps the LAST stdout line and parses it; round 5's 21 KB line did not fit its capture


def _r(x, nd=4):
    """numbers to `nd` significant digits (the detail file keeps them whole)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float("%.*g" % (nd + 2, x))
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def step_bytes(pairs, H, U, B, cluster_bytes):
    """ALL of SURVEY 8d's algorithmic bytes of one step: hashing 120 P + 16 H, index build 28 H + 16 U + 4 2^B, clustering 4 sum d + 14 sum G + 16 H_clustered"""
    return 120.0 * pairs + 16.0 * H + 28.0 * H + 16.0 * U + 4.0 * float(1 << B) + cluster_bytes


def headline(out, detail_file="bench_detail.json"):
    """The ONE line the driver parses: the contract's keys, `roofline` (dominant kernel of the headline workload on its own bytes; `at_scale` = BASELINE configs[2] at its own
    size, `genome3g` = the 3 Gb set, `step` = the WHOLE step against all SURVEY 8d bytes for each of the three), `cpu_baseline`, the parity strings and one-number summaries of
    the side blocks. Everything else (per-rank figures, exchanges, device times of the side blocks, notes) is in `detail_file`. Never longer than HEADLINE_MAX_BYTES."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    h = {k: out.get(k) for k in keep}
    cfg = out.get("config", {})
    h["config"] = {k: cfg[k] for k in ("workload", "read_pairs", "barcodes", "B", "k", "w", "hashDepthRange", "clusterThreshold", "parallelism") if k in cfg}
    r = out.get("roofline", {})
    ro = {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale")}
    ro["algorithmic_bytes"] = r.get("algorithmic_bytes_per_launch")
    ro["avg_launch_ms"] = r.get("avg_launch_ms")
    if r.get("traffic_raw"):
        ro["traffic_source"] = "profiles/" + r["traffic_raw"].get("traffic_source", "?")
    for k in ("cluster_all", "index_build"):
        if isinstance(r.get(k), dict):
            ro[k] = {"frac": r[k].get("frac"), "ms": r[k].get("ms_per_step")}
    if isinstance(r.get("other_kernels"), dict) and "mosh_extract" in r["other_kernels"]:
        ro["mosh_extract"] = {"frac": r["other_kernels"]["mosh_extract"].get("frac"), "ms": r["other_kernels"]["mosh_extract"].get("ms_per_step")}

    def scale(e):
        if not isinstance(e, dict):
            return None
        if "frac" not in e:
            return {"config": e.get("config", "")[:60], "skipped": str(e.get("skipped"))[:120]}
        return {"config": e.get("config", "")[:60], "kernel": e.get("kernel"), "frac": e.get("frac"), "achieved": e.get("achieved"), "ms": e.get("ms"),
                "algorithmic_bytes": e.get("algorithmic_bytes"), "traffic": e.get("traffic"), "traffic_stale": e.get("traffic_stale"), "step_ms": e.get("step_ms")}
    for k in ("at_scale", "genome3g"):
        if k in r:
            ro[k] = scale(r[k])
    if isinstance(r.get("step"), dict):
        ro["step"] = r["step"]
    h["roofline"] = ro
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        h["cpu_baseline"] = {k: (cb[k][:260] if k == "sample" and isinstance(cb[k], str) else cb[k]) for k in
                             ("value", "unit", "cores", "kind", "cpu_model", "sample", "cluster_seconds", "read_pairs_per_s_hashed", "barcodes_per_s_clustered") if k in cb}
    for k in ("parity_vs_cpu_on_bench_input", "parity_vs_reference_digest", "cluster_speedup_vs_cpu_1thread", "cluster_speedup_vs_cpu_all_cores", "read_pairs_per_s_hashed",
              "barcodes_per_s_clustered", "build_id", "max_rank_compute_ms", "mean_rank_compute_ms"):
        if out.get(k) is not None:
            h[k] = out[k][:200] if isinstance(out[k], str) else out[k]
    if isinstance(out.get("device_ms_per_step"), dict):
        h["device_ms_per_step"] = {k: round(v, 3) for k, v in out["device_ms_per_step"].items()}
    s = {}
    omp = out.get("cpu_baseline_omp")
    if isinstance(omp, dict):
        s["cpu_baseline_omp"] = {"value": omp.get("value"), "cores": omp.get("cores")}
    e2e = out.get("end_to_end")
    if isinstance(e2e, dict):
        s["end_to_end_wall_seconds"] = e2e.get("wall_seconds")
    sm = out.get("scaling_model_8_ranks")
    if isinstance(sm, dict):
        s["scaling_model_8_ranks"] = ({"workload": sm.get("workload"), "modelled_speedup_vs_1_gpu": sm.get("modelled_speedup_vs_1_gpu"), "modelled_step_ms": sm.get("modelled_step_ms"),
                                       "single_gpu_step_ms": sm.get("single_gpu_step_ms"), "max_rank_compute_ms": sm.get("max_rank_compute_ms"), "mean_rank_compute_ms": sm.get("mean_rank_compute_ms"),
                                       "modelled_exchange_ms": sm.get("modelled_exchange_ms"), "busiest_owner_share": sm.get("busiest_owner_share"), "a MODEL": "ranks as threads taking turns on one GPU"}
                                      if "modelled_speedup_vs_1_gpu" in sm else {"skipped": str(sm.get("skipped") or sm.get("error"))[:160]})
    fc = out.get("full_config3")
    if isinstance(fc, dict):
        e = fc.get("end_to_end") if isinstance(fc.get("end_to_end"), dict) else {}
        s["full_config3"] = ({"ms_per_step": fc.get("ms_per_step"), "read_pairs_per_s": fc.get("read_pairs_per_s"), "sizes_match_reference": fc.get("sizes_match_reference"),
                              "end_to_end": {"wall_seconds": e.get("wall_seconds"), "first_run_wall_seconds": e.get("first_run_wall_seconds"), "per_command_wall_seconds": e.get("per_command_wall_seconds"),
                                             "hash_identical_to_reference": e.get("hash_identical_to_reference"), "error": (e.get("error") or e.get("skipped"))}}
                             if "ms_per_step" in fc else {"skipped": str(fc.get("skipped") or fc.get("error"))[:160]})
    g3 = out.get("genome3g")
    if isinstance(g3, dict):
        if "ms_per_step" in g3:
            c5 = g3.get("config5") if isinstance(g3.get("config5"), dict) else {}
            acc = c5.get("accuracy") if isinstance(c5.get("accuracy"), dict) else {}
            s["genome3g"] = {"ms_per_step": g3.get("ms_per_step"), "read_pairs_per_s": g3.get("read_pairs_per_s"), "parity_vs_reference_digest": g3.get("parity_vs_reference_digest"),
                             "config5": {"purity": acc.get("purity"), "clusters": acc.get("clusters"), "mean_span_kb": acc.get("mean_span_kb"),
                                         "accuracy_vs_reference_report": c5.get("accuracy_vs_reference_report"), "cribBuild_seconds": c5.get("cribBuild_seconds"),
                                         "clusterReport_seconds": c5.get("clusterReport_seconds"), "clusterSplit_seconds": c5.get("clusterSplit_seconds"),
                                         "hbm_high_water_GB": c5.get("hbm_high_water_GB"), "error": c5.get("error")}}
        else:
            s["genome3g"] = {"skipped": str(g3.get("skipped") or g3.get("error"))[:160]}
    sec = out.get("secondary")
    if isinstance(sec, dict) and "ms_per_step" in sec:
        s["config3_tenth"] = {"ms_per_step": sec["ms_per_step"], "cluster_main_frac": (sec.get("roofline") or {}).get("frac"), "traffic_stale": sec.get("traffic_stale")}
    if isinstance(out.get("exchanges"), dict) and "error" not in out["exchanges"]:
        s["exchanges_max_rank_bytes_out"] = {k: v.get("max_rank_bytes_out") for k, v in out["exchanges"].items()}
    h["summary"] = s
    h["detail"] = detail_file

    def drop_none(x):
        if isinstance(x, dict):
            return {k: drop_none(v) for k, v in x.items() if v is not None or k in ("vs_baseline", "traffic")}
        return x
    h = drop_none(_r(h))
    line = json.dumps(h, separators=(",", ":"))
    # a belt for the braces: shed the optional blocks, largest first, until the line fits
    for victim in ("device_ms_per_step", "summary", "cpu_baseline.sample"):
        if len(line) <= HEADLINE_MAX_BYTES:
            break
        if "." in victim:
            a, b = victim.split(".")
            h.get(a, {}).pop(b, None)
        else:
            h.pop(victim, None)
        line = json.dumps(h, separators=(",", ":"))
    assert len(line) <= HEADLINE_MAX_BYTES, len(line)
    return line


def write_detail(out):
    """the whole of what the run measured: bench_detail.json at the repo root and under gpurun_out/ (both untracked scratch; the copy cited by DESIGN.md is committed under profiles/)"""
    txt = json.dumps(out, indent=1)
    for d in (REPO, os.path.join(REPO, "gpurun_out")):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "bench_detail.json"), "w") as f:
                f.write(txt + "\n")
        except OSError:
            pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="yeast-like-2.5M", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config-3-scale blocks (1/10 scale: ~15 s of generation + 6 passes; full size: 24 GB, needs >= 32 host threads)")
    ap.add_argument("--comm", default="auto", choices=["auto", "rccl", "socket"], help="multi-process backend: RCCL over xGMI (one GPU per rank), or the "
                    "host-staged socket backend where ranks share a GPU (test boxes); auto = socket when there are fewer devices than ranks")
    ap.add_argument("--sharded", action="store_true", help="use the multi-GPU code path (RCCL communicator, shard exchange) even with one rank")
    ap.add_argument("--virtual-ranks", type=int, default=0, help="model an N-GPU step on ONE GPU: N ranks as threads taking turns on the device (generator-v2 workloads: genome3g-*); prints "
                    "per-rank compute, every exchange's bytes and the modelled step instead of the benchmark line")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"], help="N > 1: weak = N x the workload (every rank generates its share of the N-fold set; "
                    "default for the yeast-scale set), strong = the SAME set on every N (default for genome3g-300M: the 3 Gb workload of BASELINE configs[3])")
    args = ap.parse_args()
    if args.scaling is None:
        args.scaling = "strong" if WORKLOADS[args.workload].get("gen") == 2 else "weak"

    # the contract is ONE JSON line on stdout: native libraries (RCCL prints a version banner) must not write there,
    # so fd 1 is pointed at stderr for the life of the process and the JSON line goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import hash10x_amd

    if args.virtual_ranks:
        if WORKLOADS[args.workload].get("gen") != 2:
            raise SystemExit("--virtual-ranks needs a generator-v2 workload (genome3g-tenth-30M, genome3g-300M)")
        blk = virtual_ranks_block(hash10x_amd, args.workload, args.virtual_ranks, steps=max(1, min(args.steps, 2)))
        blk["mode"] = "virtual ranks: a model of the %d-GPU step from one GPU, see `model`" % args.virtual_ranks
        os.write(json_fd, (json.dumps(blk) + "\n").encode())
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus %d needs torch.distributed.run (or any launcher that sets RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*), one rank per GPU" % args.gpus)

    wl = dict(WORKLOADS[args.workload])
    comm = None
    backend = "rccl"
    if world > 1 or args.sharded:
        # weak scaling: N times the yeast-scale set (N x pairs, barcodes and genome; table bits grow with log2 N),
        # barcodes sharded over the ranks, hash index exchanged by RCCL all-to-all (csrc/shard.hip)
        if args.scaling == "weak":
            wl = scaled_workload(wl, world)
        ndev = hash10x_amd.device_count()
        backend = args.comm if args.comm != "auto" else ("socket" if world > max(ndev, 1) else "rccl")
        if backend == "socket":
            local_rank = local_rank % max(ndev, 1)          # ranks share the devices there are
            comm = hash10x_amd.Comm.socket(rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1").replace("localhost", "127.0.0.1"),
                                           int(os.environ.get("MASTER_PORT", "29500")) + 61)
        else:
            comm = hash10x_amd.Comm.rccl(rank, world, rendezvous_unique_id(rank, world, hash10x_amd), local_rank)
    t_gen = time.perf_counter()
    if comm is None and wl.get("gen") != 2:
        recs = generate(wl, seed=1)                          # N = 1: the v1 set (the one the committed digests and profiles are of)
        total_pairs = recs.size // 30
    else:
        # N > 1 (and the generator-v2 workloads on one GPU): every rank generates ONLY its shard (generator v2: counter-based streams,
        # OpenMP) — barcodes whose records start in [P r / N, P (r + 1) / N)
        recs, first_record, total_pairs = generate_v2(wl, wl.get("seed", 1), rank if comm is not None else 0, world if comm is not None else 1)
    gen_s = time.perf_counter() - t_gen
    pairs = recs.size // 30
    d_recs = hash10x_amd.DeviceRecords(recs, device=local_rank)   # resident in HBM before the timed region
    hash10x_amd.synchronize(local_rank)

    h = hash10x_amd.Hash10x(B=wl["B"], device=local_rank)
    h.enable_timing(True)

    wall = {"readFQB": 0.0, "hashDepthRange": 0.0, "cluster": 0.0}

    def step():
        t0 = time.perf_counter()
        if comm is None:
            h.read_fqb_device(d_recs.ptr, pairs)
        else:
            h.shard_read_fqb_device(comm, d_recs.ptr, pairs)
        t1 = time.perf_counter()
        h.depth_range(wl["lo"], wl["hi"])
        t2 = time.perf_counter()
        h.cluster(1, 0, wl["ct"])
        t3 = time.perf_counter()
        wall["readFQB"] += t1 - t0
        wall["hashDepthRange"] += t2 - t1
        wall["cluster"] += t3 - t2

    def barrier():
        hash10x_amd.synchronize(local_rank)      # hipDeviceSynchronize (every command also syncs before returning)
        if comm is not None and h._ctx():        # (no context before the first step: --warmup 0)
            h.shard_barrier()                    # RCCL allreduce
        hash10x_amd.synchronize(local_rank)

    for _ in range(args.warmup):
        step()
    # per-kernel device times of the timed steps only
    acc = {}

    def harvest():
        for name, (ms, n) in h.timings().items():
            a = acc.setdefault(name, [0.0, 0])
            a[0] += ms
            a[1] += n

    for k in wall:
        wall[k] = 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        harvest()          # reads finished hipEvents of this step's context (contexts are per-readFQB)
    barrier()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = h.shard_allreduce_max(elapsed)

    ctr = h.counters()
    sizes = h.sizes()
    steps = args.steps
    ms_per_step = 1e3 * elapsed / steps
    value = total_pairs * steps / elapsed

    per = {k: (v[0] / max(v[1], 1), v[1] // steps if steps else 0, v[0] / steps) for k, v in acc.items()}   # avg ms/launch, launches/step, ms/step
    H, U = ctr["entries"], ctr["distinct"]
    T = float(1 << wl["B"])
    alg = {
        # SURVEY §8d algorithmic bytes per launch. The index build's 28 H + 16 U + 4 2^B is split over its four timers by what each
        # of them produces: the sort reads the 16-byte entries and yields the 4-byte barcode lists, index_rank the 12 bytes of
        # hashValue + hashDepth per distinct hash, probe_table the 4-byte slots (U filled, 2^B written), clushash_build the 8-byte records
        "mosh_extract": 120.0 * pairs + 16.0 * H,
        "sort_by_hash": 20.0 * H, "index_rank": 12.0 * U, "probe_table": 4.0 * U + 4.0 * T, "clushash_build": 8.0 * H,
        # the main cluster_kernel launch alone (its own hipEvent bracket and its own work counters: what rocprofv3 lists as
        # cluster_kernel<true, *, 1024, 0>); the few largest barcodes run beside it in a launch of their own
        # Charged with what THIS launch moves: the list entries and the 14 bytes per good hash — not the 16 bytes per (barcode, hash)
        # entry of the read-merge pass, which replay / point_sum / read_merge kernels do behind it (that term flattered the
        # launch by 11 %); `cluster_all` below has every --cluster launch against all the bytes.
        "cluster_main": 4.0 * ctr["cluster_main"][1] + 14.0 * ctr["cluster_main"][0],
    }
    alg_cluster_all = 4.0 * ctr["sum_good_depth"] + 14.0 * ctr["sum_good"] + 16.0 * ctr["sum_hash_clustered"]
    clu_all_ms = per.get("cluster_kernel", (0, 0, 0))[2]
    index_ms = sum(per.get(k, (0, 0, 0))[2] for k in ("sort_by_hash", "index_rank", "probe_table", "clushash_build"))
    index_alg = 28.0 * H + 16.0 * U + 4.0 * T
    dom = max(("mosh_extract", "sort_by_hash", "cluster_main"), key=lambda k: per.get(k, (0, 0, 0))[2])
    dom_ms = per[dom][2]                                    # one launch (bracket) of each of these per step
    achieved = alg[dom] / (dom_ms * 1e-3) / 1e9
    stage_ms = {k: round(v[2], 4) for k, v in per.items() if v[2] > 0}
    hash_ms = sum(per.get(k, (0, 0, 0))[2] for k in ("block_runs", "mosh_extract", "mosh_fallback", "compact_entries", "sort_by_hash",
                                                      "index_rank", "probe_table", "clushash_build"))
    clu_ms = per.get("cluster", (0, 0, 0))[2]

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the figure is the
    # one measured by the committed rocprofv3 --pmc passes (profiles/*_pmc_traffic.json), per step, raw counters
    tr = None
    try:
        if args.workload == "yeast-like-2.5M" and world == 1:
            pm_match = "cluster_kernel<true, 0, 1024, 0>" if dom == "cluster_main" else dom.split("_")[0]
            tr = profile_traffic("*_pmc_traffic.json", pm_match, exclude=("config3", "genome3g"))
    except Exception:
        tr = None
    traffic = tr["traffic"] if tr else None

    out = {
        "metric": "read-pairs/s through --readFQB + --hashDepthRange + --cluster (mosh construction + per-barcode clustering)",
        "value": value, "unit": "read-pairs/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling if (world > 1 or wl.get("gen") == 2) else "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": args.workload + (" x%d" % world if world > 1 and args.scaling == "weak" else ""), "read_pairs": total_pairs, "barcodes": wl["barcodes"], "B": wl["B"], "k": 21, "w": 31,
                   "hashDepthRange": [wl["lo"], wl["hi"]], "clusterThreshold": wl["ct"],
                   "parallelism": ("barcodes sharded over %d ranks, hash index by all-to-all (%s)" % (world, "RCCL over xGMI" if backend == "rccl" else
                                   "host-staged socket backend: ranks share a GPU, a functional run, not a performance figure")) if world > 1 else "single GPU"},
        "read_pairs_per_s_hashed": pairs / (hash_ms * 1e-3) if hash_ms else None,
        "barcodes_per_s_clustered": wl["barcodes"] / (clu_ms * 1e-3) if clu_ms else None,
        "device_ms_per_step": stage_ms,
        "host_wall_ms_per_step": {k: round(1e3 * v / steps, 3) for k, v in wall.items()},
        "entries_H": H, "distinct_U": U, "hashNumber": sizes["hashNumber"], "fallback_blocks": ctr["fallback_blocks"],
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_raw": tr, "traffic_note": TRAFFIC_NOTE if tr else None,
                     # the counters are collected by rocprofv3 --pmc passes of their own (they cannot be read in this process): the figure is of
                     # the build recorded beside it, and stale when that is not the library running now
                     "traffic_stale": (tr["traffic_build_id"] != hash10x_amd.build_id()) if tr else None,
                     "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": dom_ms, "barcodes_in_launch": ctr["cluster_main"][3] if dom == "cluster_main" else None,
                     "other_kernels": {k: {"GB/s": alg[k] / (per[k][2] * 1e-3) / 1e9, "frac": alg[k] / (per[k][2] * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms_per_step": per[k][2],
                                           "algorithmic_bytes": alg[k]} for k in alg if k in per and per[k][2] > 0},
                     "cluster_all": {"what": "all device work of --cluster (classification, the cluster_kernel launches, replay, point_sum, read_merge; timer cluster_kernel) against 4 sum_depth + 14 sum_good + 16 H_clustered",
                                     "GB/s": alg_cluster_all / (clu_all_ms * 1e-3) / 1e9 if clu_all_ms else None, "frac": alg_cluster_all / (clu_all_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if clu_all_ms else None,
                                     "ms_per_step": clu_all_ms, "algorithmic_bytes": alg_cluster_all},
                     "index_build": {"GB/s": index_alg / (index_ms * 1e-3) / 1e9 if index_ms else None, "frac": index_alg / (index_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if index_ms else None,
                                     "ms_per_step": index_ms, "algorithmic_bytes": index_alg, "timers": "sort_by_hash + index_rank + probe_table + clushash_build"},
                     # K1 is bounded by integer ALU, not HBM: 2 x 64-bit multiplies per k-mer (seqhash.c:58-59), 237 k-mers per pair at k = 21;
                     # SQ_INSTS_VALU per launch is in profiles/*_pmc_sq.json (collected by rocprofv3 in passes of their own)
                     "mosh_extract_int_ops": mosh_int_ops(ctr["kmers"], per.get("mosh_extract", (0, 0, 0))[2])},
    }
    out["generate_seconds"] = round(gen_s, 2)
    out["build_id"] = hash10x_amd.build_id()
    if world > 1 or args.sharded:
        # what every kind of exchange moved and how long this rank waited in it, and the ranks' compute: max / sum over the ranks (so that a scaling curve can be read)
        try:
            fig = rank_figures(h)
            kinds = sorted(fig["exchanges"])
            vals = [int(fig["exchanges"][k][f]) for k in kinds for f in ("bytes_out", "max_peer_out")] + [int(1e3 * fig["exchanges"][k]["ms"]) for k in kinds] + [int(1e3 * fig["compute_ms"])]
            mx = h.shard_allreduce_max_u64(list(vals)) if hasattr(h, "shard_allreduce_max_u64") else vals
            sm = h.shard_allreduce_sum_u64(list(vals))
            nk = len(kinds)
            out["exchanges"] = {k: {"calls_per_step": fig["exchanges"][k]["calls"], "max_rank_bytes_out": mx[2 * i], "sum_bytes_out": sm[2 * i], "busiest_peer_share_bytes": mx[2 * i + 1],
                                    "max_rank_wait_ms": mx[2 * nk + i] / 1e3} for i, k in enumerate(kinds)}
            out["max_rank_compute_ms"] = mx[3 * nk] / 1e3
            out["mean_rank_compute_ms"] = sm[3 * nk] / 1e3 / world
            out["exchanges_note"] = "of the last timed step; wait = from the call to its completion on the rank's stream (slower ranks included); compute = stage timers less the waits inside them"
        except Exception as e:
            out["exchanges"] = {"error": str(e)[:200]}
        # correctness gate of a sharded run: a checksum of every rank's blocks and ClusterHash records in file numbering,
        # summed over the ranks, against the one of the REFERENCE binary's .hash of this data set (tests/golden/manifest.json,
        # "bench_scale_digests", made in the build container by make_golden.py --scale)
        try:
            cs = sharded_state_checksum(h, rank)
            ctr_entries_global = h.shard_info()["nEntriesGlobal"]
            man = json.load(open(os.path.join(REPO, "tests", "golden", "manifest.json")))
            exp = man.get("strong_digests", {}).get(args.workload) if args.scaling == "strong" else man.get("bench_scale_digests", {}).get(str(world))
            out["state_checksum"] = ["0x%016x" % v for v in cs]
            if exp is None:
                out["parity_vs_reference_digest"] = "no reference digest committed for %s" % (args.workload if args.scaling == "strong" else "x%d" % world)
            else:
                same = [int(v, 16) for v in exp["checksum"]] == cs and exp["hash_number"] == sizes["hashNumber"] and exp["sum_nHash"] == ctr_entries_global
                out["parity_vs_reference_digest"] = "identical" if same else "DIFFERENT (reference %s, H %d, hashNumber %d)" % (exp["checksum"], exp["sum_nHash"], exp["hash_number"])
        except Exception as e:
            out["parity_vs_reference_digest"] = "check failed: " + str(e)[:200]
    if rank == 0 and world == 1 and not args.sharded and wl.get("gen") == 2:
        # a generator-v2 workload on one GPU: the reference needs most of an hour for it — the parity gate is the digest of its .hash made in the
        # build container (manifest "strong_digests"), checked through the same composable checksum the N-rank runs use
        try:
            blocks = h.export_slice(3, 1, sizes["nBlocks"] - 1)
            cs = checksum_state(blocks, 1, np.zeros(0, dtype=np.uint8), 0)
            for a in range(0, sizes["nClusHash"], 1 << 27):      # ClusterHash records 1 GB at a time
                part = checksum_state(np.zeros(0, dtype=np.uint8), 0, h.export_slice(4, a, min(1 << 27, sizes["nClusHash"] - a)), a)
                cs = [(cs[0] + part[0]) & 0xFFFFFFFFFFFFFFFF, (cs[1] + part[1]) & 0xFFFFFFFFFFFFFFFF]
            exp = json.load(open(os.path.join(REPO, "tests", "golden", "manifest.json"))).get("strong_digests", {}).get(args.workload)
            out["state_checksum"] = ["0x%016x" % v for v in cs]
            if exp is None:
                out["parity_vs_reference_digest"] = "no reference digest committed for %s" % args.workload
            else:
                same = [int(v, 16) for v in exp["checksum"]] == cs and exp["hash_number"] == sizes["hashNumber"] and exp["sum_nHash"] == sizes["nClusHash"]
                out["parity_vs_reference_digest"] = "identical" if same else "DIFFERENT (reference %s, H %d, hashNumber %d)" % (exp["checksum"], exp["sum_nHash"], exp["hash_number"])
        except Exception as e:
            out["parity_vs_reference_digest"] = "check failed: " + str(e)[:200]
    if rank == 0 and world == 1 and not args.sharded and not args.no_cpu_baseline and wl.get("gen") != 2:
        with tempfile.TemporaryDirectory() as d:
            gp = os.path.join(d, "gpu.hash")
            h.write_hash(gp)
            try:
                cb, omp, parity = cpu_baseline(wl, recs, d, gp)
                if cb.get("end_to_end"):
                    out["end_to_end"] = cb.pop("end_to_end")
                out["cpu_baseline"] = cb
                if omp:
                    out["cpu_baseline_omp"] = omp
                out["parity_vs_cpu_on_bench_input"] = parity
                if cb.get("cluster_seconds") and clu_ms:
                    out["cluster_speedup_vs_cpu_1thread"] = cb["cluster_seconds"] / (clu_ms * 1e-3)
                if omp and omp.get("cluster_seconds_estimate") and clu_ms:
                    out["cluster_speedup_vs_cpu_all_cores"] = omp["cluster_seconds_estimate"] / (clu_ms * 1e-3)
            except Exception as e:                       # the baseline is reporting only; never lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "read-pairs/s", "cores": 1, "kind": "error", "sample": str(e)[:300]}
    h.close()
    d_recs.free()
    if rank == 0 and world == 1 and not args.no_secondary and args.workload == "yeast-like-2.5M":
        try:
            out["secondary"] = secondary_block(hash10x_amd, local_rank)
        except Exception as e:                           # never lose the headline over the side block
            out["secondary"] = {"error": str(e)[:300]}
        try:                                                 # BASELINE configs[2] at its own size where the host can generate it in seconds
            out["full_config3"] = full_config3_block(hash10x_amd, local_rank) if (os.cpu_count() or 1) >= 32 else {"skipped": "fewer than 32 host threads: generating 200 M pairs would take minutes"}
        except Exception as e:
            out["full_config3"] = {"error": str(e)[:300]}
        try:                                                 # the 3 Gb strong-scaling workload on this one GPU (36 GB of records: generated in ~30 s on a GPU box's host)
            out["genome3g"] = genome3g_block(hash10x_amd, local_rank) if (os.cpu_count() or 1) >= 32 else {"skipped": "fewer than 32 host threads: generating 300 M pairs would take minutes"}
        except Exception as e:
            out["genome3g"] = {"error": str(e)[:300]}
        try:                                                 # what 8 GPUs would make of the 3 Gb-shaped set, modelled from this one (ranks as threads taking turns) — on the largest set that runs that
            # way on one GPU, the 1/2 set (1/10 sets misled three times; at full size eight ranks' state + the records do not fit): ~30 s with 32+ host threads
            out["scaling_model_8_ranks"] = virtual_ranks_block(hash10x_amd, "genome3g-half-150M" if (os.cpu_count() or 1) >= 32 else "genome3g-tenth-30M", 8, local_rank, steps=2) if (os.cpu_count() or 1) >= 16 else {"skipped": "fewer than 16 host threads"}
        except Exception as e:
            out["scaling_model_8_ranks"] = {"error": str(e)[:300]}
    if rank == 0 and world == 1:
        # the figures that matter at scale, where the driver's parser keeps them: the roofline object. The yeast-scale kernel stays the object's own
        # achieved / frac; at_scale = BASELINE configs[2] at its own size (the config that names the roofline), genome3g = the 3 Gb strong-scaling workload
        def scale_entry(blk, label):
            if not isinstance(blk, dict) or "roofline" not in blk:
                return {"config": label, "skipped": (blk or {}).get("skipped") or (blk or {}).get("error") or "not run"}
            r = blk["roofline"]
            return {"config": label, "kernel": r["kernel"], "frac": r["frac"], "achieved": r["achieved"], "unit": "GB/s", "ms": r["ms_per_step"], "algorithmic_bytes": r["algorithmic_bytes"],
                    "traffic": blk.get("traffic"), "traffic_stale": blk.get("traffic_stale"), "step_ms": blk["ms_per_step"], "read_pairs_per_s": blk["read_pairs_per_s"]}
        if "full_config3" in out:
            out["roofline"]["at_scale"] = scale_entry(out["full_config3"], "BASELINE configs[2] at its own size: 200 M pairs, 1 M barcodes, 500 Mb x 2, -B 29 (config3-full-200M)")
            out["config"]["at_scale_workload"] = "config3-full-200M (see roofline.at_scale, full_config3)"
        if "genome3g" in out:
            out["roofline"]["genome3g"] = scale_entry(out["genome3g"], "3 Gb x 2, 300 M pairs, 1.6 M barcodes, -B 30, --hashDepthRange 6 45 (genome3g-300M: BASELINE configs[3] / [4] shape)")
            out["config"]["genome3g_workload"] = "genome3g-300M (see roofline.genome3g, genome3g, genome3g.config5)"
        # the WHOLE step against ALL of SURVEY 8d's bytes (hashing + index build + clustering): where everything that is not --cluster shows
        sb = step_bytes(total_pairs, H, U, wl["B"], alg_cluster_all)
        out["roofline"]["step"] = {"what": "all SURVEY 8d bytes of a step / ms_per_step / 8 TB/s",
                                   args.workload: {"frac": sb / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes": sb, "ms": ms_per_step}}
        for key, label, Bk in (("full_config3", "config3-full-200M", None), ("genome3g", "genome3g-300M", WORKLOADS["genome3g-300M"]["B"])):
            blk = out.get(key)
            if isinstance(blk, dict) and "roofline" in blk:
                b = blk.get("B", Bk)
                sbk = step_bytes(blk["read_pairs"], blk["entries_H"], blk["distinct_U"], b, blk["roofline"]["algorithmic_bytes"])
                out["roofline"]["step"][label] = {"frac": sbk / (blk["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes": sbk, "ms": blk["ms_per_step"]}
    if rank == 0:
        # the whole record first (file + an EARLIER line on stderr), then — last thing this process writes to stdout — the compact line the driver parses
        write_detail(out)
        sys.stderr.write("bench detail: " + json.dumps(out) + "\n")
        sys.stderr.flush()
        os.write(json_fd, (headline(out) + "\n").encode())
    if comm is not None:
        comm.destroy()
    if _HUNG_THREADS:
        os._exit(0)


if __name__ == "__main__":
    main()
