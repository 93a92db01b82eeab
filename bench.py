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


def cpu_baseline(wl, recs, workdir, gpu_hash_path):
    """Reference binary (oracle/_ref, compiled -O3 from /root/reference) or the oracle port on this box's
    host cores, single thread, same input, same commands. Returns (dict, parity string)."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import orc
    fqb = os.path.join(workdir, "bench.fqb")
    recs.tofile(fqb)
    pairs = recs.size // 30
    gpu_canon = open(gpu_hash_path, "rb").read()
    if orc.have_ref():
        t0 = time.perf_counter()
        r = orc.run_ref(["-B", wl["B"], "-ct", wl["ct"], "--readFQB", "bench.fqb", "--hashDepthRange", wl["lo"], wl["hi"],
                         "--cluster", 1, 0, "--writeHash", "ref.hash"], workdir, timeout=3000)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            raise RuntimeError("reference failed: " + r.stderr.decode())
        # per-command CPU seconds printed by the reference (single thread => ~ wall)
        cmd, tm = None, {}
        for line in r.stdout.decode().splitlines():
            if line.startswith("COMMAND "):
                cmd = line.split()[1]
            elif line.strip().startswith("user") and cmd:
                f = line.split()
                tm[cmd] = tm.get(cmd, 0.0) + float(f[1]) + float(f[3])
        t_read, t_range, t_clu = tm.get("--readFQB", 0.0), tm.get("--hashDepthRange", 0.0), tm.get("--cluster", 0.0)
        ref_canon = orc.canonical_hash_bytes(open(os.path.join(workdir, "ref.hash"), "rb").read())
        parity = "identical" if ref_canon == gpu_canon else "DIFFERENT: " + orc.describe_diff(gpu_canon, ref_canon)
        os.remove(os.path.join(workdir, "ref.hash"))
        t_path = t_read + t_range + t_clu
        return ({"value": pairs / t_path, "unit": "read-pairs/s", "cores": 1, "kind": "reference",
                 "sample": "full workload (%d read pairs): reference hash10x -O3, 1 thread; readFQB %.2fs + hashDepthRange %.2fs + cluster %.2fs "
                           "(its own per-command CPU times; whole process %.2fs wall incl. file I/O)" % (pairs, t_read, t_range, t_clu, wall),
                 "read_pairs_per_s_hashed": pairs / t_read if t_read else None,
                 "barcodes_per_s_clustered": (wl["barcodes"] / t_clu) if t_clu else None,
                 "cluster_seconds": t_clu}, parity)
    o = orc.Oracle(B=wl["B"])
    t0 = time.perf_counter(); o.read_fqb(recs); t1 = time.perf_counter()
    o.depth_range(wl["lo"], wl["hi"]); t2 = time.perf_counter()
    o.cluster(1, 0, wl["ct"], 1); t3 = time.perf_counter()
    o.write_hash(os.path.join(workdir, "orc.hash"))
    ref_canon = open(os.path.join(workdir, "orc.hash"), "rb").read()
    parity = "identical" if ref_canon == gpu_canon else "DIFFERENT: " + orc.describe_diff(gpu_canon, ref_canon)
    return ({"value": pairs / (t3 - t0), "unit": "read-pairs/s", "cores": 1, "kind": "port",
             "sample": "full workload (%d read pairs): oracle C restatement, 1 thread; readFQB %.2fs + hashDepthRange %.2fs + cluster %.2fs"
                       % (pairs, t1 - t0, t2 - t1, t3 - t2),
             "read_pairs_per_s_hashed": pairs / (t1 - t0), "barcodes_per_s_clustered": wl["barcodes"] / (t3 - t2),
             "cluster_seconds": t3 - t2}, parity)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="yeast-like-2.5M", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharded", action="store_true", help="use the multi-GPU code path (RCCL communicator, shard exchange) even with one rank")
    args = ap.parse_args()

    # the contract is ONE JSON line on stdout: native libraries (RCCL prints a version banner) must not write there,
    # so fd 1 is pointed at stderr for the life of the process and the JSON line goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import hash10x_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus %d needs torch.distributed.run (or any launcher that sets RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*), one rank per GPU" % args.gpus)

    wl = dict(WORKLOADS[args.workload])
    comm = None
    if world > 1 or args.sharded:
        # weak scaling: N times the yeast-scale set (N x pairs, barcodes and genome; table bits grow with log2 N),
        # barcodes sharded over the ranks, hash index exchanged by RCCL all-to-all (csrc/shard.hip)
        wl["pairs"] *= world
        wl["barcodes"] *= world
        wl["genome"] *= world
        wl["B"] += (world - 1).bit_length()
        comm = hash10x_amd.Comm.rccl(rank, world, rendezvous_unique_id(rank, world, hash10x_amd), local_rank)
    recs = generate(wl, seed=1)                              # every rank builds the same seeded set and keeps its shard
    total_pairs = recs.size // 30
    if comm is not None:
        cut = hash10x_amd.partition(recs, world)
        recs = recs[30 * cut[rank]: 30 * cut[rank + 1]].copy()
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
        if comm is not None:
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
    alg = {
        # SURVEY §8d algorithmic bytes per launch
        "mosh_extract": 120.0 * pairs + 16.0 * H,
        "sort_by_hash": 28.0 * H + 16.0 * U + 4.0 * (1 << wl["B"]),         # whole index build; reported against the sort, its dominant kernel
        # the main cluster_kernel launch alone (its own hipEvent bracket and its own work counters: what rocprofv3 lists as
        # cluster_kernel<true, *, 1024, 0>); the few largest barcodes run beside it in a launch of their own
        "cluster_main": 4.0 * ctr["cluster_main"][1] + 14.0 * ctr["cluster_main"][0] + 16.0 * ctr["cluster_main"][2],
    }
    dom = max(alg, key=lambda k: per.get(k, (0, 0, 0))[2])
    dom_ms = per[dom][2]                                    # one launch (bracket) of each of these per step
    achieved = alg[dom] / (dom_ms * 1e-3) / 1e9
    stage_ms = {k: round(v[2], 4) for k, v in per.items() if v[2] > 0}
    hash_ms = sum(per.get(k, (0, 0, 0))[2] for k in ("block_runs", "mosh_extract", "mosh_fallback", "compact_entries", "sort_by_hash",
                                                      "index_rank", "probe_table", "clushash_build"))
    clu_ms = per.get("cluster", (0, 0, 0))[2]

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the figure is the
    # one measured by the committed rocprofv3 --pmc passes (profiles/*_pmc_traffic.json), per step, raw counters
    traffic = traffic_src = None
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(REPO, "profiles", "*_pmc_traffic.json")))
        if cands and args.workload == "yeast-like-2.5M" and world == 1:
            pm = json.load(open(cands[-1]))
            tb = sum(sum(c["bytes_per_step"] for c in v.values()) for k, v in pm["kernels"].items() if (pm.get("dominant") or dom.split("_")[0]) in k)
            if tb:
                traffic, traffic_src = tb, os.path.basename(cands[-1])
    except Exception:
        traffic = None

    out = {
        "metric": "read-pairs/s through --readFQB + --hashDepthRange + --cluster (mosh construction + per-barcode clustering)",
        "value": value, "unit": "read-pairs/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": args.workload + (" x%d" % world if world > 1 else ""), "read_pairs": total_pairs, "barcodes": wl["barcodes"], "B": wl["B"], "k": 21, "w": 31,
                   "hashDepthRange": [wl["lo"], wl["hi"]], "clusterThreshold": wl["ct"],
                   "parallelism": ("barcodes sharded over %d GPUs, hash index by RCCL all-to-all" % world) if world > 1 else "single GPU"},
        "read_pairs_per_s_hashed": pairs / (hash_ms * 1e-3) if hash_ms else None,
        "barcodes_per_s_clustered": wl["barcodes"] / (clu_ms * 1e-3) if clu_ms else None,
        "device_ms_per_step": stage_ms,
        "host_wall_ms_per_step": {k: round(1e3 * v / steps, 3) for k, v in wall.items()},
        "entries_H": H, "distinct_U": U, "hashNumber": sizes["hashNumber"], "fallback_blocks": ctr["fallback_blocks"],
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": dom_ms, "barcodes_in_launch": ctr["cluster_main"][3] if dom == "cluster_main" else None,
                     "other_kernels": {k: {"GB/s": alg[k] / (per[k][2] * 1e-3) / 1e9, "ms_per_step": per[k][2]} for k in alg if k in per and per[k][2] > 0}},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        with tempfile.TemporaryDirectory() as d:
            gp = os.path.join(d, "gpu.hash")
            h.write_hash(gp)
            try:
                cb, parity = cpu_baseline(wl, recs, d, gp)
                out["cpu_baseline"] = cb
                out["parity_vs_cpu_on_bench_input"] = parity
                if cb.get("cluster_seconds") and clu_ms:
                    out["cluster_speedup_vs_cpu_1thread"] = cb["cluster_seconds"] / (clu_ms * 1e-3)
            except Exception as e:                       # the baseline is reporting only; never lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "read-pairs/s", "cores": 1, "kind": "error", "sample": str(e)[:300]}
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    h.close()
    if comm is not None:
        comm.destroy()


if __name__ == "__main__":
    main()
