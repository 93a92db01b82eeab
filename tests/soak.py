"""Randomised differential soak (test infrastructure): the HIP path against the CPU oracle on seeded random sets and random
parameters — k, w, r, depth ranges, threshold, the cluster placement / budget knobs, a second range, clusterSplit + re-cluster,
a barcode sub-range, -N, a write/read round trip
through a .hash file, and the sharded form with 1-8 ranks as threads (fewer barcodes than ranks included). `run(n, seed)` returns the
descriptions of the cases that differ; tests/test_gpu_parity.py runs a short one, `python tests/soak.py 500 7` a long one."""
import os, random, shutil, sys, tempfile, threading, time

import numpy as np

import orc
from driver import run_commands


class ShardEngine:
    """One rank of a sharded run behind the engine interface of driver.run_commands: --readFQB takes this rank's barcode range
    (after -N), everything else is the same collective call on every rank. gather=True: rank 0 collects the state before
    --writeHash (h10x_shard_gather), else every rank writes its slices of the file."""

    def __init__(self, h, comm, nranks, gather):
        self.h, self.comm, self.nranks, self.gather = h, comm, nranks, gather

    def read_fqb(self, records, N=0, chunk=100000):
        import hash10x_amd
        flat = np.ascontiguousarray(records, dtype=np.uint32).reshape(-1)
        if N and N * 30 < flat.size:
            flat = flat[: N * 30]
        cut = hash10x_amd.partition(flat, self.nranks)
        self.h.shard_read_fqb(self.comm, flat[30 * cut[self.comm.rank]: 30 * cut[self.comm.rank + 1]])

    def depth_range(self, lo, hi): self.h.depth_range(lo, hi)
    def cluster(self, a, b, ct): self.h.cluster(a, b, ct)
    def cluster_split(self): self.h.cluster_split()

    def write_hash(self, path):
        if self.gather:
            self.h.shard_gather()
            if self.comm.rank == 0: self.h.write_hash(path)
            self.gather = None                                # the state is rank 0's from here on: nothing sharded may follow
        else:
            self.h.write_hash(path)


def run_sharded(args, cwd, nranks, k, w, r, B, opts=None, gather=False):
    """the command line `args` on nranks ranks (threads of this process, in-process communicator)"""
    import hash10x_amd
    comms = hash10x_amd.Comm.local(nranks); errs = [None] * nranks

    def work(rk):
        try:
            def make(k_, w_, r_, B_):
                h = hash10x_amd.Hash10x(k=k_, w=w_, r=r_, B=B_)
                for n, v in (opts or {}).items(): h.set_option(n, v)
                return ShardEngine(h, comms[rk], nranks, gather)
            eng = run_commands(make, args, cwd)
            eng.h.close()
        except Exception as e:                                # noqa: BLE001
            errs[rk] = e
    th = [threading.Thread(target=work, args=(rk,)) for rk in range(nranks)]
    [x.start() for x in th]; [x.join() for x in th]
    for cm in comms: cm.destroy()
    for e in errs:
        if e: raise e


def run(n_cases, seed, verbose=False, scale=1):
    """scale > 1: pairs, barcodes and genome multiplied (launch classes, table overflows and multi-round lists at real sizes)"""
    import hash10x_amd
    rng = random.Random(seed)
    bad = []
    for case in range(n_cases):
        d = tempfile.mkdtemp()
        k = rng.choice([21, 21, 21, 16, 24, 25, 31, 11, 19]); w = rng.choice([31, 31, 31, 5, 7, 32, 13, 64])
        r = rng.choice([17, 17, 3, 9, 101])
        pairs = rng.choice([60, 500, 3000, 8000, 20000, 40000]); barcodes = rng.choice([1, 2, 3, 5, 20, 60, 150, 400]); genome = rng.choice([3000, 12000, 40000, 100000, 300000])      # small genomes: hashes shared by hundreds of barcodes (long lists)
        pairs *= scale; barcodes *= max(1, scale // 3); genome *= scale
        if barcodes > pairs // 4: barcodes = max(1, pairs // 4)
        mol = rng.choice([2.0, 3.0, 4.0]); mol_len = rng.choice([2500, 5000, 8000]); err = rng.choice([0.001, 0.003, 0.01])
        lo = rng.choice([1, 2, 3, 4, 6]); hi = lo + rng.choice([1, 3, 10, 30, 100, 100000]); ct = rng.choice([1, 2, 3, 5, 40])
        B = (23 if w < 13 else 21) + (scale - 1).bit_length()
        opts = {}
        if rng.random() < 0.3: opts["cluster_first_global"] = rng.choice([1, 2, 3, 4, 4])
        if rng.random() < 0.2: opts["cluster_lds_budget"] = rng.choice([2048, 16 * 1024, 24 * 1024, 48 * 1024])
        if rng.random() < 0.15: opts["stage_a_max_slots"] = rng.choice([256, 1024, 4096])
        if rng.random() < 0.15 and "cluster_first_global" in opts: opts["cluster_first_cap"] = rng.choice([64, 500, 1500])
        if opts.get("cluster_first_global") == 4 and rng.random() < 0.5: opts["cluster_first_cap"] = rng.choice([64, 128, 200, 300, 500, 1500])   # translated placement: tables that close, second tables, overflow chain
        if rng.random() < 0.2: opts["index_no_pack"] = 1
        if rng.random() < 0.3: opts["index_priv_table"] = rng.choice([1, 1, 3])
        if rng.random() < 0.25: opts["cluster_narrow_first"] = rng.choice([1, 5, 12])
        if os.environ.get("H10X_SOAK_TRANSLATED"): opts["cluster_first_global"] = 4   # every case in the translated placement (python tests/soak.py with this set: a soak of its forms)
        if opts.get("cluster_first_global") == 4:              # forms of the translated placement (a generator of their own: the cases of a seed stay what they were)
            rng2 = random.Random((seed << 20) ^ case)
            if rng2.random() < 0.3: opts["cluster_tr_class_t"] = 0
            if rng2.random() < 0.2: opts["cluster_tr_packed"] = 0
            if rng2.random() < 0.3: opts["cluster_threads0"] = rng2.choice([512, 768])
        tail = ["-ct", ct, "--readFQB", "x.fqb", "--hashDepthRange", lo, hi, "--cluster", 1, 0]
        simple = True
        u = rng.random()
        if u < 0.12:                                          # a sub-range of barcodes
            c0 = rng.randint(1, max(1, barcodes // 2)); tail[-2:] = [c0, rng.choice([0, c0, c0 + 1, c0 + max(1, barcodes // 3)])]; simple = False   # (the set may hold fewer barcodes than asked for: stay in its lower part)
        elif u < 0.22:                                        # -N: only the first records of the file
            tail = ["-N", rng.randint(1, pairs)] + tail; simple = False
        elif u < 0.32:                                        # through a .hash file: write, read back, cluster again
            tail += ["--writeHash", "mid.hash", "--readHash", "mid.hash", "--hashDepthRange", lo, hi + 2, "--cluster", 1, 0]; simple = False
        if rng.random() < 0.3: tail += ["--hashDepthRange", lo + 1, hi + 5, "--cluster", 1, 0]; simple = False
        if rng.random() < 0.3: tail += ["--clusterSplit", "--hashDepthRange", lo, hi, "--cluster", 1, 0]; simple = False
        desc = dict(case=case, k=k, w=w, r=r, pairs=pairs, barcodes=barcodes, genome=genome, mol=mol, mol_len=mol_len, err=err, B=B, opts=opts, tail=tail)
        try:
            recs = orc.gen_fqb(os.path.join(d, "x.fqb"), pairs, barcodes, genome, err, 1000 + case, mol, 150, mol_len)
            base = ["-k", k, "-w", w, "-r", r, "-B", B] + (["-c", pairs + 1] if scale > 1 else []) + tail   # (a barcode of > 100000 pairs needs -c, as in the reference)

            def make(k_, w_, r_, B_):
                h = hash10x_amd.Hash10x(k=k_, w=w_, r=r_, B=B_)
                for n, v in opts.items(): h.set_option(n, v)
                return h
            run_commands(make, base + ["--writeHash", "hip.hash"], d)
            run_commands(lambda k_, w_, r_, B_: orc.Oracle(k_, w_, r_, B_), base + ["--writeHash", "orc.hash"], d)
            got = open(os.path.join(d, "hip.hash"), "rb").read(); exp = open(os.path.join(d, "orc.hash"), "rb").read()
            ok = got == exp
            nr = rng.choice([1, 2, 3, 4, 8])
            if ok and "--readHash" not in tail and rng.random() < 0.6:   # the same commands sharded, ranks as threads
                sopts = dict(opts)
                if random.Random((seed << 21) ^ case).random() < 0.35: sopts["shard_reply_sort"] = 1 + case % 2   # (own generator: the cases of a seed stay what they were)
                if rng.random() < 0.4: sopts["shard_row_shift"] = rng.choice([0, 1, 3])
                if rng.random() < 0.3: sopts["shard_rows_fake_base"] = rng.choice([1 << 32, (5 << 32) + 24, 1 << 35])
                gather = "--clusterSplit" not in tail and rng.random() < 0.3
                run_sharded(base + ["--writeHash", "sh.hash"], d, nr, k, w, r, B, sopts, gather)
                got = open(os.path.join(d, "sh.hash"), "rb").read()
                ok = got == exp
                desc["sharded_ranks"] = nr; desc["sharded_opts"] = sopts; desc["gather"] = gather
            if verbose: print("case %3d %s %s" % (case, "ok " if ok else "MISMATCH", desc if not ok else {x: desc[x] for x in ("k", "w", "barcodes")}), flush=True)
            if not ok:
                desc["diff"] = orc.describe_diff(got, exp)[:400]; bad.append(desc)
        except Exception as e:                                # noqa: BLE001
            desc["exception"] = repr(e); bad.append(desc)
            if verbose: print("case %3d EXCEPTION %r" % (case, e), flush=True)
        shutil.rmtree(d, ignore_errors=True)
    return bad


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    t0 = time.time()
    bad = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 2026, verbose=True, scale=int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    for b in bad: print("BAD", b)
    print("soak: %d cases, %d bad, %.0f s" % (n, len(bad), time.time() - t0))
    sys.exit(1 if bad else 0)
