#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REAL reference.

Runs only in the build container: needs oracle/_ref/ (compiled from /root/reference by
`make -C oracle ref`). Every expected output below is produced by the reference binary itself
(`MALLOC_PERTURB_=255`) and canonicalised (heap-pointer fields zeroed).
The committed artefacts are data only: inputs (.fqb / hand-built .hash) and expected outputs.

    python tests/golden/make_golden.py

Fixtures:
  kat.json            KAT-1 constants (glibc random() => factor1) and KAT-2 mosh lists printed by the
                      reference's own -DTEST seqhash main (k=16, w=32) for fixed sequences
  tiny.fqb            hand-built 16-record file with the quirks: duplicated read (min-read rule),
                      a barcode without any mosh (bogus hash-0 entry), a poly-A read, a hash shared
                      between barcodes on opposite strands, and the trailing barcode that is dropped
  small.fqb.gz        seeded synthetic linked reads (gen_fqb) small enough to commit
  *.hash.gz           canonical reference outputs for the command lines in manifest.json
  abort255.in.hash.gz hand-written state that drives codeClusterFind past 255 raw clusters
  manifest.json       command lines + sha256 of every artefact, and digest-only cases for larger
                      seeded sets that tests regenerate with gen_fqb
"""
import gzip
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402


def gz_write(path, data):
    with open(path, "wb") as raw:
        with gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0, compresslevel=9) as f:
            f.write(data)


def ref(args, cwd):
    r = orc.run_ref(args, cwd, timeout=3600)
    if r.returncode != 0:
        raise RuntimeError("reference failed: %s\n%s" % (args, r.stderr.decode()))
    return r.stdout.decode(), r.stderr.decode()


def rand_codes(rng, n):
    return rng.integers(0, 4, size=n, dtype=np.uint8)


def revcomp(c):
    return (3 - np.asarray(c, dtype=np.uint8))[::-1].copy()


def build_tiny(rng):
    o = orc.Oracle(k=21, w=31, seed=17, B=20)

    def n_mosh(tail, r2):
        rec = orc.make_record(1, tail, r2)
        s1 = np.zeros(160, np.uint8)
        s2 = np.zeros(160, np.uint8)
        orc.lib().orc_unpack160(rec[0:10].ctypes.data, s1.ctypes.data)
        orc.lib().orc_unpack160(rec[15:25].ctypes.data, s2.ctypes.data)
        return len(o.mosh(s1[23:150])[0]) + len(o.mosh(s2[0:150])[0])

    # a read pair without any mosh; ~1 in 2000 random pairs
    while True:
        t, r = rand_codes(rng, 135), rand_codes(rng, 151)
        if n_mosh(t, r) == 0:
            nomosh = (t, r)
            break
    pA0, pA2 = (rand_codes(rng, 135), rand_codes(rng, 151)), (rand_codes(rng, 135), rand_codes(rng, 151))
    polyA = np.zeros(151, np.uint8)
    recs = []
    bcA, bcB, bcC, bcD, bcE = 0x01234567, 0x1B2D3F41, 0x6789ABCD, 0xA0A0F0F0, 0xFEDCBA98
    # barcode A: read 0, an exact copy of it (every hash duplicated: the lower read index must win), read 2
    recs += [orc.make_record(bcA, *pA0), orc.make_record(bcA, *pA0), orc.make_record(bcA, *pA2)]
    # barcode B: single pair with no mosh at all
    recs += [orc.make_record(bcB, *nomosh)]
    # barcode C: poly-A read 2 (canonical hash 0 of AAAA..), a pair sharing read 2 with A's pair 2, two random
    recs += [orc.make_record(bcC, rand_codes(rng, 135), polyA), orc.make_record(bcC, rand_codes(rng, 135), pA2[1]),
             orc.make_record(bcC, rand_codes(rng, 135), rand_codes(rng, 151)),
             orc.make_record(bcC, rand_codes(rng, 135), rand_codes(rng, 151))]
    # barcode D: read 2 = reverse complement of A's read-0 read 2 (same canonical k-mers), then the no-mosh pair
    recs += [orc.make_record(bcD, rand_codes(rng, 135), revcomp(pA0[1])), orc.make_record(bcD, *nomosh)]
    # barcode E: three pairs, trailing => never hashed
    recs += [orc.make_record(bcE, rand_codes(rng, 135), rand_codes(rng, 151)) for _ in range(3)]
    # barcode F..: a few more single-pair barcodes so that -N truncation cases differ
    recs += [orc.make_record(0xFFFF0000 + i, rand_codes(rng, 135), pA0[1]) for i in range(3)]
    return np.stack(recs).astype(np.uint32)


def build_abort255():
    """600 hashes in barcode 1; each consecutive pair is also held by five private helper
    barcodes => every second good hash founds a new raw cluster => 300 > 255 => abort path."""
    B = 20
    nh = 600
    hash_number = nh + 1
    hash_value = np.zeros(hash_number, np.uint64)
    hash_value[1:] = (np.arange(1, hash_number, dtype=np.uint64) * np.uint64(2654435761)) * np.uint64(31)
    hash_index = np.zeros(1 << B, np.uint32)
    mask = (1 << B) - 1
    for ix in range(1, hash_number):
        h = int(hash_value[ix])
        s, d = h & mask, ((h >> B) & mask) | 1
        while hash_index[s]:
            s = (s + d) & mask
        hash_index[s] = ix
    n_helpers = 5 * (nh // 2)
    n_blocks = 1 + 1 + n_helpers            # slot 0 unused
    depth_dim = 1 << 20
    depth = np.zeros(depth_dim, np.uint32)
    depth[1:hash_number] = 6
    blocks = np.zeros(n_blocks, orc.BLOCK)
    blocks["nRead"][1] = nh
    blocks["nHash"][1] = nh
    blocks["nRead"][2:] = 1
    blocks["nHash"][2:] = 2
    ch = np.zeros(nh + 2 * n_helpers, orc.CLUSHASH)
    ch["hash"][:nh] = np.arange(1, nh + 1)
    ch["read"][:nh] = np.arange(nh)
    for m in range(nh // 2):
        for j in range(5):
            k = nh + 2 * (5 * m + j)
            ch["hash"][k] = 2 * m + 1
            ch["hash"][k + 1] = 2 * m + 2
    out = bytearray()
    out += b"10XH" + (2).to_bytes(4, "little") + (8).to_bytes(2, "little") + (32).to_bytes(2, "little")
    out += B.to_bytes(4, "little") + hash_index.tobytes() + hash_number.to_bytes(4, "little") + hash_value.tobytes()
    hdr = np.zeros(1, dtype="<i4,<i4,<u8,<i4,<i4,<i4,<i4")
    hdr[0] = (8918274, 0, 0, depth_dim, 4, hash_number, 0)
    out += hdr.tobytes() + depth.tobytes()
    hdr[0] = (8918274, 0, 0, n_blocks, 32, n_blocks, 0)
    out += hdr.tobytes() + blocks.tobytes() + ch.tobytes()
    return bytes(out)


def main():
    if not orc.have_ref():
        sys.exit("oracle/_ref/hash10x missing: run `make -C oracle ref` in the build container")
    man = {"note": "generated by tests/golden/make_golden.py from oracle/_ref (reference compiled -O3, "
                   "MALLOC_PERTURB_=255, pointer fields zeroed)", "cases": [], "digest_cases": []}
    tmp = tempfile.mkdtemp(prefix="h10x_golden_")
    rng = np.random.default_rng(20181227)

    # ---------------- KAT-1 / KAT-2
    kat = {"seed17_factor1": "0x%016x" % orc.lib().orc_factor1_from_seed(17),
           "k21": {"mask": "0x%x" % ((1 << 42) - 1), "shift1": 22,
                   "patternRC": ["0x%x" % ((3 - i) << 40) for i in range(4)]}}
    seqs = ["".join("ACGT"[c] for c in rand_codes(rng, n)) for n in (151, 64, 40, 16, 15, 300)]
    seqs.append("A" * 60)
    seqs.append("ACGT" * 30)
    fa = "".join(">s%d\n%s\n" % (i, s) for i, s in enumerate(seqs))
    r = subprocess.run([os.path.join(orc.REF_DIR, "seqhash_test")], input=fa.encode(), stdout=subprocess.PIPE, check=True)
    cur, table = None, []
    for line in r.stdout.decode().splitlines():
        if line.startswith("read sequence"):
            cur = {"id": line.split()[2], "len": int(line.split()[-1]), "moshes": []}
            table.append(cur)
        elif line.startswith("\t"):
            h, p, f = line.split()
            cur["moshes"].append([h, int(p), f])
    kat["seqhash_test_k16_w32_default_seed"] = {"sequences": seqs, "out": table}
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1)

    def add_case(name, inp, args, out_name):
        """Run the reference with args (input/output file names relative to tmp) and store canonical out."""
        ref(args, tmp)
        data = orc.canonical_hash_bytes(open(os.path.join(tmp, out_name), "rb").read())
        gz_write(os.path.join(HERE, out_name + ".gz"), data)
        man["cases"].append({"name": name, "input": inp, "args": [str(a) for a in args], "output": out_name + ".gz",
                             "sha256": orc.sha256(data)})
        return data

    # ---------------- tiny
    tiny = build_tiny(rng)
    tiny.tofile(os.path.join(HERE, "tiny.fqb"))
    shutil.copy(os.path.join(HERE, "tiny.fqb"), os.path.join(tmp, "tiny.fqb"))
    add_case("tiny.readfqb", "tiny.fqb", ["-B", 20, "--readFQB", "tiny.fqb", "--writeHash", "tiny.hash"], "tiny.hash")
    add_case("tiny.N9", "tiny.fqb", ["-B", 20, "-N", 9, "--readFQB", "tiny.fqb", "--writeHash", "tiny.N9.hash"], "tiny.N9.hash")
    add_case("tiny.N15.c2", "tiny.fqb", ["-B", 20, "-N", 15, "-c", 5, "--readFQB", "tiny.fqb", "--writeHash", "tiny.N15.hash"], "tiny.N15.hash")
    add_case("tiny.k16w5", "tiny.fqb", ["-k", 16, "-w", 5, "-r", 3, "-B", 21, "--readFQB", "tiny.fqb", "--writeHash", "tiny.k16.hash"], "tiny.k16.hash")
    add_case("tiny.k31w64", "tiny.fqb", ["-k", 31, "-w", 64, "-B", 20, "--readFQB", "tiny.fqb", "--writeHash", "tiny.k31.hash"], "tiny.k31.hash")
    add_case("tiny.cluster", "tiny.hash.gz", ["-B", 20, "-ct", 1, "--readHash", "tiny.hash", "--hashDepthRange", 1, 4,
                                               "--cluster", 1, 0, "--writeHash", "tiny.c.hash"], "tiny.c.hash")

    # ---------------- small (committed input)
    small = orc.gen_fqb(os.path.join(tmp, "small.fqb"), pairs=4000, barcodes=40, genome=40000, err=0.004, seed=11, mol=3.0, mol_len=2500)
    gz_write(os.path.join(HERE, "small.fqb.gz"), small.tobytes())
    add_case("small.readfqb", "small.fqb.gz", ["-B", 20, "--readFQB", "small.fqb", "--writeHash", "small.hash"], "small.hash")
    for lo, hi, ct in ((3, 14, 2), (3, 12, 3), (2, 14, 5)):
        nm = "small.c_%d_%d_%d.hash" % (lo, hi, ct)
        add_case("small.cluster_%d_%d_ct%d" % (lo, hi, ct), "small.hash.gz",
                 ["-B", 20, "-ct", ct, "--readHash", "small.hash", "--hashDepthRange", lo, hi, "--cluster", 1, 0,
                  "--writeHash", nm], nm)
    # accumulating ranges + partial code range + re-clustering an already clustered file
    add_case("small.cluster_accum", "small.hash.gz",
             ["-B", 20, "-ct", 2, "--readHash", "small.hash", "--hashDepthRange", 3, 5, "--hashDepthRange", 8, 12, "--cluster", 2, 30,
              "--writeHash", "small.accum.hash"], "small.accum.hash")
    add_case("small.recluster", "small.c_3_14_2.hash.gz",
             ["-B", 20, "-ct", 3, "--readHash", "small.c_3_14_2.hash", "--hashDepthRange", 3, 14, "--cluster", 1, 0,
              "--writeHash", "small.recluster.hash"], "small.recluster.hash")
    add_case("small.split", "small.c_3_14_2.hash.gz",
             ["-B", 20, "--readHash", "small.c_3_14_2.hash", "--clusterSplit", "--writeHash", "small.split.hash"], "small.split.hash")
    # end-to-end in one invocation (readFQB -> range -> cluster), as the README recipe chains commands
    add_case("small.e2e", "small.fqb.gz",
             ["-B", 20, "-ct", 2, "--readFQB", "small.fqb", "--hashDepthRange", 3, 14, "--cluster", 1, 0, "--writeHash", "small.e2e.hash"],
             "small.e2e.hash")

    # ---------------- > 255 raw clusters: abort path (hash10x.c:810-816)
    ab = build_abort255()
    gz_write(os.path.join(HERE, "abort255.in.hash.gz"), ab)
    open(os.path.join(tmp, "abort255.in.hash"), "wb").write(ab)
    add_case("abort255", "abort255.in.hash.gz",
             ["-B", 20, "--readHash", "abort255.in.hash", "--hashDepthRange", 2, 100, "--cluster", 1, 2,
              "--writeHash", "abort255.out.hash"], "abort255.out.hash")

    # ---------------- digest-only cases: inputs regenerated by gen_fqb (seeded), outputs pinned by sha256
    def digest_case(name, gen, B, extra):
        p = os.path.join(tmp, name + ".fqb")
        recs = orc.digest_input(p, gen)
        pre = orc.leading_options(extra)
        args = ["-B", B] + pre + ["--readFQB", name + ".fqb"] + extra[len(pre):] + ["--writeHash", name + ".hash"]
        ref(args, tmp)
        data = orc.canonical_hash_bytes(open(os.path.join(tmp, name + ".hash"), "rb").read())
        hf = orc.HashFile(data)
        man["digest_cases"].append({
            "name": name, "gen": gen, "B": B, "args": [str(a) for a in extra], "input_sha256": orc.sha256(recs.tobytes()),
            "sha256": orc.sha256(data), "size": len(data), "hash_number": hf.hash_number, "blocks_max": hf.blocks_max,
            "sum_nHash": int(hf.blocks["nHash"].sum()), "sum_nSubCluster": int(hf.blocks["nSubCluster"].sum()),
            "sum_pointToMin_hex": float(hf.blocks["pointToMin"].sum()).hex()})
        os.remove(os.path.join(tmp, name + ".hash"))

    digest_case("mid", dict(pairs=20000, barcodes=40, genome=1000000, err=0.005, seed=7, mol=10.0), 20, [])
    digest_case("mid.c", dict(pairs=20000, barcodes=40, genome=1000000, err=0.005, seed=7, mol=10.0), 20,
                ["-ct", 2, "--hashDepthRange", 3, 9, "--cluster", 1, 0])
    digest_case("dense.c", dict(pairs=60000, barcodes=200, genome=400000, err=0.002, seed=5, mol=4.0), 20,
                ["--hashDepthRange", 8, 40, "--cluster", 1, 0])
    # a barcode with > 65535 unique hashes is ignored by clustering (hash10x.c:748-753)
    digest_case("big65k.c", dict(pairs=36000, barcodes=3, genome=3000000, err=0.01, seed=3, mol=12.0), 20,
                ["--hashDepthRange", 1, 3, "--cluster", 1, 0])

    # a barcode with > 65535 read pairs (ClusterHash.read is U16: hash10x.c:37,180): BIG also has > 65535 hashes, BIG2 has few
    # hashes and is clustered with its late reads stored modulo 2^16; -c must exceed the longest run (hash10x.c:206)
    digest_case("pairs65k.c", {"builder": "pairs65k"}, 22, ["-c", 200000, "-ct", 2, "--hashDepthRange", 4, 30, "--cluster", 1, 0])
    digest_case("pairs65k.split", {"builder": "pairs65k"}, 22, ["-c", 200000, "-ct", 2, "--hashDepthRange", 4, 30, "--cluster", 1, 0, "--clusterSplit"])

    # BASELINE configs[2] proportions (500 Mb x 2, 200 M pairs, 1 M barcodes, e = 0.1 %) at 1/10 and 1/4 scale: the ranked and the
    # hashed placement of first[] at their natural sizes. The reference needs 4.5 and 12.5 minutes for them, so they are only
    # regenerated on request (`make_golden.py --big`); otherwise the committed entries are carried over.
    big = [("c3_20m.c", dict(pairs=20000000, barcodes=100000, genome=50000000, err=0.001, seed=2, mol=10.0), 26),
           ("c3_50m.c", dict(pairs=50000000, barcodes=300000, genome=125000000, err=0.001, seed=2, mol=10.0), 27)]
    if "--big" in sys.argv:
        keep = man["digest_cases"]
        man["digest_cases"] = []
        for name, gen, B in big:
            digest_case(name, gen, B, ["--hashDepthRange", 30, 100, "--cluster", 1, 0])
            os.remove(os.path.join(tmp, name + ".fqb"))
        man["big_digest_cases"], man["digest_cases"] = man["digest_cases"], keep
        # the table size BASELINE configs[2] names is too small for its read count: the reference dies (1/10 set, proportional -B)
        name, gen, B = big[0]
        orc.gen_fqb(os.path.join(tmp, "die.fqb"), **gen)
        r = orc.run_ref(["-B", B - 1, "--readFQB", "die.fqb"], tmp, timeout=3600)
        os.remove(os.path.join(tmp, "die.fqb"))
        assert r.returncode != 0 and b"hashTableSize is too small" in r.stderr, r.stderr
        try:
            with open(os.path.join(HERE, "manifest.json")) as f:
                other_dies = [d for d in json.load(f).get("die_cases", []) if d["name"] != "c3_20m.B%d" % (B - 1)]      # (c3_200m.B28: recorded by hand from a 9-minute run, see its note)
        except OSError:
            other_dies = []
        man["die_cases"] = other_dies + [{"name": "c3_20m.B%d" % (B - 1), "gen": gen, "B": B - 1, "message": "hashTableSize is too small",
                             "note": "reference binary: FATAL ERROR: hashTableSize is too small (the -B %d run of the same set counts %d hashes)"
                                     % (B, man["big_digest_cases"][0]["hash_number"])}]
    else:
        try:
            with open(os.path.join(HERE, "manifest.json")) as f:
                old = json.load(f)
            man["big_digest_cases"] = old.get("big_digest_cases", [])
            man["die_cases"] = old.get("die_cases", [])
        except OSError:
            man["big_digest_cases"] = []
            man["die_cases"] = []
    # reference digests of the weak-scaling workloads of bench.py --gpus N (N x the yeast-scale set, generator v2): the composable
    # checksum of bench.checksum_state over the reference binary's blocks + ClusterHash records (`make_golden.py --scale`)
    if "--scale" in sys.argv:
        sys.path.insert(0, orc.REPO)
        import bench
        man["bench_scale_digests"] = {}
        for n in (2, 4, 8):
            wl = bench.scaled_workload(bench.WORKLOADS["yeast-like-2.5M"], n)
            recs, _first, _total = bench.generate_v2(wl, 1)
            recs.tofile(os.path.join(tmp, "scale.fqb"))
            binary = "hash10x_omp" if os.path.exists(os.path.join(orc.REF_DIR, "hash10x_omp")) else "hash10x"
            r = orc.run_ref((["-t", os.cpu_count()] if binary == "hash10x_omp" else []) + ["-B", wl["B"], "-ct", wl["ct"], "--readFQB", "scale.fqb", "--hashDepthRange", wl["lo"], wl["hi"],
                             "--cluster", 1, 0, "--writeHash", "scale.hash"], tmp, binary=binary, timeout=7200)
            assert r.returncode == 0, r.stderr.decode()
            hf = orc.HashFile(open(os.path.join(tmp, "scale.hash"), "rb").read())
            blocks = np.frombuffer(hf.blocks[1: hf.blocks_max].tobytes(), dtype=np.uint8)      # slot 0 is nobody's block
            ch = np.frombuffer(hf.clushash.tobytes(), dtype=np.uint8)
            cs = bench.checksum_state(blocks, 1, ch, 0)
            man["bench_scale_digests"][str(n)] = {"workload": "yeast-like-2.5M x%d (gen_fqb v2, seed 1)" % n, "B": wl["B"], "input_sha256": orc.sha256(recs.tobytes()),
                                                  "checksum": ["0x%016x" % v for v in cs], "hash_number": hf.hash_number, "blocks_max": hf.blocks_max,
                                                  "sum_nHash": int(hf.blocks["nHash"].sum()), "sum_nSubCluster": int(hf.blocks["nSubCluster"].sum()), "reference_binary": binary}
            os.remove(os.path.join(tmp, "scale.hash")); os.remove(os.path.join(tmp, "scale.fqb"))
    else:
        try:
            with open(os.path.join(HERE, "manifest.json")) as f:
                man["bench_scale_digests"] = json.load(f).get("bench_scale_digests", {})
        except OSError:
            man["bench_scale_digests"] = {}
    # BASELINE configs[2] at its own size (200 M pairs, 1 M barcodes, -B 29): `make_golden.py --full <dir>` takes the reference's .hash of
    # the gen_fqb v2 set from <dir> (c3.fqb + c3.ref.hash, made there by build/gen_fqb -v 2 ... and oracle/_ref/hash10x_omp: 24 + 15 GB, half an hour)
    if "--full" in sys.argv:
        import hashlib
        d = sys.argv[sys.argv.index("--full") + 1]
        gen2 = dict(pairs=200000000, barcodes=1000000, genome=500000000, err=0.001, seed=2, mol=10.0, snp=150, mol_len=50000.0)
        fq = np.memmap(os.path.join(d, "c3.fqb"), dtype=np.uint32, mode="r")
        assert fq.size == 30 * gen2["pairs"]
        digest, info = orc.canonical_file_digest(os.path.join(d, "c3.ref.hash"))
        man["full_digest_cases"] = [{"name": "c3_200m.c", "gen2": gen2, "B": 29, "args": ["--hashDepthRange", "30", "100", "--cluster", "1", "0"],
                                     "input": {"head_sha256": hashlib.sha256(fq[: 30 << 20].tobytes()).hexdigest(), "tail_sha256": hashlib.sha256(fq[-(30 << 20):].tobytes()).hexdigest()},
                                     "sha256": digest, "size": info["size"], "hash_number": info["hash_number"], "blocks_max": info["blocks_max"],
                                     "sum_nHash": info["sum_nHash"], "sum_nSubCluster": info["sum_nSubCluster"],
                                     "reference": "oracle/_ref/hash10x_omp -t 8 -B 29 --readFQB c3.fqb --hashDepthRange 30 100 --cluster 1 0 --writeHash c3.ref.hash "
                                                  "(MALLOC_PERTURB_=255, tcache off); gen: build/gen_fqb -v 2 -P 200000000 -C 1000000 -G 500000000 -e 0.001 -s 2"}]
    else:
        try:
            with open(os.path.join(HERE, "manifest.json")) as f:
                man["full_digest_cases"] = json.load(f).get("full_digest_cases", [])
        except OSError:
            man["full_digest_cases"] = []
    # The 3 Gb strong-scaling workload of bench.py (WORKLOADS["genome3g-300M"]): `make_golden.py --g3 <dir>` takes the reference's .hash of the gen_fqb v2
    # set from <dir> (g3.fqb + g3.ref.hash, made there by bigdata/run_g3.sh: build/gen_fqb -v 2 ... and oracle/_ref/hash10x_omp -t 8 -B 30; 36 + 24 GB,
    # about an hour in the build container) and commits its sizes, the sha256 of its canonical form and the composable checksum of bench.checksum_state
    # over all blocks and ClusterHash records — what `bench.py --workload genome3g-300M --gpus N` sums over its ranks.
    try:
        with open(os.path.join(HERE, "manifest.json")) as f:
            man["strong_digests"] = json.load(f).get("strong_digests", {})
    except OSError:
        man["strong_digests"] = {}
    if "--g3" in sys.argv:                                   # (a set whose files are not in <dir> keeps its committed entry; "--only <workload>" limits the run to one)
        import hashlib
        sys.path.insert(0, orc.REPO)
        import bench
        d = sys.argv[sys.argv.index("--g3") + 1]
        only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
        for name, stem, note in (("genome3g-300M", "g3", "300 M pairs is what the build container's 62 GB of RAM hold of the reference's run (peak RSS 43.8 GB; its own cap, 2^28 - 2 distinct hashes at -B 30, "
                                                         "would allow ~380 M pairs on 3 Gb x 2); 42 minutes: readFQB 28 min, hashDepthRange 6 min, cluster 3161 CPU-s on 8 threads"),
                                 ("genome3g-tenth-30M", "g3t", "the same proportions at 1/10: the functional check of bench.py --scaling strong on test boxes")):
            path = os.path.join(d, stem + ".ref.hash")
            if not os.path.exists(path) or (only and only != name):
                continue
            wl = bench.WORKLOADS[name]
            fq = np.memmap(os.path.join(d, stem + ".fqb"), dtype=np.uint32, mode="r")
            assert fq.size == 30 * wl["pairs"]
            digest, info = orc.canonical_file_digest(path)
            off_depth = 16 + (4 << info["B"]) + 4 + 8 * info["hash_number"]                       # ArrayStruct of hashDepth: its dim at +16
            with open(path, "rb") as f:
                f.seek(off_depth + 16); depth_dim = int.from_bytes(f.read(4), "little")
            off_blocks = off_depth + 32 + 4 * depth_dim + 32                                       # behind the ArrayStruct of clusterBlocks
            blocks = np.fromfile(path, dtype=np.uint8, count=32 * (info["blocks_max"] - 1), offset=off_blocks + 32)      # slot 0 is nobody's block
            cs = bench.checksum_state(blocks, 1, np.zeros(0, dtype=np.uint8), 0)
            off_ch = off_blocks + 32 * info["blocks_dim"]
            for a in range(0, info["sum_nHash"], 1 << 27):
                ch = np.fromfile(path, dtype=np.uint8, count=8 * min(1 << 27, info["sum_nHash"] - a), offset=off_ch + 8 * a)
                part = bench.checksum_state(np.zeros(0, dtype=np.uint8), 0, ch, a)
                cs = [(cs[0] + part[0]) & 0xFFFFFFFFFFFFFFFF, (cs[1] + part[1]) & 0xFFFFFFFFFFFFFFFF]
            man["strong_digests"][name] = {
                "gen2": {k: wl[k] for k in ("pairs", "barcodes", "genome", "err", "seed", "mol", "snp", "mol_len")}, "B": wl["B"],
                "args": ["--hashDepthRange", str(wl["lo"]), str(wl["hi"]), "--cluster", "1", "0"],
                "input": {"head_sha256": hashlib.sha256(fq[: min(fq.size, 30 << 20)].tobytes()).hexdigest(), "tail_sha256": hashlib.sha256(fq[-min(fq.size, 30 << 20):].tobytes()).hexdigest()},
                "sha256": digest, "size": info["size"], "hash_number": info["hash_number"], "blocks_max": info["blocks_max"], "sum_nHash": info["sum_nHash"],
                "sum_nSubCluster": info["sum_nSubCluster"], "checksum": ["0x%016x" % v for v in cs], "note": note,
                "reference": "oracle/_ref/hash10x_omp -t 8 -B %d --readFQB %s.fqb --hashDepthRange %d %d --cluster 1 0 --writeHash %s.ref.hash (MALLOC_PERTURB_=255, tcache off); "
                             "gen: build/gen_fqb -v 2 -P %d -C %d -G %d -e %g -s %d" % (wl["B"], stem, wl["lo"], wl["hi"], stem, wl["pairs"], wl["barcodes"], wl["genome"], wl["err"], wl["seed"])}
    # BASELINE configs[4] on the 3 Gb-shaped sets: --hashDepthRange --cluster, then the crib (both truth haplotypes, gen_fqb -fa), --clusterReport over
    # every barcode, --clusterSplit, --cribSummary and --writeHash of the split state. `make_golden.py --g3split <dir>` runs the reference chain on the sets whose
    # .fqb lies in <dir> (g3t.fqb: from --readFQB, 7 minutes; g3.fqb: from --readHash of g3.ref.hash — the same state, SURVEY B.2 — without --cribSummary, whose ten
    # hash objects over 226 M indices do not fit the container beside the split state) and commits: sha256 + composable checksum of the split .hash, sha256 of the report
    # text (the -o file: COMMAND echoes, crib matches, CLUSTER_SUMMARY / CODE_CLUSTER lines, MIN_POINT_DENSITY, the split line, the summary), and the accuracy figures
    # read off the CODE_CLUSTER lines as integer sums. Outputs go through FIFOs where the disk would not hold them (14 GB of report text at 300 M pairs).
    # Command order: --clusterReport BEFORE --clusterSplit. Behind a split the reference's report reads nGoodHashes[] past its end for the new blocks (hash10x.c:919:
    # the array was sized by the --hashDepthRange before the split) — undefined there, refused here until a new --hashDepthRange.
    try:
        with open(os.path.join(HERE, "manifest.json")) as f:
            man["split_digests"] = json.load(f).get("split_digests", {})
    except OSError:
        man["split_digests"] = {}
    if "--g3split" in sys.argv:
        import threading
        sys.path.insert(0, orc.REPO)
        import bench
        d = sys.argv[sys.argv.index("--g3split") + 1]
        only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
        for name, stem, from_hash, summary, split in (("genome3g-tenth-30M", "g3t", False, True, True), ("genome3g-300M", "g3", True, False, False)):
            # (the 300 M set: report only. Its --clusterSplit makes 145 M blocks, and the reference's arrp() macro multiplies index and element size in int (array.h:81): 32 B x
            #  2^26 blocks wrap, fillHashTable's walk over the new blocks segfaults — seen here, dmesg: "segfault ... in hash10x_omp" right behind the report. No split digest exists.)
            if not os.path.exists(os.path.join(d, stem + ".fqb")) or (only and only != name):
                continue
            wl = bench.WORKLOADS[name]
            if not os.path.exists(os.path.join(d, stem + ".A.fa")):
                subprocess.run([orc.build_gen(), "-v", "2", "-P", str(wl["pairs"]), "-C", str(wl["barcodes"]), "-G", str(wl["genome"]), "-e", str(wl["err"]), "-s", str(wl["seed"]),
                                "--barcodes", "0:0", "-o", "/dev/null", "-fa", os.path.join(d, stem)], check=True)
            fa_sha = {}
            for hap in ("A", "B"):
                sha = __import__("hashlib").sha256()
                with open(os.path.join(d, "%s.%s.fa" % (stem, hap)), "rb") as f:
                    for blk in iter(lambda: f.read(1 << 24), b""):
                        sha.update(blk)
                fa_sha[hap] = sha.hexdigest()
            rep, out = os.path.join(d, stem + ".report.fifo"), os.path.join(d, stem + ".split.fifo")
            for q in (rep, out):
                if os.path.exists(q):
                    os.remove(q)
                os.mkfifo(q)
            head = (["-B", wl["B"], "--readHash", stem + ".ref.hash", "--hashDepthRange", wl["lo"], wl["hi"]] if from_hash else
                    ["-B", wl["B"], "--readFQB", stem + ".fqb", "--hashDepthRange", wl["lo"], wl["hi"], "--cluster", 1, 0])
            tail = ["-o", os.path.basename(rep), "--cribBuild", stem + ".A.fa", stem + ".B.fa", "--clusterReport", 1, 0] + (["--clusterSplit"] if split else []) + (["--cribSummary"] if summary else []) + \
                   ["-o", "-"] + (["--writeHash", os.path.basename(out)] if split else [])
            res = {}
            th = [threading.Thread(target=lambda: res.__setitem__("report", orc.report_digest(rep)))] + \
                 ([threading.Thread(target=lambda: res.__setitem__("hash", orc.canonical_file_digest(out, checksum=bench.checksum_state)))] if split else [])
            for t in th:
                t.start()
            t0 = __import__("time").time()
            r = orc.run_ref(["-t", os.cpu_count()] + head + tail, d, binary="hash10x_omp", timeout=6 * 3600)
            secs = __import__("time").time() - t0
            if r.returncode != 0:                            # (the readers end with the writer's side of the FIFOs; if it never opened them, open and close here)
                for q in (rep, out):
                    try:
                        os.close(os.open(q, os.O_WRONLY | os.O_NONBLOCK))
                    except OSError:
                        pass
            for t in th:
                t.join()
            for q in (rep, out):
                os.remove(q)
            if r.returncode != 0:
                man["split_digests"][name] = {"failed": "the reference ended with code %d after %.0f s: %s" % (r.returncode, secs, r.stderr.decode()[-300:])}
                continue
            rep_sha, rep_info = res["report"]
            if not split:
                man["split_digests"][name] = {
                    "gen2": {k: wl[k] for k in ("pairs", "barcodes", "genome", "err", "seed", "mol", "snp", "mol_len")}, "B": wl["B"], "fasta_sha256": fa_sha,
                    "commands_after_cluster": [str(a) for a in tail], "report_sha256": rep_sha, "report": rep_info, "reference_seconds": round(secs),
                    "no_split_digest": "the reference segfaults in --clusterSplit on this set: 145 M blocks x 32 B overflow the int arithmetic of its arrp() macro (array.h:81) in fillHashTable's walk over the new blocks",
                    "reference": "oracle/_ref/hash10x_omp -t %d %s (MALLOC_PERTURB_=255, tcache off)" % (os.cpu_count(), " ".join(str(a) for a in head + tail))}
                continue
            h_sha, h_info = res["hash"]
            man["split_digests"][name] = {
                "gen2": {k: wl[k] for k in ("pairs", "barcodes", "genome", "err", "seed", "mol", "snp", "mol_len")}, "B": wl["B"], "fasta_sha256": fa_sha,
                "commands_after_cluster": [str(a) for a in tail], "report_sha256": rep_sha, "report": rep_info,
                "sha256": h_sha, "size": h_info["size"], "hash_number": h_info["hash_number"], "blocks_max": h_info["blocks_max"], "sum_nHash": h_info["sum_nHash"],
                "checksum": h_info["checksum"], "reference_seconds": round(secs),
                "reference": "oracle/_ref/hash10x_omp -t %d %s (MALLOC_PERTURB_=255, tcache off)" % (os.cpu_count(), " ".join(str(a) for a in head + tail))}
    man["big_note"] = ("big_digest_cases: BASELINE configs[2] proportions at 1/10 and 1/4 scale, generated by `make_golden.py --big` from "
                       "oracle/_ref (4.5 and 12.5 minutes of the reference); only the GPU tests run them")
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(man, f, indent=1)
    shutil.rmtree(tmp)
    tot = sum(os.path.getsize(os.path.join(HERE, x)) for x in os.listdir(HERE))
    print("golden fixtures written: %d cases, %d digest cases, %.1f KiB total" % (len(man["cases"]), len(man["digest_cases"]), tot / 1024))


if __name__ == "__main__":
    main()
