"""Test helpers: ctypes binding of the CPU oracle (oracle/libh10x_oracle.so), `.hash` parsing and
canonicalisation, and thin wrappers around the real reference binaries in
oracle/_ref/ when they exist (this container only).

TEST INFRASTRUCTURE — never imported by hash10x_amd/.
"""
import ctypes
import gzip
import hashlib
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
GOLDEN = os.path.join(REPO, "tests", "golden")

CLUSHASH = np.dtype([("hash", "<u4"), ("read", "<u2"), ("subCluster", "u1"), ("flags", "u1")])
BLOCK = np.dtype([("nRead", "<u4"), ("nHash", "<u4"), ("nSubCluster", "<u4"), ("clusterParent", "<u4"),
                  ("ptr", "<u8"), ("pointToMin", "<f8")])
assert CLUSHASH.itemsize == 8 and BLOCK.itemsize == 32


def build_oracle():
    subprocess.run(["make", "-C", ORACLE_DIR, "libh10x_oracle.so"], check=True,
                   stdout=subprocess.DEVNULL)
    return os.path.join(ORACLE_DIR, "libh10x_oracle.so")


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build_oracle())
        vp, ci, cu64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64
        L.orc_factor1_from_seed.restype = cu64
        L.orc_factor1_from_seed.argtypes = [ci]
        L.orc_create.restype = vp
        L.orc_create.argtypes = [ci, ci, ci, ci, ctypes.c_char_p, ci]
        L.orc_destroy.argtypes = [vp]
        L.orc_last_error.restype = ctypes.c_char_p
        L.orc_last_error.argtypes = [vp]
        L.orc_mosh_sequence.restype = ci
        L.orc_mosh_sequence.argtypes = [vp, vp, ci, vp, vp, ci]
        L.orc_unpack160.argtypes = [vp, vp]
        L.orc_read_fqb.argtypes = [vp, vp, cu64, ci, ci]
        L.orc_depth_range.argtypes = [vp, ci, ci]
        L.orc_cluster.argtypes = [vp, ci, ci, ci, ci]
        L.orc_cluster_split.argtypes = [vp]
        L.orc_write_hash.argtypes = [vp, ctypes.c_char_p]
        L.orc_read_hash.argtypes = [vp, ctypes.c_char_p]
        L.orc_hash_number.restype = ctypes.c_uint32
        L.orc_hash_number.argtypes = [vp]
        L.orc_hash_index.restype = vp
        L.orc_hash_index.argtypes = [vp]
        L.orc_hash_value.restype = vp
        L.orc_hash_value.argtypes = [vp]
        L.orc_hash_depth.restype = vp
        L.orc_hash_depth.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.orc_blocks.restype = vp
        L.orc_blocks.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.orc_block_clushash.restype = vp
        L.orc_block_clushash.argtypes = [vp, ci]
        L.orc_hash_codes.restype = vp
        L.orc_hash_codes.argtypes = [vp, ctypes.c_uint32]
        L.orc_good_hashes.restype = vp
        L.orc_good_hashes.argtypes = [vp, ci, ctypes.POINTER(ci)]
        L.orc_sum_good_depth.restype = cu64
        L.orc_sum_good_depth.argtypes = [vp, ci, ci, ctypes.POINTER(cu64), ctypes.POINTER(cu64)]
        _lib = L
    return _lib


def _np_from(ptr, dtype, n):
    if not ptr or n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (ctypes.c_char * (np.dtype(dtype).itemsize * n)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


class OracleError(RuntimeError):
    pass


class Oracle:
    """The reference's global state + commands (hash10x.c:85-104, 1158-1279) over the CPU restatement."""

    def __init__(self, k=21, w=31, seed=17, B=20):
        err = ctypes.create_string_buffer(512)
        self.h = lib().orc_create(k, w, seed, B, err, 512)
        if not self.h:
            raise OracleError(err.value.decode())
        self.k, self.w, self.seed, self.B = k, w, seed, B

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.orc_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise OracleError(lib().orc_last_error(self.h).decode())

    def mosh(self, seq_codes):
        s = np.ascontiguousarray(seq_codes, dtype=np.uint8)
        cap = max(len(s), 1)
        hs = np.zeros(cap, dtype=np.uint64)
        ps = np.zeros(cap, dtype=np.int32)
        n = lib().orc_mosh_sequence(self.h, s.ctypes.data, len(s), hs.ctypes.data, ps.ctypes.data, cap)
        return hs[:n], ps[:n]

    def read_fqb(self, records, N=0, chunk=100000):
        r = np.ascontiguousarray(records, dtype=np.uint32).reshape(-1)
        assert r.size % 30 == 0
        self._chk(lib().orc_read_fqb(self.h, r.ctypes.data, r.size // 30, N, chunk))

    def depth_range(self, lo, hi):
        self._chk(lib().orc_depth_range(self.h, lo, hi))

    def cluster(self, code_min=1, code_max=0, threshold=5, threads=1):
        self._chk(lib().orc_cluster(self.h, code_min, code_max, threshold, threads))

    def cluster_split(self):
        self._chk(lib().orc_cluster_split(self.h))

    def write_hash(self, path):
        self._chk(lib().orc_write_hash(self.h, os.fsencode(path)))

    def read_hash(self, path):
        self._chk(lib().orc_read_hash(self.h, os.fsencode(path)))

    # ---- state views
    @property
    def hash_number(self):
        return lib().orc_hash_number(self.h)

    def hash_index(self):
        return _np_from(lib().orc_hash_index(self.h), np.uint32, 1 << self.B)

    def hash_value(self):
        return _np_from(lib().orc_hash_value(self.h), np.uint64, self.hash_number)

    def hash_depth(self):
        dim, mx = ctypes.c_int(), ctypes.c_int()
        p = lib().orc_hash_depth(self.h, ctypes.byref(dim), ctypes.byref(mx))
        return _np_from(p, np.uint32, dim.value), dim.value, mx.value

    def blocks(self):
        dim, mx = ctypes.c_int(), ctypes.c_int()
        p = lib().orc_blocks(self.h, ctypes.byref(dim), ctypes.byref(mx))
        return _np_from(p, BLOCK, dim.value), dim.value, mx.value

    def clushash(self, code):
        b, _, mx = self.blocks()
        assert 0 < code < mx
        return _np_from(lib().orc_block_clushash(self.h, code), CLUSHASH, int(b["nHash"][code]))

    def hash_codes(self, index):
        d, _, _ = self.hash_depth()
        return _np_from(lib().orc_hash_codes(self.h, index), np.uint32, int(d[index]))

    def good_hashes(self, code):
        n = ctypes.c_int()
        p = lib().orc_good_hashes(self.h, code, ctypes.byref(n))
        return _np_from(p, np.uint16, n.value)

    def sum_good_depth(self, code_min=1, code_max=0):
        g, h = ctypes.c_uint64(), ctypes.c_uint64()
        s = lib().orc_sum_good_depth(self.h, code_min, code_max, ctypes.byref(g), ctypes.byref(h))
        return int(s), int(g.value), int(h.value)


# ------------------------------------------------------------------------------------ .hash files
class HashFile:
    """Parsed `.hash` v2 (writer hash10x.c:244-267; layout SURVEY App. B)."""

    def __init__(self, data):
        b = memoryview(data)
        assert bytes(b[0:4]) == b"10XH", "not a 10X hash file"
        self.version = int.from_bytes(b[4:8], "little")
        self.sz_clushash = int.from_bytes(b[8:10], "little")
        self.sz_block = int.from_bytes(b[10:12], "little")
        self.B = int.from_bytes(b[12:16], "little")
        off = 16
        n = 1 << self.B
        self.hash_index = np.frombuffer(b, dtype="<u4", count=n, offset=off)
        off += 4 * n
        self.hash_number = int.from_bytes(b[off:off + 4], "little")
        off += 4
        self.hash_value = np.frombuffer(b, dtype="<u8", count=self.hash_number, offset=off)
        off += 8 * self.hash_number
        self.depth_hdr_off = off
        magic, _, _, dim, size, mx, _ = np.frombuffer(b, dtype="<i4,<i4,<u8,<i4,<i4,<i4,<i4", count=1, offset=off)[0]
        assert magic == 8918274 and size == 4
        self.depth_dim, self.depth_max = int(dim), int(mx)
        off += 32
        self.hash_depth = np.frombuffer(b, dtype="<u4", count=self.depth_dim, offset=off)
        off += 4 * self.depth_dim
        self.blocks_hdr_off = off
        magic, _, _, dim, size, mx, _ = np.frombuffer(b, dtype="<i4,<i4,<u8,<i4,<i4,<i4,<i4", count=1, offset=off)[0]
        assert magic == 8918274 and size == 32
        self.blocks_dim, self.blocks_max = int(dim), int(mx)
        off += 32
        self.blocks_off = off
        self.blocks = np.frombuffer(b, dtype=BLOCK, count=self.blocks_dim, offset=off)
        off += 32 * self.blocks_dim
        tot = int(self.blocks["nHash"][1:self.blocks_max].sum()) if self.blocks_max > 1 else 0
        self.clushash = np.frombuffer(b, dtype=CLUSHASH, count=tot, offset=off)
        self.block_off = np.zeros(max(self.blocks_max, 1) + 1, dtype=np.int64)
        if self.blocks_max > 1:
            self.block_off[2:self.blocks_max + 1] = np.cumsum(self.blocks["nHash"][1:self.blocks_max])
        off += 8 * tot
        self.size = off
        assert off == len(b), (off, len(b))

    def block_clushash(self, code):
        return self.clushash[self.block_off[code]:self.block_off[code + 1]]


def canonical_hash_bytes(data):
    """Zero the heap pointers the reference leaks into the file."""
    hf = HashFile(data)
    b = bytearray(data)
    for hdr in (hf.depth_hdr_off, hf.blocks_hdr_off):
        b[hdr + 8:hdr + 16] = bytes(8)
    blk = np.frombuffer(b, dtype=BLOCK, count=hf.blocks_dim, offset=hf.blocks_off)
    blk["ptr"] = 0
    return bytes(b)


def sha256(data):
    return hashlib.sha256(data).hexdigest()


def read_maybe_gz(path):
    if path.endswith(".gz"):
        with gzip.open(path, "rb") as f:
            return f.read()
    with open(path, "rb") as f:
        return f.read()


def describe_diff(a, b):
    """Human-readable first difference between two canonical .hash images."""
    if len(a) != len(b):
        return "sizes differ: %d vs %d" % (len(a), len(b))
    x, y = HashFile(a), HashFile(b)
    for name in ("B", "hash_number", "depth_dim", "depth_max", "blocks_dim", "blocks_max"):
        if getattr(x, name) != getattr(y, name):
            return "%s: %r vs %r" % (name, getattr(x, name), getattr(y, name))
    for name in ("hash_index", "hash_value", "hash_depth"):
        u, v = getattr(x, name), getattr(y, name)
        d = np.nonzero(u != v)[0]
        if d.size:
            return "%s differs at %d entries, first %d: %r vs %r" % (name, d.size, d[0], u[d[0]], v[d[0]])
    for f in BLOCK.names:
        u, v = x.blocks[f], y.blocks[f]
        d = np.nonzero(~((u == v) | ((u != u) & (v != v))))[0]
        if d.size:
            return "blocks.%s differs at %d blocks, first code %d: %r vs %r" % (f, d.size, d[0], u[d[0]], v[d[0]])
    for f in CLUSHASH.names:
        u, v = x.clushash[f], y.clushash[f]
        d = np.nonzero(u != v)[0]
        if d.size:
            code = int(np.searchsorted(x.block_off, d[0], side="right") - 1)
            return "clusHash.%s differs at %d entries, first entry %d (code %d): %r vs %r" % (
                f, d.size, d[0], code, u[d[0]], v[d[0]])
    return "identical"


# ------------------------------------------------------------------------------------ reference binaries
def have_ref():
    return os.path.exists(os.path.join(REF_DIR, "hash10x"))


def run_ref(args, cwd, binary="hash10x", timeout=600):
    """Run the real reference (oracle/_ref) deterministically (ClusterHash.subCluster/flags are never
    initialised by --readFQB): MALLOC_PERTURB_=255 makes malloc hand out zero-filled memory, and the tcache must be
    off because chunks recycled through it skip that fill (they come back full of the 0xFF free pattern — seen on
    barcodes with ~100 hashes in a 10 M-pair run, where codeClusterReadMerge then indexes trueCluster[255])."""
    env = dict(os.environ, MALLOC_PERTURB_="255", GLIBC_TUNABLES="glibc.malloc.tcache_count=0")
    return subprocess.run([os.path.join(REF_DIR, binary)] + [str(a) for a in args], cwd=cwd, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


# ------------------------------------------------------------------------------------ synthetic input
def build_gen():
    out = os.path.join(REPO, "build", "gen_fqb")
    src = os.path.join(REPO, "hash10x_amd", "tools", "gen_fqb.c")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-o", out, src, "-lm"], check=True)
    return out


def gen_fqb(path, pairs, barcodes, genome, err=0.005, seed=1, mol=10.0, snp=150, mol_len=50000, fa=None):
    """fa = prefix: also writes the two haplotypes as prefix.A.fa / prefix.B.fa (the truth genomes for --cribBuild)."""
    subprocess.run([build_gen(), "-o", str(path), "-P", str(pairs), "-C", str(barcodes), "-G", str(genome),
                    "-e", str(err), "-s", str(seed), "-m", str(mol), "-S", str(snp), "-L", str(mol_len)] + (["-fa", str(fa)] if fa else []),
                   check=True, stderr=subprocess.DEVNULL)
    return np.fromfile(path, dtype=np.uint32).reshape(-1, 30)


def pack_read(seq_codes):
    """2-bit pack a 151-base read like the reference's seqPack (fq2b.c:33-42): 9 full words MSB-first
    and the last 7 bases in the LOW bits of word 9."""
    s = np.asarray(seq_codes, dtype=np.uint32)
    assert s.size == 151
    out = np.zeros(10, dtype=np.uint32)
    for wi in range(9):
        w = 0
        for j in range(16):
            w = (w << 2) | int(s[16 * wi + j])
        out[wi] = w
    w = 0
    for j in range(144, 151):
        w = (w << 2) | int(s[j])
    out[9] = w
    return out


def make_record(barcode_word, read1_tail_codes, read2_codes):
    """One 30-word .fqb record: read1 = barcode(16) + read1_tail(135), read2 = 151 bases."""
    bc = [(barcode_word >> (2 * (15 - i))) & 3 for i in range(16)]
    r1 = np.concatenate([np.array(bc, dtype=np.uint32), np.asarray(read1_tail_codes, dtype=np.uint32)])
    rec = np.zeros(30, dtype=np.uint32)
    rec[0:10] = pack_read(r1)
    rec[15:25] = pack_read(read2_codes)
    rec[10:14] = 0xFFFFFFFF
    rec[14] = 0x007FFFFF
    rec[25:29] = 0xFFFFFFFF
    rec[29] = 0x007FFFFF
    assert rec[0] == barcode_word
    return rec


def build_pairs65k(path):
    """Input of the > 65535-read-pairs edge (ClusterHash.read is U16: hash10x.c:37,180): seeded records of gen_fqb
    regrouped so that the file holds, between ordinary barcodes, BIG = ~84 k distinct pairs under one barcode (more than 65535
    pairs AND more than 65535 unique hashes: ignored by clustering, hash10x.c:748-753) and BIG2 = 66 000 records repeating 400
    pairs followed by ~5 k new pairs (more than 65535 pairs but few hashes: clustered, its late reads stored modulo 2^16 and so
    sharing read numbers with early ones). Needs -c 200000; the last barcode is the file's unhashed trailing block."""
    base = gen_fqb(str(path) + ".base", pairs=330000, barcodes=60, genome=400000, err=0.004, seed=23, mol=6.0, mol_len=20000)
    os.remove(str(path) + ".base")
    w0 = base[:, 0]
    starts = np.flatnonzero(np.r_[True, w0[1:] != w0[:-1]]).tolist() + [len(w0)]
    run = lambda r, n=None: base[starts[r]: (starts[r + 1] if n is None else min(starts[r] + n, starts[r + 1]))]
    parts = [run(r, 1500) for r in range(0, 3)]
    big = np.concatenate([run(r) for r in range(30, 44)]).copy(); big[:, 0] = 0x0BADC0DE
    big[:, 1:] = big[:, 1:]                                   # (bases 16.. of read 1 keep their own barcode's tail: never hashed before base 23)
    parts.append(big)
    parts += [run(r, 1500) for r in range(3, 14)]
    rep = run(45, 400)
    big2 = np.concatenate([np.tile(rep, (165, 1)), run(46), run(47)]).copy(); big2[:, 0] = 0x5EED5EED
    parts.append(big2)
    parts += [run(r, 1500) for r in range(14, 30)]
    parts.append(run(58, 300))                                # trailing block: never hashed
    out = np.concatenate(parts).astype(np.uint32)
    assert len(big) > 70000 and len(big2) > 66000 + 4000
    out.tofile(str(path))
    return out


def digest_input(path, gen):
    """records of a digest case of tests/golden/manifest.json: seeded gen_fqb parameters, or {"builder": name} for a hand-arranged set"""
    if "builder" in gen:
        return globals()["build_" + gen["builder"]](path)
    return gen_fqb(path, **gen)


def leading_options(extra):
    """the option pairs (-ct n, -c n) at the head of a digest case's argument list: they go in front of --readFQB"""
    n = 0
    while n + 1 < len(extra) and str(extra[n]) in ("-ct", "-c"):
        n += 2
    return list(extra[:n])


def canonical_file_digest(path, chunk=1 << 26, checksum=None):
    """sha256 of the CANONICAL form of a .hash file (heap-pointer fields zeroed) without holding it in memory — files of
    10+ GB (full-size BASELINE configs[2]). Also returns the sizes read on the way: (hexdigest, {hash_number, blocks_max, blocks_dim, sum_nHash,
    sum_nSubCluster, size})."""
    sha = hashlib.sha256()
    info = {}
    with open(path, "rb") as raw:
        f = _CountingReader(raw)                             # (bytes counted on the way: the input may be a FIFO the reference writes into, which cannot tell())
        head = f.read(16); sha.update(head)
        assert head[:4] == b"10XH" and int.from_bytes(head[4:8], "little") == 2
        B = int.from_bytes(head[12:16], "little")
        left = 4 << B
        while left:                                          # hashIndex
            b = f.read(min(chunk, left)); sha.update(b); left -= len(b)
        b = f.read(4); sha.update(b); hash_number = int.from_bytes(b, "little")
        left = 8 * hash_number
        while left:                                          # hashValue
            b = f.read(min(chunk, left)); sha.update(b); left -= len(b)
        hdr = bytearray(f.read(32)); hdr[8:16] = bytes(8); sha.update(hdr)      # ArrayStruct of hashDepth: base pointer zeroed
        dim = int.from_bytes(hdr[16:20], "little")
        left = 4 * dim
        while left:
            b = f.read(min(chunk, left)); sha.update(b); left -= len(b)
        hdr = bytearray(f.read(32)); hdr[8:16] = bytes(8); sha.update(hdr)      # ArrayStruct of clusterBlocks
        bdim, bmax = int.from_bytes(hdr[16:20], "little"), int.from_bytes(hdr[24:28], "little")
        sum_hash = sum_sub = 0
        done = 0
        cs = [0, 0]
        while done < bdim:                                   # ClusterBlock[dim]: clusHash pointer zeroed
            nb = min(chunk // 32, bdim - done)
            blk = np.frombuffer(bytearray(f.read(32 * nb)), dtype=BLOCK).copy()
            blk["ptr"] = 0
            lo, hi = max(1 - done, 0), max(min(bmax - done, nb), 0)
            if hi > lo:
                sum_hash += int(blk["nHash"][lo:hi].sum(dtype=np.uint64)); sum_sub += int(blk["nSubCluster"][lo:hi].sum(dtype=np.uint64))
            sha.update(blk.tobytes())
            if checksum is not None and hi > lo:             # bench.checksum_state over blocks 1 .. max - 1 (slot 0 is nobody's block), composable over ranks and slices
                part = checksum(np.frombuffer(blk[lo:hi].tobytes(), dtype=np.uint8), done + lo, np.zeros(0, dtype=np.uint8), 0)
                cs = [(cs[0] + part[0]) & 0xFFFFFFFFFFFFFFFF, (cs[1] + part[1]) & 0xFFFFFFFFFFFFFFFF]
            done += nb
        at = 0
        while True:                                          # ClusterHash records
            b = f.read(chunk)
            if not b:
                break
            sha.update(b)
            if checksum is not None:
                part = checksum(np.zeros(0, dtype=np.uint8), 0, np.frombuffer(b, dtype=np.uint8), at)
                cs = [(cs[0] + part[0]) & 0xFFFFFFFFFFFFFFFF, (cs[1] + part[1]) & 0xFFFFFFFFFFFFFFFF]
            at += len(b) // 8
        info = {"B": B, "hash_number": hash_number, "blocks_max": bmax, "blocks_dim": bdim, "sum_nHash": sum_hash, "sum_nSubCluster": sum_sub, "size": f.n}
        if checksum is not None:
            info["checksum"] = ["0x%016x" % v for v in cs]
    return sha.hexdigest(), info


class _CountingReader:
    def __init__(self, f):
        self.f, self.n = f, 0

    def read(self, k):
        out = bytearray()
        while len(out) < k:                                  # a pipe hands over what it has: ask until k bytes or the end
            b = self.f.read(k - len(out))
            if not b:
                break
            out += b
        self.n += len(out)
        return bytes(out)


def report_digest(path, chunk=1 << 24):
    """sha256 of a report text (the -o file of --cribBuild / --clusterReport / --clusterSplit / --cribSummary) and the accuracy figures read off
    its CODE_CLUSTER lines (hash10x.c:926-946): clusters, clusters located on one chromosome only (no 'OTHER' hashes), mean span in crib position
    units (pos >> 10 of the k-mer offset, hash10x.c:440) of the located ones, reads and hashes per cluster."""
    sha = hashlib.sha256()
    n = pure = located = 0
    span = reads = hashes = size = 0
    tail = b""
    with open(path, "rb") as f:                              # (may be a FIFO)
        while True:
            b = f.read(chunk)
            if not b:
                break
            sha.update(b); size += len(b)
            lines = (tail + b).split(b"\n")
            tail = lines.pop()
            for ln in lines:
                if not ln.startswith(b"    CODE_CLUSTER "):
                    continue
                n += 1
                w = ln.split()
                reads += int(w[4]); hashes += int(w[6])
                if b" OTHER " not in ln:
                    pure += 1
                if b" chr " in ln:
                    k = w.index(b"chr"); located += 1; span += int(w[k + 4])
    return sha.hexdigest(), {"clusters": n, "clusters_without_OTHER": pure, "clusters_located": located, "sum_span": span, "sum_reads": reads, "sum_hashes": hashes,
                             "purity": (pure / n) if n else None, "mean_span": (span / located) if located else None, "size": size}


def slice_digest(h, step=1 << 25):
    """sha256 + size of the canonical .hash a single-GPU state would be written as (hash10x.c:244-267), assembled from h10x_export_slice in file order without a
    file: header, hashIndex[2^B], hashNumber, hashValue[], ArrayStruct + hashDepth[dim], ArrayStruct + ClusterBlock[dim], ClusterHash records. The Array dims
    follow arrayExtend (array.c:144-170) through the host library's own helper."""
    import hash10x_amd
    hip, host = hash10x_amd.load_native()
    z = h.sizes()
    sha = hashlib.sha256(); size = 0

    def put(b):
        nonlocal size
        sha.update(b); size += len(b)
    put(b"10XH" + (2).to_bytes(4, "little") + (8).to_bytes(2, "little") + (32).to_bytes(2, "little") + int(z["B"]).to_bytes(4, "little"))

    def table(t, first, count, zero_ptr=False):
        for a in range(0, count, step):
            n = min(step, count - a)
            b = h.export_slice(t, first + a, n)
            if zero_ptr:
                b = np.frombuffer(bytes(b), dtype=BLOCK).copy(); b["ptr"] = 0
            put(b.tobytes())
    table(0, 0, 1 << z["B"])
    put(int(z["hashNumber"]).to_bytes(4, "little"))
    table(1, 0, z["hashNumber"])
    depth_dim = host.h10x_host_array_dim(1 << 20, 4, z["hashNumber"] - 1)
    blocks_dim = host.h10x_host_array_dim(1200, 32, z["nBlocks"] - 1)
    hdr = np.zeros(1, dtype="<i4,<i4,<u8,<i4,<i4,<i4,<i4")
    hdr[0] = (8918274, 0, 0, depth_dim, 4, z["hashNumber"], 0); put(hdr.tobytes())
    table(2, 0, z["hashNumber"]); put(bytes(4 * (depth_dim - z["hashNumber"])))
    hdr[0] = (8918274, 0, 0, blocks_dim, 32, z["nBlocks"], 0); put(hdr.tobytes())
    table(3, 0, z["nBlocks"], zero_ptr=True); put(bytes(32 * (blocks_dim - z["nBlocks"])))
    table(4, 0, z["nClusHash"])
    return sha.hexdigest(), size
