"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol that
include/h10x.h declares, fails loudly without a GPU (no fallback), and the C host helpers (Array dim
rules, readFQB chunk replay, shard partition) behave like the reference. No compute calls here."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import orc

REPO = orc.REPO


@pytest.fixture(scope="module")
def native():
    import hash10x_amd
    try:
        return hash10x_amd.load_native()
    except hash10x_amd.Hash10xError:
        import __graft_entry__
        __graft_entry__.build()
        return hash10x_amd.load_native()


def _declared(header):
    txt = open(os.path.join(REPO, header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(h10x_[a-z0-9_A-Z]+)\s*\(", txt)))


def test_c_abi_exports_every_declared_symbol(native):
    hip, host = native
    names = _declared("include/h10x.h")
    assert len(names) >= 20
    for n in names:
        assert hasattr(hip, n), "libh10x_hip.so does not export %s" % n
    for n in _declared("hash10x_amd/host/h10x_host.h"):
        assert hasattr(host, n), "libh10x_host.so does not export %s" % n
    import hash10x_amd
    assert hip.h10x_abi_version() == hash10x_amd.ABI_VERSION == 3


def test_no_cpu_fallback(native):
    import hash10x_amd
    if hash10x_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(hash10x_amd.Hash10xError, match="no HIP device available"):
        hash10x_amd.Hash10x(B=20).read_fqb(np.zeros(60, np.uint32))
    r = subprocess.run([os.path.join(REPO, "bin", "hash10x-amd"), "-B", "20", "--readFQB", os.path.join(orc.GOLDEN, "tiny.fqb")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 255 and b"FATAL ERROR: no HIP device available" in r.stderr      # die(): exit(-1)


def test_factor1_matches_glibc_random_kat(native):
    assert native[0].h10x_factor1_from_seed(17) == 0x49308BB9003CB3AD       # SURVEY KAT-1
    assert native[0].h10x_factor1_from_seed(17) == orc.lib().orc_factor1_from_seed(17)
    assert native[0].h10x_factor1_from_seed(1) == orc.lib().orc_factor1_from_seed(1)


def test_array_dim_rules(native):
    dim = native[1].h10x_host_array_dim
    # hashDepth: arrayCreate(1<<20, U32)
    assert dim(1 << 20, 4, (1 << 20) - 1) == 1 << 20
    assert dim(1 << 20, 4, 1 << 20) == 1 << 21
    assert dim(1 << 20, 4, (1 << 21)) == (1 << 21) + 2098176
    assert dim(1 << 20, 4, (1 << 21) + 2098176) == (1 << 21) + 2 * 2098176
    # clusterBlocks: arrayCreate(1200, ClusterBlock): doubles to 307200, then +263168 per step
    assert dim(1200, 32, 1199) == 1200 and dim(1200, 32, 1200) == 2400 and dim(1200, 32, 10000) == 19200
    assert dim(1200, 32, 307199) == 307200 and dim(1200, 32, 307200) == 307200 + 263168
    # and against what the oracle (pinned to the reference) reports for real data
    for name in ("small.hash.gz", "tiny.hash.gz"):
        hf = orc.HashFile(orc.read_maybe_gz(os.path.join(orc.GOLDEN, name)))
        assert dim(1 << 20, 4, hf.hash_number - 1) == hf.depth_dim
        assert dim(1200, 32, hf.blocks_max - 1) == hf.blocks_dim


def test_chunk_replay_matches_reference_rule(native):
    chk = native[1].h10x_host_check_chunks
    recs = np.fromfile(os.path.join(orc.GOLDEN, "tiny.fqb"), dtype=np.uint32)
    n = recs.size // 30
    err = ctypes.create_string_buffer(256)
    assert chk(recs.ctypes.data, n, 0, 100000, err, 256) == n
    assert chk(recs.ctypes.data, n, 9, 100000, err, 256) == 9
    assert chk(recs.ctypes.data, n, 0, 3, err, 256) == -1 and err.value == b"chunkSize too small"   # barcode C holds 4 pairs
    assert chk(recs.ctypes.data, n, 0, 5, err, 256) == n
    for chunk in (3, 4, 5, 7):                          # same verdict as the oracle's replay of hash10x.c:202-223
        o = orc.Oracle(B=20)
        try:
            o.read_fqb(recs, 0, chunk)
            ok = True
        except orc.OracleError:
            ok = False
        assert (chk(recs.ctypes.data, n, 0, chunk, err, 256) >= 0) == ok


def test_partition_cuts_on_barcode_boundaries(native):
    part = native[1].h10x_host_partition
    recs = np.frombuffer(orc.read_maybe_gz(os.path.join(orc.GOLDEN, "small.fqb.gz")), dtype=np.uint32).reshape(-1, 30)
    n = recs.shape[0]
    flat = np.ascontiguousarray(recs).reshape(-1)
    for parts in (1, 2, 3, 8):
        cut = (ctypes.c_uint64 * (parts + 1))()
        assert part(flat.ctypes.data, n, parts, cut) == 0
        c = list(cut)
        assert c[0] == 0 and c[-1] == n and c == sorted(c)
        for x in c[1:-1]:
            assert x == n or recs[x, 0] != recs[x - 1, 0]
        assert max(np.diff(c)) <= n / parts + 400       # balanced up to one barcode


@pytest.mark.skipif(not os.path.exists(os.path.join(orc.REF_DIR, "fq2b")), reason="oracle/_ref/fq2b not present")
@pytest.mark.parametrize("whitelist", [False, True])
def test_fq2b_matches_reference_bytes_and_stats(tmp_path, whitelist):
    """bin/fq2b-amd against the reference's fq2b on the same gzipped FASTQ pair: identical .fqb bytes (2-bit packing with
    the unjustified tail word, quality bits, N -> A) and identical statistics, with and without the 10x whitelist
    (one-mismatch correction, dropped pairs)."""
    import gzip
    rng = np.random.default_rng(5)
    wl = ["".join(rng.choice(list("ACGT"), 16)) for _ in range(40)]
    # whitelist lines that collide in the reference's byte table (later lines win there, fq2b.c:71-94): neighbours one and
    # two substitutions apart, listed before AND after the barcode they shadow, and a repeated line
    def sub(b, pos, c):
        return b[:pos] + c + b[pos + 1:]
    other = lambda ch: "ACGT"[("ACGT".index(ch) + 1) % 4]
    wl += [sub(wl[3], 5, other(wl[3][5])), sub(sub(wl[4], 2, other(wl[4][2])), 9, other(wl[4][9])), wl[6]]
    wl = [sub(wl[8], 15, other(wl[8][15]))] + wl
    n = 900
    with gzip.open(tmp_path / "r1.fq.gz", "wt") as f1, gzip.open(tmp_path / "r2.fq.gz", "wt") as f2:
        for i in range(n):
            bc = list(wl[rng.integers(len(wl))])
            u = rng.random()
            if u < 0.3:
                bc[rng.integers(16)] = "ACGT"[rng.integers(4)]                     # at most one mismatch (or none)
            elif u < 0.4:
                bc = list(rng.choice(list("ACGT"), 16))                              # random: usually dropped
            elif u < 0.45:
                bc[rng.integers(16)] = "N"
            s1 = "".join(bc) + "".join(rng.choice(list("ACGTN"), 135, p=[.24, .24, .24, .24, .04]))
            s2 = "".join(rng.choice(list("acgtACGTN"), 151))
            q1 = "".join(chr(int(c)) for c in rng.integers(35, 75, 151))
            q2 = "".join(chr(int(c)) for c in rng.integers(35, 75, 151))
            f1.write("@read%d 1:N:0\n%s\n+\n%s\n" % (i, s1, q1))
            f2.write("@read%d %s\n%s\n+\n%s\n" % (i, "1:N:0" if i != 7 else "2:N:0", s2, q2))
    (tmp_path / "wl.txt").write_text("\n".join(wl) + "\n")
    opts = ["-10x", "wl.txt"] if whitelist else []
    ref = subprocess.run([os.path.join(orc.REF_DIR, "fq2b")] + opts + ["-o", "ref.fqb", "r1.fq.gz", "r2.fq.gz"], cwd=tmp_path, stderr=subprocess.PIPE)
    got = subprocess.run([os.path.join(REPO, "bin", "fq2b-amd")] + opts + ["-o", "hip.fqb", "r1.fq.gz", "r2.fq.gz"], cwd=tmp_path, stderr=subprocess.PIPE)
    assert ref.returncode == 0 and got.returncode == 0, (ref.stderr, got.stderr)
    a, b = (tmp_path / "ref.fqb").read_bytes(), (tmp_path / "hip.fqb").read_bytes()
    assert len(a) % 120 == 0 and len(a) > 0 and a == b
    assert got.stderr == ref.stderr and b"read pairs 151 + 151 bp packed in 30 word records" in got.stderr
    if whitelist:
        assert len(a) < n * 120 and b"were error corrected" in got.stderr


def test_rolling_products_equal_the_multiplies_of_hashFunc():
    """The arithmetic mosh_lds_kernel (csrc/stage_a.hip, FAST path) relies on, checked with plain integers: with F = factor1 and b(i) base i of
    a run, the product of the NEXT forward word is 4 x this product + [b(j+k) F - b(j) (F << 2k)] (mod 2^64), and the product of the PREVIOUS
    reverse-complement word (seqhash.c:75) is 4 x this one + [(3 - b(j)) F - (3 - b(j+k)) (F << 2k)] — the two bracketed terms are what the
    kernel's 16-entry LDS table holds for the index b(j) << 2 | b(j+k). hashFunc itself (seqhash.c:58-59) = (word * F mod 2^64) >> (64 - 2k)."""
    import random
    rng = random.Random(5)
    M = (1 << 64) - 1
    for k in (17, 21, 24, 30):
        F = rng.getrandbits(64) | 1
        bases = [rng.randrange(4) for _ in range(k + 40)]
        top = (F << (2 * k)) & M
        fwd_term = {(a << 2) | b: (b * F - a * top) & M for a in range(4) for b in range(4)}
        rev_term = {(a << 2) | b: ((3 - a) * F - (3 - b) * top) & M for a in range(4) for b in range(4)}

        def word(j):                                          # bases j .. j+k-1, base j in the top two bits
            w = 0
            for i in range(k):
                w = (w << 2) | bases[j + i]
            return w

        def revcomp(j):                                       # complement of base j in the lowest two bits (advanceHashRC's word)
            w = 0
            for i in range(k):
                w |= (3 - bases[j + i]) << (2 * i)
            return w

        n = 8
        pf = word(0) * F & M
        for j in range(n - 1):                                # forward walk
            pf = (4 * pf + fwd_term[(bases[j] << 2) | bases[j + k]]) & M
            assert pf == word(j + 1) * F & M
        pr = revcomp(n - 1) * F & M
        for j in range(n - 2, -1, -1):                        # backward walk
            pr = (4 * pr + rev_term[(bases[j] << 2) | bases[j + k]]) & M
            assert pr == revcomp(j) * F & M
        # min before the shift = the shift of the min (the shift is monotone): what the kernel tests for divisibility by w
        for j in range(n):
            a, b = word(j) * F & M, revcomp(j) * F & M
            assert min(a, b) >> (64 - 2 * k) == min(a >> (64 - 2 * k), b >> (64 - 2 * k))


REFHASH_GOLDEN = {(3000000, 400000, 1): 399806, (6000000, 1500000, 2): 1472343, (5000000, 3000000, 3): 2439655}     # oracle/_ref/refhash (the reference's hash.c), build container


def _splitmix_keys(n, distinct, seed):
    x = (np.uint64(seed) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
    with np.errstate(over="ignore"):
        z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (1 + (z % np.uint64(distinct))).astype(np.int32)


@pytest.mark.parametrize("case", sorted(REFHASH_GOLDEN))
def test_refhash_restatement_counts_like_the_reference(native, case):
    """--cribSummary's "distinct hashes" figures are hashCount() of the reference's HASH object (hash.c), which over-counts once a table has doubled (its doubling
    re-inserts keys with one stride for all: hash.c:123-160) — by an amount that depends on the order of insertion. host/h10x_host.c restates it (RefHash); here the
    restatement meets the reference's own hash.c (oracle/_ref/refhash, a driver of ours around it) on seeded key sequences below and beyond the first doubling: golden
    counts from the build container, and the live binary where it is present. The third case counts 2 439 655 keys where 2 433 166 are distinct."""
    import ctypes
    import subprocess
    _hip, host = native
    host.h10x_host_refhash_count.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    n, distinct, seed = case
    keys = _splitmix_keys(n, distinct, seed)
    got = host.h10x_host_refhash_count(keys.ctypes.data, n)
    assert got == REFHASH_GOLDEN[case]
    if case == (5000000, 3000000, 3):
        assert got > np.unique(keys).size == 2433166
    exe = os.path.join(orc.REF_DIR, "refhash")
    if os.path.exists(exe):
        assert int(subprocess.run([exe, str(n), str(distinct), str(seed)], stdout=subprocess.PIPE, check=True).stdout) == got
