"""Parity of the HIP path (through the C ABI / C session layer) against the reference's golden
vectors and against the CPU oracle on seeded inputs. Needs a real MI355X: `pytest -m gpu`.

Bar: bit-exact canonical .hash (integer/byte work; the one floating-point field, pointToMin, is an
ordered IEEE double sum and is compared bit-for-bit too)."""
import json
import os
import sys

import numpy as np
import pytest

import orc
from driver import run_commands

pytestmark = pytest.mark.gpu

with open(os.path.join(orc.GOLDEN, "manifest.json")) as _f:
    MAN = json.load(_f)


def _factory(**opts):
    import hash10x_amd

    def make(k, w, r, B):
        h = hash10x_amd.Hash10x(k=k, w=w, r=r, B=B)
        for n, v in opts.items():
            h.set_option(n, v)
        return h
    return make


def test_native_library_is_the_path():
    import hash10x_amd
    hip, host = hash10x_amd.load_native()
    assert hip.h10x_abi_version() == hash10x_amd.ABI_VERSION == 3
    assert hash10x_amd.device_count() >= 1
    assert hip.h10x_factor1_from_seed(17) == 0x49308BB9003CB3AD
    # the library that is mapped was built from the sources of this tree (the .so files are git-ignored and ship prebuilt)
    import glob, hashlib
    csrc = os.path.join(orc.REPO, "hash10x_amd", "csrc")
    names = sorted(os.path.basename(f) for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")))
    sha = hashlib.sha256()
    for f in [os.path.join(csrc, n) for n in names] + [os.path.join(orc.REPO, "include", "h10x.h")]:
        sha.update(open(f, "rb").read())
    assert hash10x_amd.build_id() == "src:" + sha.hexdigest()[:16], "libh10x_hip.so is stale: rebuild with make -C hash10x_amd/csrc"


@pytest.mark.parametrize("case", MAN["cases"], ids=[c["name"] for c in MAN["cases"]])
def test_hip_matches_reference_golden(case, workdir):
    workdir.need(case["input"])
    args = list(case["args"])
    run_commands(_factory(), args, workdir.path)
    got = open(workdir.file(args[-1]), "rb").read()
    exp = orc.read_maybe_gz(os.path.join(orc.GOLDEN, case["output"]))
    assert got == exp, orc.describe_diff(got, exp)


@pytest.mark.parametrize("case", MAN["digest_cases"], ids=[c["name"] for c in MAN["digest_cases"]])
def test_hip_matches_reference_digests(case, workdir):
    recs = orc.digest_input(workdir.file(case["name"] + ".fqb"), case["gen"])
    assert orc.sha256(recs.tobytes()) == case["input_sha256"]
    extra = list(case["args"])
    pre = orc.leading_options(extra)
    args = ["-B", case["B"]] + pre + ["--readFQB", case["name"] + ".fqb"] + extra[len(pre):] + ["--writeHash", "out.hash"]
    run_commands(_factory(), args, workdir.path)
    got = open(workdir.file("out.hash"), "rb").read()
    if orc.sha256(got) != case["sha256"]:
        # recompute the expectation with the oracle to say where it differs
        run_commands(lambda k, w, r, B: orc.Oracle(k, w, r, B), args[:-1] + ["exp.hash"], workdir.path)
        exp = open(workdir.file("exp.hash"), "rb").read()
        assert got == exp, orc.describe_diff(got, exp)
        raise AssertionError("digest mismatch but oracle agrees?!")


def _against_oracle(workdir, recs_name, args_tail, k=21, w=31, r=17, B=20, **opts):
    base = ["-k", k, "-w", w, "-r", r, "-B", B] + args_tail
    run_commands(_factory(**opts), base + ["--writeHash", "hip.hash"], workdir.path)
    run_commands(lambda k_, w_, r_, B_: orc.Oracle(k_, w_, r_, B_), base + ["--writeHash", "orc.hash"], workdir.path)
    got = open(workdir.file("hip.hash"), "rb").read()
    exp = open(workdir.file("orc.hash"), "rb").read()
    assert got == exp, orc.describe_diff(got, exp)
    return orc.HashFile(got)


def test_global_mosh_path_forced(workdir):
    """Every barcode through the global-memory mosh path (LDS table capped below what any block needs)."""
    workdir.need("small.fqb.gz")
    hf = _against_oracle(workdir, "small.fqb", ["--readFQB", "small.fqb", "--hashDepthRange", 3, 14, "-ct", 2, "--cluster", 1, 0],
                         stage_a_max_slots=256)
    assert hf.blocks["nSubCluster"].sum() > 0


@pytest.mark.parametrize("budget", [2048, 24 * 1024])
def test_cluster_hbm_scratch_and_big_lds_paths(workdir, budget):
    """cluster_kernel on the per-workgroup HBM scratch (LDS budget too small for any barcode) and with a
    budget that splits barcodes between the LDS classes."""
    orc.gen_fqb(workdir.file("x.fqb"), 30000, 150, 300000, 0.003, 41, 4.0, 150, 6000)
    hf = _against_oracle(workdir, "x.fqb", ["-ct", 3, "--readFQB", "x.fqb", "--hashDepthRange", 4, 30, "--cluster", 1, 0],
                         cluster_lds_budget=budget)
    assert hf.blocks["nSubCluster"].sum() > 0


@pytest.mark.parametrize("mode,cap,packed", [(2, 0, 1), (2, 64, 1), (4, 0, 0), (4, 1500, 0), (4, 200, 0), (4, 64, 0), (4, 0, 1), (4, 1500, 1), (4, 200, 1), (4, 64, 1)])
def test_cluster_ranked_and_hashed_first_tables(workdir, mode, cap, packed):
    """Placements of first[] for data sets with many barcodes, forced on a small set: mode 2 = ranked (presence bitmap +
    popcount prefix, first[] sized by the barcodes present), mode 4 = translated (round 4: one pass turns every list entry into the 16-bit slot number of its barcode in
    such a table, the list loop then runs on those handles; cap 200: the first table is closed at 175 of the ~300 barcodes and the
    rest goes into a second one; packed = round 5's form of it: the ranks ascend in list length, so four lists of up to 16 entries, two of
    up to 32 share a wave instruction — the depth range 4-40 puts lists into the classes Q, H and F). A small cap makes the tables overflow:
    the blocks are re-run with the whole LDS of a CU and, when that fails too, with first[] dense on an HBM slot."""
    orc.gen_fqb(workdir.file("x.fqb"), 60000, 300, 400000, 0.003, 43, 4.0, 150, 6000)
    opts = dict(cluster_first_global=mode, cluster_tr_packed=packed)
    if cap:
        opts["cluster_first_cap"] = cap
    hf = _against_oracle(workdir, "x.fqb", ["-ct", 3, "--readFQB", "x.fqb", "--hashDepthRange", 4, 40, "--cluster", 1, 0], **opts)
    assert hf.blocks["nSubCluster"].sum() > 0


@pytest.mark.parametrize("packed,class_t", [(0, 1), (1, 1), (1, 0)])
def test_cluster_translated_long_lists(workdir, packed, class_t):
    """The translated placement on lists of every class of the packed form at once: a small genome under 400 barcodes puts hashes into 4 .. 345
    barcodes, so one block holds lists of up to 16, 32, 64, 96, 128 entries (Q, H, F, T, D: four, two, one list per wave instruction, two lists in
    three chunks, one list in two chunks; class_t = 0: the lists of 65 .. 96 entries as class D) and beyond (X: the per-list loop, incl. lists of
    256 entries and more — row_mode_long)."""
    orc.gen_fqb(workdir.file("x.fqb"), 100000, 400, 30000, 0.003, 45, 12.0, 150, 6000)
    hf = _against_oracle(workdir, "x.fqb", ["-ct", 3, "--readFQB", "x.fqb", "--hashDepthRange", 4, 400, "--cluster", 1, 0], cluster_first_global=4, cluster_tr_packed=packed,
                         cluster_tr_class_t=class_t)
    d = hf.hash_depth[1: hf.hash_number]
    assert ((d > 64) & (d <= 96)).sum() > 16                   # class T has work to do (a few lists per block here: the next test is about it)
    assert hf.blocks["nSubCluster"].sum() > 0
    assert hf.hash_depth[1: hf.hash_number].max() > 256


@pytest.mark.parametrize("opts", [{}, {"cluster_tr_class_t": 0}, {"cluster_first_cap": 120}, {"cluster_lds_budget": 24 * 1024}, {"cluster_first_cap": 64}],
                         ids=["default", "as_class_D", "second_tables", "few_list_waves", "tables_overflow"])
def test_cluster_translated_class_t(workdir, opts):
    """Class T of the packed translated placement (lists of 65 .. 96 entries two to a unit: a whole chunk each and one chunk shared by their tails — four lists in
    five on BASELINE configs[2]): 200 barcodes that each cover ~40 % of a 30 kb genome put 400 hashes into 65 .. 96 barcodes (and a few hundred each into the
    classes Q, H, F, D), so most blocks hold dozens of such lists, an odd number of them in half of the blocks. Also with a capped first table (second tables; the
    ranks left open by the list loop read their handles again from the shared chunk), few list-loop waves, and tables that overflow (whole-CU class, HBM slot)."""
    orc.gen_fqb(workdir.file("x.fqb"), 16000, 200, 30000, 0.003, 45, 12.0, 150, 6000)
    hf = _against_oracle(workdir, "x.fqb", ["-ct", 3, "--readFQB", "x.fqb", "--hashDepthRange", 4, 130, "--cluster", 1, 0], cluster_first_global=4, **opts)
    d = hf.hash_depth[1: hf.hash_number]
    assert ((d > 64) & (d <= 96)).sum() > 300 and ((d > 96) & (d <= 128)).sum() > 100 and hf.blocks["nSubCluster"].sum() > 0
    if not opts:                                              # and on three ranks with the lists padded to 8-entry units (the tail chunk of a class-T unit reads behind a padded list's first 64 entries)
        recs = np.fromfile(workdir.file("x.fqb"), dtype=np.uint32)
        o = orc.Oracle(B=20)
        o.read_fqb(recs); o.depth_range(4, 130); o.cluster(1, 0, 3)
        o.write_hash(workdir.file("orc3.hash"))
        _run_sharded(recs, 3, 20, 4, 130, 3, workdir.file("sh.hash"), opts={"cluster_first_global": 4, "shard_row_shift": 3})
        got = open(workdir.file("sh.hash"), "rb").read(); exp = open(workdir.file("orc3.hash"), "rb").read()
        assert got == exp, orc.describe_diff(got, exp)


def test_cluster_first_table_on_hbm_slots(workdir):
    """Hybrid placement used when the data set has too many barcodes for first[] to sit in LDS: first[] on a
    per-workgroup HBM slot (atomics + L1-bypassing loads), per-rank arrays in LDS."""
    orc.gen_fqb(workdir.file("x.fqb"), 60000, 300, 400000, 0.003, 43, 4.0, 150, 6000)
    hf = _against_oracle(workdir, "x.fqb", ["-ct", 3, "--readFQB", "x.fqb", "--hashDepthRange", 4, 40, "--cluster", 1, 0],
                         cluster_first_global=1)
    assert hf.blocks["nSubCluster"].sum() > 0


def test_more_than_65535_barcodes(workdir):
    """70 000 barcodes: barcode numbers beyond 16 bits in the lists, 64-bit (depth, position) keys for the good lists (the
    depth bound no longer fits 16 bits), ranked placement of first[] by itself, and the hashed one forced."""
    orc.gen_fqb(workdir.file("x.fqb"), 840000, 70000, 150000, 0.002, 91, 1.0, 150, 3000)
    hf = _against_oracle(workdir, "x.fqb", ["-ct", 1, "--readFQB", "x.fqb", "--hashDepthRange", 3, 200, "--cluster", 1, 0], B=23)
    assert hf.blocks["nSubCluster"].sum() > 10000 and hf.blocks_max > 65535
    _against_oracle(workdir, "x.fqb", ["-ct", 1, "--readFQB", "x.fqb", "--hashDepthRange", 3, 200, "--cluster", 1, 0], B=23, cluster_first_global=3)
    _against_oracle(workdir, "x.fqb", ["-ct", 1, "--readFQB", "x.fqb", "--hashDepthRange", 3, 200, "--cluster", 1, 0], B=23, cluster_first_global=4)   # lists of up to 199 entries: four chunks of handles
    _against_oracle(workdir, "x.fqb", ["-ct", 1, "--readFQB", "x.fqb", "--hashDepthRange", 3, 200, "--cluster", 1, 0], B=23, cluster_first_global=4, cluster_lds_budget=24 * 1024)   # tables compacted, few list-loop waves
    _against_oracle(workdir, "x.fqb", ["-ct", 1, "--readFQB", "x.fqb", "--hashDepthRange", 3, 200, "--cluster", 1, 0], B=23, cluster_first_global=4, cluster_first_cap=1024)   # second tables, some of them too small as well
    # and sharded over 4 ranks (global barcode numbers beyond 16 bits in every rank's lists)
    exp = open(workdir.file("orc.hash"), "rb").read()
    recs = np.fromfile(workdir.file("x.fqb"), dtype=np.uint32)
    _run_sharded(recs, 4, 23, 3, 200, 1, workdir.file("sh.hash"))
    got = open(workdir.file("sh.hash"), "rb").read()
    assert got == exp, orc.describe_diff(got, exp)


def test_cluster_tiny_set_front_queue_only(workdir):
    """60 read pairs in 15 barcodes, depth range [3,4): the mean number of good hashes per barcode is 0, so every barcode with
    any counts as 'large' and sits in the front queue while the ordinary queue is empty (a null first[] slot in the hybrid
    placement, found by tests/soak.py)."""
    orc.gen_fqb(workdir.file("x.fqb"), 60, 15, 40000, 0.001, 1507, 4.0, 150, 2500)
    for mode in (0, 1, 2, 3, 4):
        _against_oracle(workdir, "x.fqb", ["-ct", 5, "--readFQB", "x.fqb", "--hashDepthRange", 3, 4, "--cluster", 1, 0],
                        k=24, w=32, r=9, B=21, cluster_first_global=mode)
        _against_oracle(workdir, "x.fqb", ["-ct", 1, "--readFQB", "x.fqb", "--hashDepthRange", 2, 4, "--cluster", 1, 0],
                        k=24, w=32, r=9, B=21, cluster_first_global=mode)


@pytest.mark.parametrize("k,w,r", [(21, 31, 17), (16, 5, 3), (24, 31, 5), (25, 31, 17), (31, 7, 1), (11, 32, 9), (21, 1, 3), (19, 29, 7), (17, 2, 5), (30, 31, 2)])
def test_other_hashers(workdir, k, w, r):
    """k > 24 cannot pack (hash, read) into 64 bits and takes the global path; w != 31 the generic modulo. w = 1 / 2: every (other)
    k-mer is a mosh — each wave's survivor queue is emptied after every slot and the LDS set overflows into the global path; w = 29: as
    many survivors as the table still holds; k = 17 and 30: the ends of the range that hashes on 32-bit halves."""
    orc.gen_fqb(workdir.file("x.fqb"), 3000, 20, 40000, 0.004, 100 + k, 3.0, 150, 2500)
    _against_oracle(workdir, "x.fqb", ["--readFQB", "x.fqb"], k=k, w=w, r=r, B=21)


@pytest.mark.parametrize("barcodes", [7, 15])
def test_packed_entries_at_the_width_limit(workdir, barcodes):
    """k = 31, w = 5: 60 key bits. With 7 barcodes key and block number fill 63 bits (packed one-word entries, the widest the index
    build packs), with 15 they would fill 64 and the build must take separate arrays: rocPRIM's keys-only sort mis-sorts a bit range
    that starts above bit 0 and ends at bit 64 (tests/soak.py case; scratch/sort_bits_check.hip)."""
    orc.gen_fqb(workdir.file("x.fqb"), 60 * barcodes // 15 + 40, barcodes, 100000, 0.003, 1076, 3.0, 150, 8000)
    _against_oracle(workdir, "x.fqb", ["-ct", 2, "--readFQB", "x.fqb", "--hashDepthRange", 3, 33, "--cluster", 1, 0], k=31, w=5, r=17, B=23)
    orc.gen_fqb(workdir.file("y.fqb"), 30000, barcodes, 200000, 0.003, 1077, 3.0, 150, 8000)          # enough entries for the merge-sort sizes
    _against_oracle(workdir, "y.fqb", ["-ct", 2, "--readFQB", "y.fqb", "--hashDepthRange", 3, 33, "--cluster", 1, 0], k=31, w=5, r=17, B=25)


@pytest.mark.parametrize("seed,pairs,barcodes,genome,mol,mol_len,lo,hi,ct", [
    (41, 30000, 150, 300000, 4.0, 6000, 4, 30, 3),
    (42, 50000, 60, 150000, 3.0, 5000, 6, 60, 5),
    (43, 20000, 400, 100000, 2.0, 3000, 3, 40, 2),
])
def test_random_sets_end_to_end(workdir, seed, pairs, barcodes, genome, mol, mol_len, lo, hi, ct):
    orc.gen_fqb(workdir.file("x.fqb"), pairs, barcodes, genome, 0.003, seed, mol, 150, mol_len)
    hf = _against_oracle(workdir, "x.fqb", ["-ct", ct, "--readFQB", "x.fqb", "--hashDepthRange", lo, hi, "--cluster", 1, 0])
    assert hf.blocks["nSubCluster"].sum() > 0
    # --clusterSplit, then cluster the split barcodes again (hashCodes rebuilt for the new block numbering)
    hf2 = _against_oracle(workdir, "x.fqb", ["-ct", ct, "--readFQB", "x.fqb", "--hashDepthRange", lo, hi, "--cluster", 1, 0, "--clusterSplit",
                                             "--hashDepthRange", lo, hi, "--cluster", 1, 0])
    assert hf2.blocks_max > hf.blocks_max


@pytest.mark.parametrize("opts", [dict(index_no_pack=1), dict(cluster_narrow_first=1), dict(cluster_narrow_first=5),
                                  dict(cluster_first_global=2, cluster_narrow_first=1), dict(cluster_first_global=2, cluster_lds_budget=40 * 1024),
                                  dict(index_priv_table=1), dict(index_priv_table=3), dict(index_priv_table=1, index_no_pack=1), dict(index_priv_table=2)])
def test_alternative_layouts_give_the_same_bytes(workdir, opts):
    """The forms the default build does not take on a small set: index build with separate key / block arrays (default: one packed
    word per entry), first[] of the cluster kernel at 2 bytes per entry everywhere / at 4 bytes even where that costs list-loop
    waves (default: 4 bytes where free), in the dense and in the ranked placement; the entry look-ups of the index build through the library's
    own one-read table (default only at -B 29 / 30), also undersized so that half the hashes take its fall-back."""
    orc.gen_fqb(workdir.file("x.fqb"), 40000, 120, 250000, 0.003, 77, 4.0, 150, 6000)
    hf = _against_oracle(workdir, "x.fqb", ["-ct", 3, "--readFQB", "x.fqb", "--hashDepthRange", 4, 40, "--cluster", 1, 0], **opts)
    assert hf.blocks["nSubCluster"].sum() > 0


@pytest.mark.parametrize("knob,form,B,k", [(1, 2, 20, 21), (3, 3, 20, 21), (2, 1, 20, 21), (1, 2, 22, 25), (1, 2, 20, 17)],
                         ids=["probed", "probed_table_fails", "never_probed", "probed_k25", "probed_k17"])
def test_index_table_in_the_probed_format(workdir, knob, form, B, k):
    """Round 6: where index | hash / w does not fit a 64-bit entry (-B 29 / 30 at k = 21) the index build's wide table holds index | hash >> B | probe number in the
    reference's own probe geometry (stage_b.hip probe_insertP_kernel) — hashIndex[] is its index column, the entry look-ups read it, and the two tables round 5 built
    there are gone. Forced here at small -B (knob 1) and for other k; knob 3 gives the probe number one bit, so the table FAILS and the round-5 pair is built (the
    fall-back); knob 2 = never. h10x_counters.index_table_form says what was built; the whole .hash (hashIndex[] byte for byte) equals the oracle's every time."""
    import hash10x_amd
    recs = orc.gen_fqb(workdir.file("x.fqb"), 40000, 120, 250000, 0.003, 77, 4.0, 150, 6000)
    o = orc.Oracle(k, 31, 17, B)
    o.read_fqb(recs); o.depth_range(4, 40); o.cluster(1, 0, 3)
    o.write_hash(workdir.file("orc.hash"))
    h = hash10x_amd.Hash10x(k=k, w=31, r=17, B=B)
    h.set_option("index_probed_table", knob)
    h.read_fqb(recs)
    assert h.counters()["index_table_form"] == form
    h.depth_range(4, 40); h.cluster(1, 0, 3)
    h.write_hash(workdir.file("hip.hash"))
    h.close()
    got, exp = open(workdir.file("hip.hash"), "rb").read(), open(workdir.file("orc.hash"), "rb").read()
    assert got == exp, orc.describe_diff(got, exp)


def test_big_barcode_uses_medium_lds_class_and_wide_lists(workdir):
    """~1200 pairs per barcode (128 KB LDS class) and depths > 64/128 (multi-chunk register lists)."""
    orc.gen_fqb(workdir.file("x.fqb"), 60000, 50, 20000, 0.002, 77, 2.0, 150, 4000)
    hf = _against_oracle(workdir, "x.fqb", ["-ct", 5, "--readFQB", "x.fqb", "--hashDepthRange", 10, 400, "--cluster", 1, 0])
    assert hf.blocks["nSubCluster"].sum() > 0


def test_empty_and_single_barcode_inputs(workdir):
    np.zeros(0, np.uint32).tofile(workdir.file("empty.fqb"))
    _against_oracle(workdir, "empty.fqb", ["--readFQB", "empty.fqb"])
    recs = np.fromfile(workdir.need("tiny.fqb"), dtype=np.uint32).reshape(-1, 30)
    recs[:3].tofile(workdir.file("one.fqb"))            # a single barcode: it is the trailing one => nothing hashed
    _against_oracle(workdir, "one.fqb", ["--readFQB", "one.fqb"])


def test_die_conditions_match_reference_text(workdir):
    import hash10x_amd
    recs = np.fromfile(workdir.need("tiny.fqb"), dtype=np.uint32)
    with pytest.raises(hash10x_amd.Hash10xError, match="out of range 20-30"):
        hash10x_amd.Hash10x(B=19).read_fqb(recs)
    with pytest.raises(hash10x_amd.Hash10xError, match="chunkSize too small"):
        hash10x_amd.Hash10x(B=20).read_fqb(recs, 0, 3)
    h = hash10x_amd.Hash10x(B=20)
    h.read_fqb(recs)
    with pytest.raises(hash10x_amd.Hash10xError, match="you must set hashDepthRange before cluster"):
        h.cluster(1, 0, 5)
    orc.gen_fqb(workdir.file("x.fqb"), 200000, 100, 3000000, 0.02, 5, 10.0, 150, 50000)
    big = np.fromfile(workdir.file("x.fqb"), dtype=np.uint32)
    with pytest.raises(hash10x_amd.Hash10xError, match="hashTableSize is too small"):
        hash10x_amd.Hash10x(B=20).read_fqb(big)       # > 2^18 - 2 distinct hashes (hash10x.c:149)


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not present")
def test_sharded_table_too_small_dies_like_the_reference(workdir):
    """BASELINE configs[3] as written (1.2 B pairs at -B 30) holds more distinct hashes than 2^28 - 2: the reference dies with
    "hashTableSize is too small" (hash10x.c:149). A proportional small set — 200 k pairs, 2 % errors, -B 20: 2^18 - 2 hashes allowed —
    must die the same way when the barcodes are sharded: the distinct hashes are only counted after the owners' exchange
    (shard.hip), on every rank; the C program prints the reference's message and leaves with its exit code under --gpus 2 / 3."""
    import subprocess
    import hash10x_amd
    recs = orc.gen_fqb(workdir.file("x.fqb"), 200000, 100, 3000000, 0.02, 5, 10.0, 150, 50000)
    r = orc.run_ref(["-B", 20, "--readFQB", "x.fqb"], workdir.path)
    assert r.returncode != 0 and b"FATAL ERROR: hashTableSize is too small" in r.stderr
    for nranks in (2, 3):
        with pytest.raises(hash10x_amd.Hash10xError, match="hashTableSize is too small"):
            _run_sharded(recs, nranks, 20, 3, 30, 3, workdir.file("never.hash"))
        g = subprocess.run([os.path.join(orc.REPO, "bin", "hash10x-amd"), "--gpus", str(nranks), "-B", "20", "--readFQB", "x.fqb"], cwd=workdir.path,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, H10X_COMM="local"))
        assert g.returncode == (r.returncode & 0xFF) == 255 and b"FATAL ERROR: hashTableSize is too small" in g.stderr, g.stderr.decode()
    assert not os.path.exists(workdir.file("never.hash"))


def test_context_reuse_is_deterministic(workdir):
    """The C session keeps one device context (stream + recycled device blocks) across --readFQB calls with
    unchanged parameters; every pass must give the same bytes as the first and as the oracle. (A stream-ordered
    hipMallocAsync pool failed exactly this on ROCm 7.2 at this size.)"""
    import hash10x_amd
    recs = orc.gen_fqb(workdir.file("x.fqb"), 300000, 1500, 2000000, 0.004, 9, 8.0, 150, 30000)
    o = orc.Oracle(B=22)
    o.read_fqb(recs)
    o.depth_range(6, 60)
    o.cluster(1, 0, 3)
    o.write_hash(workdir.file("orc.hash"))
    exp = open(workdir.file("orc.hash"), "rb").read()
    h = hash10x_amd.Hash10x(B=22)
    fresh = []
    for it in range(4):
        h.read_fqb(recs)
        h.depth_range(6, 60)
        h.cluster(1, 0, 3)
        h.write_hash(workdir.file("hip.hash"))
        got = open(workdir.file("hip.hash"), "rb").read()
        assert got == exp, "pass %d: %s" % (it, orc.describe_diff(got, exp))
        fresh.append(hash10x_amd.alloc_stats()[0])
    assert orc.HashFile(exp).blocks["nSubCluster"].sum() > 0
    # a repeated pass finds every device block it needs among those the pass before gave back: nothing comes from hipMalloc any more (the tables of the
    # state being replaced are released before their successors are built; at 200 M read pairs a pass that allocated lost 50-150 ms to it)
    assert fresh[1] == fresh[2] == fresh[3], fresh


def test_warm_loads_the_device_code_from_another_thread():
    """h10x_warm (include/h10x.h): called from a thread of its own while the main thread works on a context, as the C host does beside the file read"""
    import threading
    import hash10x_amd
    rc = []
    th = threading.Thread(target=lambda: rc.append(hash10x_amd.warm(0)))
    th.start()
    h = hash10x_amd.Hash10x(B=20)
    h.read_fqb(np.zeros(0, dtype=np.uint32))
    th.join()
    assert rc == [0] and hash10x_amd.warm(0) == 0
    h.close()


def _run_sharded(recs, nranks, B, lo, hi, ct, out_path, gather=False, opts=None, after=None):
    """N ranks as N threads on one GPU (in-process communicator): the whole sharded pipeline. Every rank writes its slices of
    the .hash file (gather=True: rank 0 collects the state first, the round-1 way). after(h, rank): more collective commands
    before the file is written."""
    import threading
    import hash10x_amd
    cut = hash10x_amd.partition(recs, nranks)
    comms = hash10x_amd.Comm.local(nranks)
    flat = np.ascontiguousarray(recs, dtype=np.uint32).reshape(-1)
    errs = [None] * nranks

    def work(r):
        try:
            h = hash10x_amd.Hash10x(B=B)
            for n, v in (opts or {}).items():
                h.set_option(n, v)
            h.shard_read_fqb(comms[r], flat[30 * cut[r]: 30 * cut[r + 1]])
            h.depth_range(lo, hi)
            h.cluster(1, 0, ct)
            if after:
                after(h, r)
            if gather:
                h.shard_gather()
                if r == 0:
                    h.write_hash(out_path)
            else:
                h.write_hash(out_path)
            h.close()
        except Exception as e:              # noqa: BLE001
            errs[r] = e
    th = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    for c in comms:
        c.destroy()
    for e in errs:
        if e is not None:
            raise e


@pytest.mark.parametrize("nranks", [1, 2, 3, 4, 8])
def test_sharded_equals_single_gpu_and_oracle(workdir, nranks):
    """SURVEY §8e invariant: N-rank canonical .hash == 1-GPU == oracle (hash-owner all-to-all, index numbering by
    allgathered first-barcode counts, allgather of in-range barcode lists, gather to rank 0)."""
    recs = orc.gen_fqb(workdir.file("x.fqb"), 40000, 180, 300000, 0.003, 61, 4.0, 150, 6000)
    o = orc.Oracle(B=20)
    o.read_fqb(recs)
    o.depth_range(4, 30)
    o.cluster(1, 0, 3)
    o.write_hash(workdir.file("orc.hash"))
    exp = open(workdir.file("orc.hash"), "rb").read()
    _run_sharded(recs, nranks, 20, 4, 30, 3, workdir.file("hip.hash"))
    got = open(workdir.file("hip.hash"), "rb").read()
    assert got == exp, orc.describe_diff(got, exp)
    assert orc.HashFile(exp).blocks["nSubCluster"].sum() > 0
    if nranks in (2, 3):                                     # the gather to rank 0 gives the same file
        _run_sharded(recs, nranks, 20, 4, 30, 3, workdir.file("hip2.hash"), gather=True)
        assert open(workdir.file("hip2.hash"), "rb").read() == exp


@pytest.mark.parametrize("nranks,delta", [(1, 1), (3, 0), (3, 1), (4, -1)])
def test_sharded_lists_travel_delta_coded(workdir, nranks, delta):
    """The in-range barcode lists can go between ranks as 16-bit steps (a list with a step beyond 16 bits goes as it is): forced on one
    rank, off and on with three; the words received must be about half the plain ones where it is on. By default (four ranks, -1) they travel
    coded only where bytes are dear — the host-staged TCP backend — and plain between the ranks of one process as over xGMI (round 5: at 8 ranks
    the coding costs every rank more compute than it saves link time, bench.py --virtual-ranks)."""
    import hash10x_amd
    orc.gen_fqb(workdir.file("x.fqb"), 60000, 300, 400000, 0.003, 43, 4.0, 150, 6000)
    run_commands(lambda k, w, r, B: orc.Oracle(k, w, r, B), ["-B", 22, "-ct", 3, "--readFQB", "x.fqb", "--hashDepthRange", 4, 40, "--cluster", 1, 0, "--writeHash", "orc.hash"], workdir.path)
    exp = open(workdir.file("orc.hash"), "rb").read()
    recs = np.fromfile(workdir.file("x.fqb"), dtype=np.uint32)
    words = {}
    _run_sharded(recs, nranks, 22, 4, 40, 3, workdir.file("sh.hash"), opts=dict(shard_delta_lists=delta),
                 after=lambda h, r: words.__setitem__(r, h.counters()["list_words"]))
    got = open(workdir.file("sh.hash"), "rb").read()
    assert got == exp, orc.describe_diff(got, exp)
    plain, coded = words[0]
    if delta <= 0:
        assert plain == 0 and coded == 0
    else:
        assert plain > 0 and 0.5 * plain <= coded <= 0.62 * plain, (plain, coded)


@pytest.mark.parametrize("shift,fake", [(0, 1 << 32), (3, 0), (2, (7 << 32) + 4), (1, 1 << 35)])
def test_sharded_list_offsets_beyond_32_bits(workdir, shift, fake):
    """BASELINE configs[3]/[4] hold more than 2^32 in-range list entries: cluster_kernel then keeps a list's offset as a 32-bit
    count of 2^shift-entry units (lists padded to that alignment by the exchange). Exercised on a small set by forcing the
    alignment and by letting the offsets start `fake` entries in front of the array (64-bit address arithmetic in the kernel)."""
    recs = orc.gen_fqb(workdir.file("x.fqb"), 40000, 180, 300000, 0.003, 61, 4.0, 150, 6000)
    o = orc.Oracle(B=20)
    o.read_fqb(recs)
    o.depth_range(4, 30)
    o.cluster(1, 0, 3)
    o.write_hash(workdir.file("orc.hash"))
    exp = open(workdir.file("orc.hash"), "rb").read()
    _run_sharded(recs, 3, 20, 4, 30, 3, workdir.file("hip.hash"), opts={"shard_row_shift": shift, "shard_rows_fake_base": fake})
    got = open(workdir.file("hip.hash"), "rb").read()
    assert got == exp, orc.describe_diff(got, exp)


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4], ids=["lookup", "scatter", "lookup_then_scatter", "lookup_table_fails", "lookup_table_does_not_fit"])
def test_sharded_owner_replies(workdir, mode):
    """How a hash owner tells every entry its index (shard.hip step 5): by look-up in a table of its distinct hashes, arrival order (the default where entries travel packed),
    by scattering from its sorted order (the sort carries arrival positions: round 4's form, still what unpacked entries take), and the look-up's fall-backs: a look-up that
    succeeds answered by scatter all the same (2), a table whose displacement limit is 1 so that it really FAILS (3: the recovery path of the default mechanism), a table that
    does not fit the free memory (4). h10x_counters.shard_reply_path says which path answered (mode 2 used to be mode 1 in disguise). 3 ranks and 1 rank, each
    byte-equal to the oracle."""
    recs = orc.gen_fqb(workdir.file("x.fqb"), 40000, 180, 300000, 0.003, 61, 4.0, 150, 6000)
    o = orc.Oracle(B=20)
    o.read_fqb(recs); o.depth_range(4, 30); o.cluster(1, 0, 3)
    o.write_hash(workdir.file("orc.hash"))
    exp = open(workdir.file("orc.hash"), "rb").read()
    want = {0: 1, 1: 2, 2: 3, 3: 3, 4: 3}[mode]
    for nranks in (3, 1):
        path = [None] * nranks
        _run_sharded(recs, nranks, 20, 4, 30, 3, workdir.file("hip.hash"), opts={"shard_reply_sort": mode},
                     after=lambda h, r: path.__setitem__(r, h.counters()["shard_reply_path"]))
        got = open(workdir.file("hip.hash"), "rb").read()
        assert got == exp, orc.describe_diff(got, exp)
        assert path == [want] * nranks, path


def test_hash_owner_shares_are_balanced(workdir):
    """Hash owners hold equal SHARES of the entries, not equal value ranges: a canonical hash is min(hashF, hashR) (seqhash.c:67-68), density 2 (1 - x), so equal ranges gave owner 0
    of 8 1 - (7/8)^2 = 23.4 %. Cut at the quantiles (shard.hip ownerCuts) the busiest owner of 8 receives at most 14 % of the index replies; the old cut (knob
    shard_owner_cut = 1) shows the 23 %; both byte-equal to the oracle."""
    recs = orc.gen_fqb(workdir.file("x.fqb"), 60000, 240, 400000, 0.003, 67, 4.0, 150, 6000)
    o = orc.Oracle(B=21)
    o.read_fqb(recs); o.depth_range(4, 30); o.cluster(1, 0, 3)
    o.write_hash(workdir.file("orc.hash"))
    exp = open(workdir.file("orc.hash"), "rb").read()
    share = {}
    for cut in (0, 1):
        out = [None] * 8
        _run_sharded(recs, 8, 21, 4, 30, 3, workdir.file("hip.hash"), opts={"shard_owner_cut": cut},
                     after=lambda h, r: out.__setitem__(r, h.exchanges()["indices_back (all-to-all)"]["bytes_out"]))
        assert open(workdir.file("hip.hash"), "rb").read() == exp
        share[cut] = max(out) / sum(out)
    assert share[0] <= 0.14, share
    assert 0.21 <= share[1] <= 0.26, share


@pytest.mark.parametrize("overlap", [1, 0], ids=["exchange_stream", "main_stream"])
def test_sharded_exchanges_beside_compute(workdir, overlap):
    """Round 6: the exchanges whose result a LATER stage needs run on the context's exchange stream beside the main stream's kernels — the in-range barcode lists beside the
    good lists (wanted by --cluster), hashDepth[] of the other owners beside the ClusterHash records (wanted by --hashDepthRange); knob shard_overlap 0 = everything on the
    main stream as in round 5. Both byte-equal to the oracle, on 3 ranks and on 1, and h10x_exchange_beside says which way the exchanges went."""
    recs = orc.gen_fqb(workdir.file("x.fqb"), 40000, 180, 300000, 0.003, 61, 4.0, 150, 6000)
    o = orc.Oracle(B=20)
    o.read_fqb(recs); o.depth_range(4, 30); o.cluster(1, 0, 3)
    o.write_hash(workdir.file("orc.hash"))
    exp = open(workdir.file("orc.hash"), "rb").read()
    for nranks in (3, 1):
        beside = [None] * nranks
        _run_sharded(recs, nranks, 20, 4, 30, 3, workdir.file("hip.hash"), opts={"shard_overlap": overlap},
                     after=lambda h, r: beside.__setitem__(r, {k.split(" ")[0]: v["beside"] for k, v in h.exchanges().items()}))
        got = open(workdir.file("hip.hash"), "rb").read()
        assert got == exp, orc.describe_diff(got, exp)
        for b in beside:
            assert b["list_data"] == ("good_hashes" if overlap else None) and b["index_depth"] == ("clushash_build" if overlap else None), b
            assert b["entries_to_hash_owners"] is None and b["indices_back"] is None


def test_sharded_blocks_too_large_for_a_workgroup_sort(workdir):
    """Sharded --readFQB where a block holds more entries than a workgroup sorts in LDS (BLOCK_SORT_MAX = 8192): the ClusterHash records then take the device-wide path, fed by
    reply_key_kernel (the owners' replies gathered into keys) instead of the block kernel's own gather. 8 barcodes of 10 000 read pairs over 40 molecules each, on 3 ranks and on 1, byte-equal to the oracle."""
    recs = orc.gen_fqb(workdir.file("x.fqb"), 80000, 8, 2000000, 0.003, 71, 40.0, 150, 10000)
    o = orc.Oracle(B=21)
    o.read_fqb(recs); o.depth_range(2, 9); o.cluster(1, 0, 1)
    o.write_hash(workdir.file("orc.hash"))
    exp = open(workdir.file("orc.hash"), "rb").read()
    assert orc.HashFile(exp).blocks["nHash"].max() > 8192 and orc.HashFile(exp).blocks["nSubCluster"].sum() > 0
    for nranks in (3, 1):
        _run_sharded(recs, nranks, 21, 2, 9, 1, workdir.file("hip.hash"))
        got = open(workdir.file("hip.hash"), "rb").read()
        assert got == exp, orc.describe_diff(got, exp)


def test_gather_then_continue_on_one_gpu(workdir):
    """h10x_shard_gather leaves rank 0 a complete single-GPU context: its barcode lists are rebuilt, so a new depth range and
    another clustering on rank 0 alone give what one GPU gives from the start (the lists used to be stale)."""
    import threading
    import hash10x_amd
    recs = orc.gen_fqb(workdir.file("x.fqb"), 40000, 180, 300000, 0.003, 61, 4.0, 150, 6000)
    o = orc.Oracle(B=20)
    o.read_fqb(recs); o.depth_range(4, 30); o.cluster(1, 0, 3); o.depth_range(2, 40); o.cluster(1, 0, 2)
    o.write_hash(workdir.file("orc.hash"))
    exp = open(workdir.file("orc.hash"), "rb").read()
    nranks = 3
    cut = hash10x_amd.partition(recs, nranks); comms = hash10x_amd.Comm.local(nranks)
    flat = np.ascontiguousarray(recs, dtype=np.uint32).reshape(-1); errs = [None] * nranks

    def work(r):
        try:
            h = hash10x_amd.Hash10x(B=20)
            h.shard_read_fqb(comms[r], flat[30 * cut[r]: 30 * cut[r + 1]])
            h.depth_range(4, 30); h.cluster(1, 0, 3)
            h.shard_gather()
            if r == 0:
                h.depth_range(2, 40); h.cluster(1, 0, 2)
                h.write_hash(workdir.file("hip.hash"))
            h.close()
        except Exception as e:              # noqa: BLE001
            errs[r] = e
    th = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in th]; [t.join(600) for t in th]
    [c.destroy() for c in comms]
    for e in errs:
        if e is not None:
            raise e
    got = open(workdir.file("hip.hash"), "rb").read()
    assert got == exp, orc.describe_diff(got, exp)


@pytest.mark.parametrize("nranks", [2, 4])
def test_sharded_split_recluster_and_second_split(workdir, nranks):
    """--clusterSplit on shards: the new blocks are numbered behind the blocks of ALL ranks in the order of their parents, the
    hash owners rebuild their barcode lists, and the split blocks cluster (and split) again like the reference's."""
    recs = orc.gen_fqb(workdir.file("x.fqb"), 40000, 180, 300000, 0.003, 61, 4.0, 150, 6000)
    o = orc.Oracle(B=20)
    o.read_fqb(recs); o.depth_range(4, 30); o.cluster(1, 0, 3); o.cluster_split()
    o.depth_range(3, 30); o.cluster(1, 0, 2); o.cluster_split(); o.depth_range(3, 30); o.cluster(1, 0, 2)
    o.write_hash(workdir.file("orc.hash"))
    exp = open(workdir.file("orc.hash"), "rb").read()

    def after(h, r):
        h.cluster_split(); h.depth_range(3, 30); h.cluster(1, 0, 2); h.cluster_split(); h.depth_range(3, 30); h.cluster(1, 0, 2)
    _run_sharded(recs, nranks, 20, 4, 30, 3, workdir.file("hip.hash"), after=after)
    got = open(workdir.file("hip.hash"), "rb").read()
    assert got == exp, orc.describe_diff(got, exp)
    hf = orc.HashFile(exp)
    assert hf.blocks_max > 181 and hf.blocks["clusterParent"].max() > 181          # second-generation blocks exist


@pytest.mark.parametrize("nranks", [2, 4, 8])
def test_sharded_with_fewer_barcodes_than_ranks(workdir, nranks):
    """Trailing (and middle) shards without a record: the file's unhashed last barcode is then the last barcode
    of the last shard that holds any, and an empty shard contributes no block."""
    recs = orc.gen_fqb(workdir.file("x.fqb"), 6000, 3, 60000, 0.003, 77, 3.0, 150, 4000)
    o = orc.Oracle(B=20)
    o.read_fqb(recs)
    o.depth_range(2, 30)
    o.cluster(1, 0, 2)
    o.write_hash(workdir.file("orc.hash"))
    exp = open(workdir.file("orc.hash"), "rb").read()
    _run_sharded(recs, nranks, 20, 2, 30, 2, workdir.file("hip.hash"))
    got = open(workdir.file("hip.hash"), "rb").read()
    assert got == exp, orc.describe_diff(got, exp)
    # no record at all: rank 0 keeps the reference's empty block 1 (hash10x.c:200-201)
    empty = recs[:0]
    o = orc.Oracle(B=20)
    o.read_fqb(empty)
    o.depth_range(2, 30)
    o.cluster(1, 0, 2)
    o.write_hash(workdir.file("orc0.hash"))
    _run_sharded(empty, nranks, 20, 2, 30, 2, workdir.file("hip0.hash"))
    assert open(workdir.file("hip0.hash"), "rb").read() == open(workdir.file("orc0.hash"), "rb").read()


def test_random_parameter_soak():
    """24 seeded random cases of tests/soak.py (parameters, knobs, second range, clusterSplit, sharded with 1-4 ranks)."""
    import soak
    bad = soak.run(24, 314)
    assert not bad, bad


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not present")
def test_random_command_line_soak():
    """12 seeded random command lines of tests/soak_cli.py: bin/hash10x-amd against the reference binary, .hash bytes and
    every report line (--hashStats, --codeStats, --cribBuild, --clusterReport, --clusterSplit, --cribSummary)."""
    import soak_cli
    bad = soak_cli.run(12, 271)
    assert not bad, bad


def test_sharded_golden_small(workdir):
    recs = np.frombuffer(orc.read_maybe_gz(os.path.join(orc.GOLDEN, "small.fqb.gz")), dtype=np.uint32)
    _run_sharded(recs, 4, 20, 3, 14, 2, workdir.file("hip.hash"))
    got = open(workdir.file("hip.hash"), "rb").read()
    exp = orc.read_maybe_gz(os.path.join(orc.GOLDEN, "small.e2e.hash.gz"))
    assert got == exp, orc.describe_diff(got, exp)


def test_rccl_communicator_single_rank(workdir):
    """The RCCL backend with one rank (all a 1-GPU box can run): init, the self-copy path of alltoallv, teardown."""
    import hash10x_amd
    recs = np.frombuffer(orc.read_maybe_gz(os.path.join(orc.GOLDEN, "small.fqb.gz")), dtype=np.uint32)
    comm = hash10x_amd.Comm.rccl(0, 1, hash10x_amd.Comm.unique_id(), 0)
    h = hash10x_amd.Hash10x(B=20)
    h.shard_read_fqb(comm, recs)
    h.depth_range(3, 14)
    h.cluster(1, 0, 2)
    assert h.shard_allreduce_max(1.5) == 1.5
    h.shard_gather()
    h.write_hash(workdir.file("hip.hash"))
    h.close()
    comm.destroy()
    got = open(workdir.file("hip.hash"), "rb").read()
    exp = orc.read_maybe_gz(os.path.join(orc.GOLDEN, "small.e2e.hash.gz"))
    assert got == exp, orc.describe_diff(got, exp)


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not present")
def test_cli_matches_reference_commands_and_reports(workdir):
    """bin/hash10x-amd (C host program) against the reference binary on the same command line: identical .hash bytes and
    identical --hashStats / --codeStats report lines."""
    import subprocess
    orc.gen_fqb(workdir.file("x.fqb"), 30000, 150, 300000, 0.003, 41, 4.0, 150, 6000)
    args = ["-B", "20", "-ct", "3", "--readFQB", "x.fqb", "--hashStats", "--hashDepthRange", "4", "30", "--cluster", "1", "0", "--codeStats",
            "--writeHash", "OUT"]
    r = orc.run_ref([a if a != "OUT" else "ref.hash" for a in args], workdir.path)
    assert r.returncode == 0, r.stderr.decode()
    g = subprocess.run([os.path.join(orc.REPO, "bin", "hash10x-amd")] + [a if a != "OUT" else "hip.hash" for a in args], cwd=workdir.path,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert g.returncode == 0, g.stderr.decode()
    exp = orc.canonical_hash_bytes(open(workdir.file("ref.hash"), "rb").read())
    got = open(workdir.file("hip.hash"), "rb").read()
    assert got == exp, orc.describe_diff(got, exp)

    def report(txt):
        return [ln for ln in txt.decode().splitlines() if ln.startswith(("HASH_COUNT_", "CODE_SIZE_", "CODE_CLUSTER_"))]
    assert report(g.stdout) == report(r.stdout) and len(report(g.stdout)) > 50
    # die() conditions of the command line
    bad = subprocess.run([os.path.join(orc.REPO, "bin", "hash10x-amd"), "-B", "19", "--readFQB", "x.fqb"], cwd=workdir.path, stderr=subprocess.PIPE, stdout=subprocess.PIPE)
    assert bad.returncode == 255 and b"FATAL ERROR: hashTableBits 19 out of range 20-30" in bad.stderr


def _hash_v2_to_v1(data):
    """A version-1 .hash from a version-2 one: the same bytes, but hashValue[] is stored as an Array (ArrayStruct header, dim = max = hashNumber,
    size 8: hash10x.c:286-292, array.c:213-218) instead of a count + the values."""
    import struct
    B = struct.unpack_from("<i", data, 12)[0]
    at = 16 + (4 << B)
    n = struct.unpack_from("<I", data, at)[0]
    head = struct.pack("<iiQiiii", 8918274, 0, 0, n, 8, n, 0)                 # magic, pad, base pointer, dim, size, max, pad (array.h:41-50)
    return data[:4] + struct.pack("<I", 1) + data[8:at] + head + data[at + 4:]


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not present")
def test_cli_reads_a_version_1_hash_file(workdir):
    """readHashFile accepts version 1 files, which store hashValue[] as an Array (hash10x.c:286-292): a v1 file made from the small golden .hash
    goes through --readHash --hashDepthRange --cluster --writeHash of bin/hash10x-amd and of the reference binary with the same bytes out (always v2)."""
    import subprocess
    v2 = orc.read_maybe_gz(os.path.join(orc.GOLDEN, "small.hash.gz"))
    with open(workdir.file("v1.hash"), "wb") as f:
        f.write(_hash_v2_to_v1(v2))
    args = ["-B", "20", "-ct", "2", "--readHash", "v1.hash", "--hashDepthRange", "3", "14", "--cluster", "1", "0", "--writeHash", "OUT"]
    r = orc.run_ref([a if a != "OUT" else "ref.hash" for a in args], workdir.path)
    assert r.returncode == 0, r.stderr.decode()
    g = subprocess.run([os.path.join(orc.REPO, "bin", "hash10x-amd")] + [a if a != "OUT" else "hip.hash" for a in args], cwd=workdir.path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert g.returncode == 0, g.stderr.decode()
    exp = orc.canonical_hash_bytes(open(workdir.file("ref.hash"), "rb").read())
    got = open(workdir.file("hip.hash"), "rb").read()
    assert got == exp, orc.describe_diff(got, exp)
    assert got[4:8] == b"\x02\x00\x00\x00"
    # onto several GPUs a v1 file is refused (the sharded reader preads the v2 layout)
    bad = subprocess.run([os.path.join(orc.REPO, "bin", "hash10x-amd"), "--gpus", "2", "-B", "20", "--readHash", "v1.hash"], cwd=workdir.path, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, env=dict(os.environ, H10X_COMM="local"))
    assert bad.returncode == 255 and b"only version 2 files" in bad.stderr


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not present")
@pytest.mark.parametrize("gpus", [1, 3])
def test_cli_verbose_cluster_lines_match_reference(workdir, gpus):
    """--verbose: the per-barcode lines of codeClusterFind / codeClusterReadMerge (hash10x.c:827-834, 867: reads, hashes, good hashes, how many of them
    were labelled, clusters before and after the read merge) equal the reference binary's, and so does its note for a barcode given up at the 256th
    cluster (hash10x.c:813, stderr, verbose or not) — on one GPU and on three shards."""
    import subprocess
    orc.gen_fqb(workdir.file("x.fqb"), 30000, 150, 300000, 0.003, 41, 4.0, 150, 6000)
    pre = ["--gpus", str(gpus)] if gpus > 1 else []
    env = dict(os.environ, H10X_COMM="local")

    def lines(txt):
        return [ln for ln in txt.decode().splitlines() if ln.startswith(("  code ", " then "))]
    for args in (["-B", "20", "-ct", "3", "--verbose", "--readFQB", "x.fqb", "--hashDepthRange", "4", "30", "--cluster", "1", "0", "--hashDepthRange", "30", "31", "--cluster", "1", "60"],
                 ["-B", "20", "--verbose", "--readHash", "abort255.hash", "--hashDepthRange", "2", "100", "--cluster", "1", "2"]):
        if "abort255.hash" in args:
            with open(workdir.file("abort255.hash"), "wb") as f:
                f.write(orc.read_maybe_gz(os.path.join(orc.GOLDEN, "abort255.in.hash.gz")))
        r = orc.run_ref(args, workdir.path)
        assert r.returncode == 0, r.stderr.decode()
        g = subprocess.run([os.path.join(orc.REPO, "bin", "hash10x-amd")] + pre + args, cwd=workdir.path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert g.returncode == 0, g.stderr.decode()
        assert lines(g.stdout) == lines(r.stdout) and len(lines(r.stdout)) > 0
        note = [ln for ln in r.stderr.decode().splitlines() if "too many clusters" in ln]
        assert [ln for ln in g.stderr.decode().splitlines() if "too many clusters" in ln] == note
        if "abort255.hash" in args:
            assert len(note) == 1


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not present")
def test_cli_crib_reports_match_reference(workdir):
    """--cribBuild (both haplotypes hashed and looked up on the device), --clusterReport, --clusterSplit, --cribSummary:
    every report line of bin/hash10x-amd equals the reference binary's on the same command line (config 5's accuracy
    check), including the CRIB_TABLE, the per-cluster chr / pos spans and the OTHER hash lists."""
    import subprocess
    orc.gen_fqb(workdir.file("x.fqb"), 30000, 150, 300000, 0.003, 41, 4.0, 150, 6000, fa=workdir.file("x"))
    # cut both haplotypes into 30 kb "chromosomes" so that clusters straddle sequence boundaries (OTHER lists)
    for hap in ("A", "B"):
        lines = open(workdir.file("x.%s.fa" % hap)).read().splitlines()[1:]
        with open(workdir.file("x.%s.fa" % hap), "w") as f:
            for i in range(0, len(lines), 500):
                f.write(">c%d\n%s\n" % (i // 500 + 1, "\n".join(lines[i:i + 500])))
    # more sequences in haplotype B: lower case, Ns, a record shorter than k, a repeat of part of A ('mul' class)
    with open(workdir.file("x.B.fa"), "a") as f:
        f.write(">extra some description\n" + "acgtnACGTN" * 30 + "\n>tiny\nACGT\n>dup\n")
        a = open(workdir.file("x.A.fa")).read().splitlines()[1:40]
        f.write("\n".join(a) + "\n")
    args = ["-B", "20", "-ct", "3", "--readFQB", "x.fqb", "--hashDepthRange", "4", "30", "--cluster", "1", "0", "--tables",
            "--cribBuild", "x.A.fa", "x.B.fa", "--clusterReport", "0", "0", "--clusterReport", "3", "9", "--clusterSplit", "--cribSummary",
            "--hashDepthRange", "4", "30", "--clusterReport", "140", "175", "--cluster", "1", "0", "--codeStats", "--hashStats", "--clusterReport", "150", "400",
            "--writeHash", "ref.hash"]
    r = orc.run_ref(args, workdir.path)
    assert r.returncode == 0, r.stderr.decode()
    def report(txt):
        keep = ("  crib matches", "    hom", "    het", "    mul", "    err", "CRIB_TABLE", "  CLUSTER_SUMMARY", "    CODE_CLUSTER", "  made ", "  filled hash table",
                "  MIN_POINT_DENSITY", "CODE_SIZE_", "CODE_CLUSTER_", "HASH_COUNT_")
        return [ln for ln in txt.decode().splitlines() if ln.startswith(keep) or " base codes " in ln or " cluster codes " in ln or " in crib genome" in ln]
    exp = report(r.stdout)
    exp_hash = orc.canonical_hash_bytes(open(workdir.file("ref.hash"), "rb").read())
    for gpus in (1, 4, 3):                                   # --gpus N: the same commands on N shards (ranks share this box's GPU), no gather
        g = subprocess.run([os.path.join(orc.REPO, "bin", "hash10x-amd")] + (["--gpus", str(gpus)] if gpus > 1 else []) +
                           [a if a != "ref.hash" else "hip%d.hash" % gpus for a in args], cwd=workdir.path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert g.returncode == 0, g.stderr.decode()
        got = report(g.stdout)
        for i, (a, b) in enumerate(zip(got, exp)):
            assert a == b, "--gpus %d, line %d differs:\n  hip: %s\n  ref: %s" % (gpus, i, a, b)
        assert len(got) == len(exp) and len(got) > 400
        assert any("OTHER" in ln for ln in got) and any(" mul," in ln for ln in got)
        got_hash = open(workdir.file("hip%d.hash" % gpus), "rb").read()
        assert got_hash == exp_hash, "--gpus %d: %s" % (gpus, orc.describe_diff(got_hash, exp_hash))


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not present")
def test_cli_barcode_with_more_than_65535_pairs(workdir):
    """ClusterHash.read is a U16 (hash10x.c:37,180), so a barcode with more than 65535 read pairs has its late
    reads stored modulo 2^16 — stage A sends such a block to the global-memory mosh path, the read merge and the reports size
    their per-read tables by nRead. Input (orc.build_pairs65k): BIG = 74 k pairs and > 65535 hashes (ignored by clustering,
    hash10x.c:748-753), BIG2 = 75 k pairs of which the first 66 000 repeat 400 pairs (clustered; its late reads share numbers
    with early ones), ordinary barcodes around them. The README's recipe behind it (range, cluster, report, split, cluster
    again): .hash bytes and every report line of bin/hash10x-amd equal the reference binary's, on one GPU and on 3 shards.
    (The same set is pinned by two reference digests in manifest.json: test_hip_matches_reference_digests[pairs65k.*].)"""
    import subprocess
    orc.build_pairs65k(workdir.file("x.fqb"))
    args = ["-B", "22", "-c", "200000", "-ct", "2", "--readFQB", "x.fqb", "--hashDepthRange", "4", "30", "--cluster", "1", "0", "--clusterReport", "0", "0",
            "--codeStats", "--clusterSplit", "--hashDepthRange", "4", "30", "--cluster", "1", "0", "--clusterReport", "0", "0", "--codeStats", "--writeHash", "ref.hash"]
    r = orc.run_ref(args, workdir.path)
    assert r.returncode == 0, r.stderr.decode()

    def report(txt):
        keep = ("  CLUSTER_SUMMARY", "    CODE_CLUSTER", "  MIN_POINT_DENSITY", "CODE_SIZE_", "CODE_CLUSTER_")
        return [ln for ln in txt.decode().splitlines() if ln.startswith(keep)]
    exp = report(r.stdout)
    exp_hash = orc.canonical_hash_bytes(open(workdir.file("ref.hash"), "rb").read())
    hf = orc.HashFile(exp_hash)
    assert len(exp) > 100 and int(hf.blocks["nRead"].max()) > 65535
    # with a chunk size below the longest run the reference dies on this file, and so must we (hash10x.c:206)
    rbad = orc.run_ref(["-B", "22", "-c", "70000", "--readFQB", "x.fqb"], workdir.path)
    assert rbad.returncode != 0 and b"FATAL ERROR: chunkSize too small" in rbad.stderr
    bad = subprocess.run([os.path.join(orc.REPO, "bin", "hash10x-amd"), "-B", "22", "-c", "70000", "--readFQB", "x.fqb"], cwd=workdir.path, stderr=subprocess.PIPE, stdout=subprocess.PIPE)
    assert bad.returncode == 255 and b"FATAL ERROR: chunkSize too small" in bad.stderr
    for gpus in (1, 3):
        g = subprocess.run([os.path.join(orc.REPO, "bin", "hash10x-amd")] + (["--gpus", str(gpus)] if gpus > 1 else []) +
                           [a if a != "ref.hash" else "hip%d.hash" % gpus for a in args], cwd=workdir.path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert g.returncode == 0, g.stderr.decode()
        got = report(g.stdout)
        for i, (x, y) in enumerate(zip(got, exp)):
            assert x == y, "--gpus %d, line %d differs:\n  hip: %s\n  ref: %s" % (gpus, i, x, y)
        assert len(got) == len(exp)
        got_hash = open(workdir.file("hip%d.hash" % gpus), "rb").read()
        assert got_hash == exp_hash, "--gpus %d: %s" % (gpus, orc.describe_diff(got_hash, exp_hash))


def test_sort_fqb_on_device(workdir):
    """--sortFQB (the reference leaves this step to `bsort -k 4 -r 120`): records ordered by their first four bytes,
    stably — equal to numpy's stable sort on the byte-swapped barcode word; the sorted file then reads like any other."""
    import subprocess
    recs = orc.gen_fqb(workdir.file("x.fqb"), 20000, 300, 200000, 0.003, 47, 3.0, 150, 5000)
    rng = np.random.default_rng(3)
    shuffled = recs[rng.permutation(recs.shape[0])]
    shuffled.tofile(workdir.file("shuf.fqb"))
    g = subprocess.run([os.path.join(orc.REPO, "bin", "hash10x-amd"), "--sortFQB", "shuf.fqb", "sorted.fqb", "-B", "20", "--readFQB", "sorted.fqb",
                        "--writeHash", "hip.hash"], cwd=workdir.path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert g.returncode == 0, g.stderr.decode()
    got = np.fromfile(workdir.file("sorted.fqb"), dtype=np.uint32).reshape(-1, 30)
    exp = shuffled[np.argsort(shuffled[:, 0].byteswap(), kind="stable")]
    assert got.shape == exp.shape and np.array_equal(got, exp)
    o = orc.Oracle(B=20); o.read_fqb(exp.reshape(-1)); o.write_hash(workdir.file("orc.hash"))
    a, b = open(workdir.file("hip.hash"), "rb").read(), open(workdir.file("orc.hash"), "rb").read()
    assert a == b, orc.describe_diff(a, b)


def test_full_size_properties_and_round_trip(workdir):
    """BASELINE configs[1] at full size (2.5 M read pairs, 10 k barcodes, -B 24; the reference needs 13 s for it, bench.py
    does that byte comparison): size-independent properties of the state instead — clusHash strictly ascending inside every
    block, hashDepth = barcodes per hash, every stored index present in hashValue[], labels within nSubCluster, a second
    pass bit-identical, --writeHash -> --readHash -> --writeHash a fixed point, and re-clustering the read-back state
    reproduces the clustered file."""
    import sys
    sys.path.insert(0, orc.REPO)
    import bench
    import hash10x_amd
    wl = bench.WORKLOADS["yeast-like-2.5M"]
    recs = bench.generate(wl, 1)
    h = hash10x_amd.Hash10x(B=wl["B"])
    d = hash10x_amd.DeviceRecords(recs)
    h.read_fqb_device(d.ptr, d.n_records)
    h.depth_range(wl["lo"], wl["hi"])
    h.cluster(1, 0, wl["ct"])
    z = h.sizes(); b = h.export_blocks(); ch = h.export_clushash(); depth = h.export_depth()
    assert z["nRecords"] == wl["pairs"] and z["nBlocks"] == wl["barcodes"] + 1 and b["nHash"][-1] == 0        # last barcode never hashed
    off = np.concatenate([[0], np.cumsum(b["nHash"][1:].astype(np.int64))])
    ix = ch["hash"].astype(np.int64)
    starts = np.zeros(ix.size, bool); starts[off[:-1][off[:-1] < ix.size]] = True
    assert np.all((np.diff(ix) > 0) | starts[1:])
    assert ix.min() >= 1 and ix.max() == z["hashNumber"] - 1
    assert np.array_equal(np.bincount(ix, minlength=z["hashNumber"])[1:], depth[1:])
    assert np.all(ch["subCluster"] <= np.repeat(b["nSubCluster"][1:], b["nHash"][1:])) and b["nSubCluster"].max() <= 255
    assert np.all(ch["read"] < np.repeat(b["nRead"][1:], b["nHash"][1:]))
    h.write_hash(workdir.file("a.hash"))
    # second pass on the same context
    h.read_fqb_device(d.ptr, d.n_records); h.depth_range(wl["lo"], wl["hi"]); h.cluster(1, 0, wl["ct"])
    h.write_hash(workdir.file("b.hash"))
    a = open(workdir.file("a.hash"), "rb").read()
    assert a == open(workdir.file("b.hash"), "rb").read()
    h.close()
    # file round trip, then re-cluster from the file
    g = hash10x_amd.Hash10x(B=wl["B"])
    g.read_hash(workdir.file("a.hash")); g.write_hash(workdir.file("c.hash"))
    assert a == open(workdir.file("c.hash"), "rb").read()
    g.depth_range(wl["lo"], wl["hi"]); g.cluster(1, 0, wl["ct"]); g.write_hash(workdir.file("d.hash"))
    assert a == open(workdir.file("d.hash"), "rb").read()
    g.close()


# ---------------------------------------------------------------------------------------- BASELINE configs[2] proportions
_BIG = {}


def _big_set(workdir_factory, case):
    """the seeded input of a big digest case, generated once per test session"""
    key = json.dumps(case["gen"], sort_keys=True)
    if key not in _BIG:
        d = workdir_factory.mktemp("big")
        recs = orc.gen_fqb(str(d / "big.fqb"), **case["gen"])
        os.remove(str(d / "big.fqb"))
        _BIG.clear()                                         # one 2.4 - 6 GB image at a time
        _BIG[key] = recs
    return _BIG[key]


def _roomy_dir(tmp_path_factory, name, need):
    """a directory for multi-GB test files: memory-backed storage where it has `need` bytes and to spare (the files of the full-size cases go through a GPU box's disk at a
    few GB/s otherwise), else the test's temporary directory"""
    import tempfile
    try:
        st = os.statvfs("/dev/shm")
        if st.f_bavail * st.f_frsize > need + (16 << 30):
            return tempfile.mkdtemp(prefix="h10x_%s_" % name, dir="/dev/shm")
    except OSError:
        pass
    return str(tmp_path_factory.mktemp(name))


@pytest.mark.parametrize("case", MAN.get("big_digest_cases", []), ids=[c["name"] for c in MAN.get("big_digest_cases", [])])
def test_config3_proportions_match_reference_digests(case, tmp_path_factory, workdir):
    """BASELINE configs[2] (500 Mb x 2 haplotypes, 200 M pairs, 1 M barcodes, e = 0.1 %) at 1/10 and 1/4 scale: both run the
    RANKED placement of first[] by themselves (100 k barcodes as always; 300 k since the ranked form is tried wherever its bitmap
    leaves the whole-CU class room: nearly every block of that set sits there), and the 300 k set is then clustered again in the
    HASHED placement, which is what sets of a million barcodes get: same blocks, same labels. Several thousand barcodes take the
    > 255 clusters abort path. Expected = sha256 of the reference binary's canonical .hash, generated in the build container
    (tests/golden/make_golden.py --big: 4.5 and 12.5 minutes of the reference)."""
    import hashlib
    import hash10x_amd
    recs = _big_set(tmp_path_factory, case)
    assert orc.sha256(recs.tobytes()) == case["input_sha256"]
    h = hash10x_amd.Hash10x(B=case["B"])
    d = hash10x_amd.DeviceRecords(recs)
    h.read_fqb_device(d.ptr, d.n_records)
    d.free()
    a = case["args"]
    h.depth_range(int(a[1]), int(a[2]))
    h.cluster(int(a[4]), int(a[5]), 5)
    c = h.counters(); z = h.sizes()
    assert z["hashNumber"] == case["hash_number"] and z["nBlocks"] == case["blocks_max"] and z["nClusHash"] == case["sum_nHash"]
    assert c["cluster_first_mode"] == 1
    h.write_hash(workdir.file("big.hash"))
    if case["gen"]["barcodes"] > 262144:                      # the translated placement (what a million barcodes get) on the same state
        first = hashlib.sha256(h.export_blocks().tobytes() + h.export_clushash().tobytes()).hexdigest()
        h.set_option("cluster_first_global", 4)
        h.depth_range(int(a[1]), int(a[2]))
        h.cluster(int(a[4]), int(a[5]), 5)
        assert h.counters()["cluster_first_mode"] == 4
        assert hashlib.sha256(h.export_blocks().tobytes() + h.export_clushash().tobytes()).hexdigest() == first
    h.close()
    sha = hashlib.sha256()
    with open(workdir.file("big.hash"), "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            sha.update(blk)
    os.remove(workdir.file("big.hash"))
    assert sha.hexdigest() == case["sha256"]


@pytest.mark.parametrize("case", MAN.get("full_digest_cases", []), ids=[c["name"] for c in MAN.get("full_digest_cases", [])])
def test_config3_full_size_matches_reference_digest(case, tmp_path_factory):
    """BASELINE configs[2] AT ITS OWN SIZE: 200 M read pairs, 1 M barcodes, 500 Mb x 2 haplotypes, e = 0.1 %, --hashDepthRange 30 100
    --cluster 1 0, with the table size the reference accepts for it (-B 29; -B 28 dies, see the next test). The whole canonical .hash
    (15 GB) must have the sha256 of the REFERENCE binary's (oracle/_ref/hash10x_omp, 8 threads, in the build container: make_golden.py
    --full; 16 min of --readFQB and 14 of --cluster there). This is the regime the placement logic changes in: a million barcodes, hashed
    first[] tables, tens of thousands of blocks on the > 255 clusters path. The input is generated here by gen_fqb v2 (OpenMP, seeded)."""
    import hashlib
    import hash10x_amd
    sys.path.insert(0, orc.REPO)
    import bench
    g = case["gen2"]
    wl = dict(pairs=g["pairs"], barcodes=g["barcodes"], genome=g["genome"], err=g["err"], mol=g["mol"], snp=g["snp"], mol_len=g["mol_len"])
    recs, _first, total = bench.generate_v2(wl, g["seed"])
    assert total == g["pairs"] and recs.size == 30 * g["pairs"]
    for key, sl in (("head_sha256", recs[: 30 << 20]), ("tail_sha256", recs[-(30 << 20):])):          # 1 M records of each end (the whole image is 24 GB)
        assert hashlib.sha256(sl.tobytes()).hexdigest() == case["input"][key], "gen_fqb v2 is not reproducing the seeded input (%s)" % key
    d = hash10x_amd.DeviceRecords(recs)
    del recs
    h = hash10x_amd.Hash10x(B=case["B"])
    h.read_fqb_device(d.ptr, d.n_records)
    a = case["args"]
    h.depth_range(int(a[1]), int(a[2]))
    h.cluster(int(a[4]), int(a[5]), 5)
    z = h.sizes()
    assert z["hashNumber"] == case["hash_number"] and z["nBlocks"] == case["blocks_max"] and z["nClusHash"] == case["sum_nHash"]
    # BASELINE names -B 28 for this set: 101.9 M distinct hashes against the cap of 2^26 - 2 = 67.1 M — the reference dies there
    # (manifest.json "die_cases": c3_200m.B28, 11 minutes into --readFQB), and so must we, with its message
    h28 = hash10x_amd.Hash10x(B=28)
    with pytest.raises(hash10x_amd.Hash10xError, match="hashTableSize is too small"):
        h28.read_fqb_device(d.ptr, d.n_records)
    h28.close()
    d.free()
    dd = _roomy_dir(tmp_path_factory, "full", case["size"])
    out = os.path.join(dd, "full.hash")
    try:
        h.write_hash(out)
        h.close()
        digest, info = orc.canonical_file_digest(out)
    finally:
        import shutil
        shutil.rmtree(dd, ignore_errors=True)
    assert info["sum_nSubCluster"] == case["sum_nSubCluster"] and info["size"] == case["size"]
    assert digest == case["sha256"]


def test_config3_dies_at_the_table_size_of_BASELINE(tmp_path_factory):
    """BASELINE configs[2] says -B 28 for 200 M pairs. At that scale the set holds ~104 M distinct hashes (the six padding
    bases hashed at every read end, SURVEY F6, give ~0.3 new hashes per pair whatever the error rate) against the cap of
    2^26 - 2 = 67.1 M: the reference dies with "hashTableSize is too small" — shown in the build container on the 1/10 set
    at the proportional table size (-B 25: 10.43 M hashes against 8.39 M, tests/golden/manifest.json "die_cases"). The HIP
    path must die the same way."""
    import hash10x_amd
    case = MAN["die_cases"][0]
    recs = _big_set(tmp_path_factory, case)
    h = hash10x_amd.Hash10x(B=case["B"])
    d = hash10x_amd.DeviceRecords(recs)
    with pytest.raises(hash10x_amd.Hash10xError, match=case["message"]):
        h.read_fqb_device(d.ptr, d.n_records)
    d.free(); h.close()


# ---------------------------------------------------------------------------------------- chunk boundaries of readFQB
def test_all_A_barcode_run_ending_at_a_chunk_boundary(workdir):
    """hash10x.c:212 `if (!barcode) barcode = u[0]`: a run of the all-A barcode (word 0) that ends exactly
    where a chunk ends swallows the next barcode's run. The chunk boundaries are replayed from the run starts; -c also gives
    the reference's "chunkSize too small" for a barcode of chunkSize or more pairs."""
    import hash10x_amd
    recs = orc.gen_fqb(workdir.file("x.fqb"), 3000, 12, 40000, 0.003, 5, 3.0, 150, 3000).copy()
    starts = [0] + [i for i in range(1, recs.shape[0]) if recs[i, 0] != recs[i - 1, 0]] + [recs.shape[0]]
    chunk = 700
    assert max(b - a for a, b in zip(starts, starts[1:])) < chunk
    # replay the loop to find where chunks end, then give the barcode that holds a chunk end the word 0 and cut it there
    ends, pos, head = [], 0, 0
    while pos < recs.shape[0]:
        end = min(head + chunk, recs.shape[0]); ends.append(end)
        head = max(s for s in starts[:-1] if s <= end - 1); pos = end
    e = next(x for x in ends if x not in starts and x < recs.shape[0] - 300)
    run0 = max(s for s in starts if s < e)
    run1 = min(s for s in starts if s > e)
    # records run0 .. e-1 become the all-A barcode; e .. run1-1 keep their own: the zero run now ends at the chunk end e
    recs[run0:e, 0] = 0
    recs.tofile(workdir.file("q.fqb"))
    for c, expect_merge in ((chunk, True), (chunk + 1, False)):
        o = orc.Oracle(B=20); o.read_fqb(recs.reshape(-1), 0, c); o.write_hash(workdir.file("orc.hash"))
        h = hash10x_amd.Hash10x(B=20); h.read_fqb(recs.reshape(-1), 0, c); h.write_hash(workdir.file("hip.hash"))
        a, b = open(workdir.file("hip.hash"), "rb").read(), open(workdir.file("orc.hash"), "rb").read()
        assert a == b, "-c %d: %s" % (c, orc.describe_diff(a, b))
        assert (h.sizes()["nBlocks"] == len(starts) - 1 + 1) == expect_merge      # one barcode more (the cut), one less (the merge)
        h.read_fqb_file(workdir.file("q.fqb"), 0, c); h.write_hash(workdir.file("hip2.hash"))     # the file path streams and replays alike
        assert open(workdir.file("hip2.hash"), "rb").read() == b
        h.close()
    if orc.have_ref():
        r = orc.run_ref(["-B", 20, "-c", chunk, "--readFQB", "q.fqb", "--writeHash", "ref.hash"], workdir.path)
        assert r.returncode == 0, r.stderr.decode()
        h = hash10x_amd.Hash10x(B=20); h.read_fqb(recs.reshape(-1), 0, chunk); h.write_hash(workdir.file("hip.hash")); h.close()
        assert orc.canonical_hash_bytes(open(workdir.file("ref.hash"), "rb").read()) == open(workdir.file("hip.hash"), "rb").read()
    with pytest.raises(hash10x_amd.Hash10xError, match="chunkSize too small"):
        h = hash10x_amd.Hash10x(B=20); h.read_fqb(recs.reshape(-1), 0, 100)


def test_last_block_of_exactly_chunk_size_records(workdir):
    """unless -N ends it, the reference's read loop comes round once more at end of file and tests
    `chunkSize - b->nRead <= 0` before the fread that finds the end (hash10x.c:202-208): a file whose LAST block holds exactly
    chunkSize records dies with "chunkSize too small" although no block exceeds the chunk. -N equal to the record count ends
    the loop first (no death); -N beyond it does not. Oracle, reference binary and the HIP path (memory image, file, CLI, 2 shards)."""
    import subprocess
    import hash10x_amd
    src = orc.gen_fqb(workdir.file("x.fqb"), 3000, 12, 40000, 0.003, 5, 3.0, 150, 3000)
    starts = [0] + [i for i in range(1, src.shape[0]) if src[i, 0] != src[i - 1, 0]] + [src.shape[0]]
    parts = [src[starts[r]: starts[r] + 20] for r in range(4)] + [src[starts[4]: starts[4] + 30]]
    recs = np.concatenate(parts).astype(np.uint32); n = recs.shape[0]             # blocks of 20, 20, 20, 20 and a last one of 30 records
    recs.tofile(workdir.file("q.fqb"))
    exe = os.path.join(orc.REPO, "bin", "hash10x-amd")
    for N, c, dies in ((0, 30, True), (0, 31, False), (n, 30, False), (n + 5, 30, True), (n - 1, 30, False)):
        o = orc.Oracle(B=20)
        if dies:
            with pytest.raises(orc.OracleError, match="chunkSize too small"):
                o.read_fqb(recs.reshape(-1), N, c)
        else:
            o.read_fqb(recs.reshape(-1), N, c); o.write_hash(workdir.file("orc.hash"))
        for how in ("mem", "file"):
            h = hash10x_amd.Hash10x(B=20)
            read = (lambda: h.read_fqb(recs.reshape(-1), N, c)) if how == "mem" else (lambda: h.read_fqb_file(workdir.file("q.fqb"), N, c))
            if dies:
                with pytest.raises(hash10x_amd.Hash10xError, match="chunkSize too small"):
                    read()
            else:
                read(); h.write_hash(workdir.file("hip.hash"))
                assert open(workdir.file("hip.hash"), "rb").read() == open(workdir.file("orc.hash"), "rb").read(), (N, c, how)
            h.close()
        args = ["-B", "20", "-c", str(c)] + (["-N", str(N)] if N else []) + ["--readFQB", "q.fqb", "--writeHash", "cli.hash"]
        for gpus in (1, 2):
            g = subprocess.run([exe] + (["--gpus", str(gpus)] if gpus > 1 else []) + args, cwd=workdir.path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
            if dies:
                assert g.returncode == 255 and b"FATAL ERROR: chunkSize too small" in g.stderr, (N, c, gpus, g.stderr.decode()[-300:])
            else:
                assert g.returncode == 0, g.stderr.decode()
                assert open(workdir.file("cli.hash"), "rb").read() == open(workdir.file("orc.hash"), "rb").read(), (N, c, gpus)
        if orc.have_ref():
            r = orc.run_ref(["-B", 20, "-c", c] + (["-N", N] if N else []) + ["--readFQB", "q.fqb", "--writeHash", "ref.hash"], workdir.path)
            assert (r.returncode != 0 and b"chunkSize too small" in r.stderr) if dies else r.returncode == 0, (N, c, r.stderr.decode()[-300:])
            if not dies:
                assert orc.canonical_hash_bytes(open(workdir.file("ref.hash"), "rb").read()) == open(workdir.file("orc.hash"), "rb").read()


@pytest.mark.parametrize("gpus", [2, 3])
def test_sharded_read_hash_of_a_truncated_file_fails_on_every_rank(workdir, gpus):
    """every rank of a sharded --readHash preads its own ClusterHash records, so a file cut short inside the LAST
    rank's records fails on that rank alone ("read fail 3") — the others must hear of it before they enter the collective load
    instead of waiting for ever. Likewise a bad hash index inside one rank's cut ("corrupt hash file" from that rank's validation)."""
    import subprocess
    exe = os.path.join(orc.REPO, "bin", "hash10x-amd")
    data = orc.read_maybe_gz(os.path.join(orc.GOLDEN, "small.c_3_14_2.hash.gz"))
    hf = orc.HashFile(data)
    open(workdir.file("short.hash"), "wb").write(data[:-4000])                  # ends inside the last rank's records
    bad = bytearray(data); bad[-8:-4] = (hf.hash_number + 5).to_bytes(4, "little")  # last ClusterHash.hash: the last rank's cut only
    open(workdir.file("bad.hash"), "wb").write(bad)
    for name, msg in (("short.hash", b"read fail 3"), ("bad.hash", b"corrupt hash file")):
        g = subprocess.run([exe, "--gpus", str(gpus), "-B", "20", "--readHash", name, "--hashDepthRange", "3", "14", "--cluster", "1", "0"], cwd=workdir.path,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert g.returncode == 255 and b"FATAL ERROR" in g.stderr, g.stderr.decode()[-500:]
        assert msg in g.stderr or b"another rank failed" in g.stderr or b"failed to load its part" in g.stderr, g.stderr.decode()[-500:]


def test_cli_gpus_changed_between_commands(workdir):
    """--gpus after a state was loaded drops that state with its team (its context is bound to the old communicator):
    the commands that follow a new --readFQB give the golden bytes; a command that needs a state right after the change is told so."""
    import subprocess
    workdir.need("small.fqb.gz")
    exe = os.path.join(orc.REPO, "bin", "hash10x-amd")
    exp = orc.read_maybe_gz(os.path.join(orc.GOLDEN, "small.e2e.hash.gz"))
    tail = ["--readFQB", "small.fqb", "--hashDepthRange", "3", "14", "--cluster", "1", "0", "--writeHash"]
    g = subprocess.run([exe, "-B", "20", "-ct", "2", "--gpus", "3", "--readFQB", "small.fqb", "--gpus", "1"] + tail + ["a.hash", "--gpus", "2"] + tail + ["b.hash"],
                       cwd=workdir.path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert g.returncode == 0, g.stderr.decode()
    assert open(workdir.file("a.hash"), "rb").read() == exp and open(workdir.file("b.hash"), "rb").read() == exp
    g = subprocess.run([exe, "-B", "20", "--gpus", "2", "--readFQB", "small.fqb", "--gpus", "1", "--hashDepthRange", "3", "14", "--writeHash", "c.hash"],
                       cwd=workdir.path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert g.returncode == 255 and b"no hash state loaded" in g.stderr, g.stderr.decode()[-500:]


def _device_count():
    import hash10x_amd
    return hash10x_amd.device_count()


@pytest.mark.skipif("_device_count() < 2", reason="RCCL needs one GPU per rank: this box has fewer than 2")
def test_rccl_two_gpus_cli_and_two_processes(workdir):
    """RCCL with more than one rank (needs a box with two GPUs; skipped on the 1-GPU test boxes): (1) the C program's --gpus 2 builds its
    team with ncclCommInitAll and must give the golden bytes; (2) two PROCESSES, one GPU each, rendezvous by ncclUniqueId over a file
    (tests/shard_worker.py --comm rccl) through --clusterSplit and the slice-wise --writeHash, equal to the oracle."""
    import subprocess, sys
    workdir.need("small.fqb.gz")
    exe = os.path.join(orc.REPO, "bin", "hash10x-amd")
    exp = orc.read_maybe_gz(os.path.join(orc.GOLDEN, "small.e2e.hash.gz"))
    env = dict(os.environ, H10X_COMM="rccl")
    g = subprocess.run([exe, "-B", "20", "-ct", "2", "--gpus", "2", "--readFQB", "small.fqb", "--hashDepthRange", "3", "14", "--cluster", "1", "0", "--writeHash", "r.hash"],
                       cwd=workdir.path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
    assert g.returncode == 0, g.stderr.decode()[-1500:]
    assert b"communicator: rccl" in g.stdout
    assert open(workdir.file("r.hash"), "rb").read() == exp
    recs = orc.gen_fqb(workdir.file("x.fqb"), 40000, 180, 300000, 0.003, 61, 4.0, 150, 6000)
    o = orc.Oracle(B=20)
    o.read_fqb(recs); o.depth_range(4, 30); o.cluster(1, 0, 3); o.cluster_split(); o.depth_range(4, 30); o.cluster(1, 0, 3)
    o.write_hash(workdir.file("orc.hash"))
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shard_worker.py")
    ps = [subprocess.Popen([sys.executable, worker, str(r), "2", "rccl:" + workdir.file("uid.bin"), workdir.file("x.fqb"), "20", "4", "30", "3", workdir.file("p.hash")],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in ps]
    for p, (so, se) in zip(ps, outs):
        assert p.returncode == 0, se.decode()[-1500:]
    got, want = open(workdir.file("p.hash"), "rb").read(), open(workdir.file("orc.hash"), "rb").read()
    assert got == want, orc.describe_diff(got, want)


def test_bench_with_two_ranks_carries_a_parity_gate(workdir):
    """a correctness gate for every timed run — also with more than one rank. bench.py --gpus 2 as its launcher starts it
    (one process per rank, RANK / WORLD_SIZE / MASTER_* in the environment; socket backend: both ranks share this box's GPU), every rank
    generating only its own shard (gen_fqb v2): the line must say that the summed checksum of all ranks' blocks + ClusterHash records equals
    the one of the REFERENCE binary's .hash of that data set (manifest.json "bench_scale_digests", made by make_golden.py --scale)."""
    import subprocess, sys
    port = 32000 + os.getpid() % 2000
    ps = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        ps.append(subprocess.Popen([sys.executable, os.path.join(orc.REPO, "bench.py"), "--gpus", "2", "--comm", "socket", "--steps", "1", "--warmup", "1"],
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=workdir.path))
    outs = [p.communicate(timeout=600) for p in ps]
    for p, (so, se) in zip(ps, outs):
        assert p.returncode == 0, se.decode()[-1500:]
    line = json.loads(outs[0][0].decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["read_pairs"] == 5000000
    assert line["parity_vs_reference_digest"] == "identical", line
    assert outs[1][0].decode().strip() == ""                # one JSON line, on rank 0


@pytest.mark.skipif("genome3g-tenth-30M" not in MAN.get("strong_digests", {}), reason="no strong-scaling digest in the manifest")
@pytest.mark.parametrize("world", [1, 2])
def test_bench_strong_scaling_on_one_fixed_set(workdir, world):
    """bench.py --workload genome3g-tenth-30M --scaling strong (the 3 Gb workload of BASELINE configs[3] at 1/10: the same proportions and depth range): the SAME
    30 M-pair set on one rank and on two (socket backend: both share this box's GPU; every rank generates only its shard), and in both cases the line's checksum
    of all blocks and ClusterHash records equals the one of the REFERENCE binary's .hash of that set (manifest "strong_digests", make_golden.py --g3)."""
    import subprocess, sys
    port = 34000 + os.getpid() % 2000
    ps = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        ps.append(subprocess.Popen([sys.executable, os.path.join(orc.REPO, "bench.py"), "--workload", "genome3g-tenth-30M", "--gpus", str(world), "--comm", "socket", "--steps", "1", "--warmup", "1",
                                    "--no-secondary"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=workdir.path))
    outs = [p.communicate(timeout=900) for p in ps]
    for p, (so, se) in zip(ps, outs):
        assert p.returncode == 0, se.decode()[-1500:]
    line = json.loads(outs[0][0].decode().strip().splitlines()[-1])
    assert line["n_gpus"] == world and line["scaling"] == "strong" and line["config"]["read_pairs"] == 30000000 and line["config"]["workload"] == "genome3g-tenth-30M"
    assert line["parity_vs_reference_digest"] == "identical", line


def test_cli_streams_files_larger_than_one_slab(workdir):
    """bin/hash10x-amd reads the .fqb on a pool of reader threads into six page-locked slabs (64 MiB each; here, through H10X_SLAB_MIB, 4 and 9 MiB so that a 72 MB file cycles
    through the slots three times and slabs hold one and several 4 MiB pieces), out of a mapping of the file or by pread (H10X_NO_MMAP), with 16 readers or 3, uploads queued
    in file order beside the reads (h10x_ingest_fqb_async): the bytes of the library path on the same records every time, on one GPU and on 2 shards (every shard's range
    larger or smaller than a slab). A slab is a whole number of 120-byte records, which a round-3 build got wrong — bench.py's end-to-end check caught it."""
    import subprocess
    import hash10x_amd
    recs = orc.gen_fqb(workdir.file("x.fqb"), 600000, 2400, 3000000, 0.003, 77, 8.0, 150, 30000)
    assert recs.nbytes > (64 << 20)
    h = hash10x_amd.Hash10x(B=22); h.read_fqb(recs.reshape(-1)); h.depth_range(10, 60); h.cluster(1, 0, 4); h.write_hash(workdir.file("lib.hash")); h.close()
    exp = open(workdir.file("lib.hash"), "rb").read()
    exe = os.path.join(orc.REPO, "bin", "hash10x-amd")
    for gpus, extra in ((1, {}), (2, {}), (1, {"H10X_SLAB_MIB": "4"}), (2, {"H10X_SLAB_MIB": "9", "H10X_READERS": "3"}), (1, {"H10X_SLAB_MIB": "4", "H10X_NO_MMAP": "1"})):
        g = subprocess.run([exe] + (["--gpus", str(gpus)] if gpus > 1 else []) + ["-B", "22", "-ct", "4", "--readFQB", "x.fqb", "--hashDepthRange", "10", "60", "--cluster", "1", "0",
                           "--writeHash", "cli.hash"], cwd=workdir.path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, **extra))
        assert g.returncode == 0, g.stderr.decode()
        got = open(workdir.file("cli.hash"), "rb").read()
        assert got == exp, "--gpus %d %r: %s" % (gpus, extra, orc.describe_diff(got, exp))


def test_streaming_ingest_at_the_c_abi(workdir):
    """h10x_ingest_fqb: the reference reads the file chunk by chunk (hash10x.c:202-223) and so can a caller of the C ABI — chunks of 7
    records (barcode runs straddle them at random), of one record, and one chunk for everything, with and without h10x_ingest_reserve,
    all give the golden bytes of `--readFQB small.fqb --hashDepthRange 3 14 --cluster 1 0`; -N is the caller's (fewer records fed); an
    ingest given up half way leaves the context usable."""
    import hash10x_amd
    recs = np.frombuffer(orc.read_maybe_gz(os.path.join(orc.GOLDEN, "small.fqb.gz")), dtype=np.uint32).reshape(-1, 30)
    gold = orc.read_maybe_gz(os.path.join(orc.GOLDEN, "small.e2e.hash.gz"))
    h = hash10x_amd.Hash10x(B=20)
    for step, reserve in ((7, 0), (1, 0), (recs.shape[0], 0), (333, recs.shape[0]), (7, recs.shape[0])):
        h.ingest_fqb((recs[a: a + step] for a in range(0, recs.shape[0], step)), reserve=reserve)
        h.depth_range(3, 14); h.cluster(1, 0, 2); h.write_hash(workdir.file("ing.hash"))
        got = open(workdir.file("ing.hash"), "rb").read()
        assert got == gold, (step, reserve, orc.describe_diff(got, gold))
    # a barcode of chunkSize pairs dies as from a file (the chunk semantics are those of the whole image)
    with pytest.raises(hash10x_amd.Hash10xError, match="chunkSize too small"):
        h.ingest_fqb((recs[a: a + 50] for a in range(0, recs.shape[0], 50)), chunk=20)
    # an ingest that is never closed: dropped by reserve(0); the context reads a file afterwards as if nothing had happened
    hip = hash10x_amd.load_native()[0]
    assert hip.h10x_ingest_fqb(h._ctx(), recs[:100].ctypes.data, 100, 0) == 0
    assert hip.h10x_ingest_reserve(h._ctx(), 5) != 0 and hip.h10x_ingest_reserve(h._ctx(), 0) == 0
    h.read_fqb(recs.reshape(-1)); h.depth_range(3, 14); h.cluster(1, 0, 2); h.write_hash(workdir.file("ing.hash"))
    assert open(workdir.file("ing.hash"), "rb").read() == gold
    h.close()


def test_fault_between_fork_and_join_leaves_the_context_usable(workdir):
    """Five places fork independent launch classes onto side streams (mosh table classes, clusHash classes, good-list classes, cluster
    classes, the ordered sums beside the read merges). An error between fork and join must not hand the buffers of still running kernels
    back to the block cache: every region is made to fail once ("fault_inject"), the call reports it, and the
    SAME context then gives the oracle's bytes. The set has blocks of more than 3072 entries, so the block-sorting classes really fork."""
    import hash10x_amd
    recs = orc.gen_fqb(workdir.file("x.fqb"), 30000, 40, 600000, 0.003, 9, 6.0, 150, 20000).reshape(-1)
    o = orc.Oracle(B=20); o.read_fqb(recs); o.depth_range(3, 30); o.cluster(1, 0, 2); o.write_hash(workdir.file("orc.hash"))
    exp = open(workdir.file("orc.hash"), "rb").read()
    assert 6144 < int(orc.HashFile(exp).blocks["nHash"].max()) <= 8192     # all three block-sorting classes, none beyond them
    h = hash10x_amd.Hash10x(B=20)
    h.read_fqb(recs)                                          # a context (and its side streams) to inject into
    for region in (1, 2, 3, 4, 5):
        assert hash10x_amd.load_native()[0].h10x_set_option(h._ctx(), b"fault_inject", region) == 0     # (on the live context: the knob clears itself when it fires)
        with pytest.raises(hash10x_amd.Hash10xError, match="injected fault in fork/join region %d" % region):
            h.read_fqb(recs); h.depth_range(3, 30); h.cluster(1, 0, 2)
        h.read_fqb(recs); h.depth_range(3, 30); h.cluster(1, 0, 2); h.write_hash(workdir.file("hip.hash"))
        got = open(workdir.file("hip.hash"), "rb").read()
        assert got == exp, "after a fault in region %d: %s" % (region, orc.describe_diff(got, exp))
    h.close()


def test_corrupt_hash_file_is_refused(workdir):
    """--readHash checks what it later uses as an index: a hash index beyond hashNumber in clusHash or in
    hashIndex[] gives an error message, not a device fault."""
    import hash10x_amd
    data = bytearray(orc.read_maybe_gz(os.path.join(orc.GOLDEN, "small.c_3_14_2.hash.gz")))
    hf = orc.HashFile(bytes(data))
    bad = bytearray(data); bad[-8:-4] = (hf.hash_number + 5).to_bytes(4, "little")         # last ClusterHash.hash
    open(workdir.file("bad1.hash"), "wb").write(bad)
    bad = bytearray(data); bad[16:20] = (hf.hash_number + 1000).to_bytes(4, "little")       # hashIndex[0]
    open(workdir.file("bad2.hash"), "wb").write(bad)
    for f in ("bad1.hash", "bad2.hash"):
        h = hash10x_amd.Hash10x(B=20)
        with pytest.raises(hash10x_amd.Hash10xError, match="corrupt hash file"):
            h.read_hash(workdir.file(f))
        h.close()


def test_two_processes_over_the_socket_communicator(workdir):
    """One process per rank, as under torch.distributed.run / a C launcher: rendezvous, one context per process, every
    collective of the sharded path between PROCESSES (host-staged over TCP: RCCL will not put two ranks on this box's one
    GPU), incl. --clusterSplit and the slice-wise --writeHash into one file."""
    import subprocess, sys
    recs = orc.gen_fqb(workdir.file("x.fqb"), 40000, 180, 300000, 0.003, 61, 4.0, 150, 6000)
    o = orc.Oracle(B=20)
    o.read_fqb(recs); o.depth_range(4, 30); o.cluster(1, 0, 3); o.cluster_split(); o.depth_range(4, 30); o.cluster(1, 0, 3)
    o.write_hash(workdir.file("orc.hash"))
    exp = open(workdir.file("orc.hash"), "rb").read()
    port = 31000 + os.getpid() % 2000
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shard_worker.py")
    for n in (2, 3):
        ps = [subprocess.Popen([sys.executable, worker, str(r), str(n), str(port + 10 * n), workdir.file("x.fqb"), "20", "4", "30", "3", workdir.file("p%d.hash" % n)],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(n)]
        outs = [p.communicate(timeout=300) for p in ps]
        for p, (so, se) in zip(ps, outs):
            assert p.returncode == 0, se.decode()[-1500:]
        got = open(workdir.file("p%d.hash" % n), "rb").read()
        assert got == exp, orc.describe_diff(got, exp)


@pytest.mark.parametrize("gpus", [2, 3])
def test_cli_read_hash_onto_several_gpus(workdir, gpus):
    """The README's main recipe (`--readHash x.hash --hashDepthRange .. --cluster 1 0 --clusterSplit --writeHash`, README.md:29) under
    --gpus N: every rank loads the replicated tables and its own cut of the file's blocks, the hash owners' barcode lists are
    rebuilt by an exchange, and the result is the single-GPU / oracle file. Also a file that already holds split blocks."""
    import subprocess
    workdir.need("small.hash.gz")
    exe = os.path.join(orc.REPO, "bin", "hash10x-amd")
    tail = ["-B", "20", "-ct", "2", "--readHash", "small.hash", "--hashDepthRange", "3", "14", "--cluster", "1", "0", "--clusterSplit", "--writeHash", "OUT"]
    o = orc.Oracle(B=20); o.read_hash(workdir.file("small.hash")); o.depth_range(3, 14); o.cluster(1, 0, 2); o.cluster_split(); o.write_hash(workdir.file("orc.hash"))
    exp = open(workdir.file("orc.hash"), "rb").read()
    g = subprocess.run([exe, "--gpus", str(gpus)] + [a if a != "OUT" else "hip.hash" for a in tail], cwd=workdir.path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert g.returncode == 0, g.stderr.decode()
    got = open(workdir.file("hip.hash"), "rb").read()
    assert got == exp, orc.describe_diff(got, exp)
    # second generation: read the split file back onto the GPUs, cluster the split blocks, split again
    tail2 = ["-B", "20", "-ct", "2", "--readHash", "hip.hash", "--hashDepthRange", "3", "14", "--cluster", "1", "0", "--clusterSplit", "--codeStats", "--writeHash", "OUT"]
    o = orc.Oracle(B=20); o.read_hash(workdir.file("orc.hash")); o.depth_range(3, 14); o.cluster(1, 0, 2); o.cluster_split(); o.write_hash(workdir.file("orc2.hash"))
    g = subprocess.run([exe, "--gpus", str(gpus)] + [a if a != "OUT" else "hip2.hash" for a in tail2], cwd=workdir.path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert g.returncode == 0, g.stderr.decode()
    got, exp = open(workdir.file("hip2.hash"), "rb").read(), open(workdir.file("orc2.hash"), "rb").read()
    assert got == exp, orc.describe_diff(got, exp)
    # a file whose entries do not add up to its depths is refused by every rank together
    bad = bytearray(open(workdir.file("small.hash"), "rb").read()); hf = orc.HashFile(bytes(bad))
    bad[-8:-4] = (1).to_bytes(4, "little") if hf.clushash["hash"][-1] != 1 else (2).to_bytes(4, "little")
    open(workdir.file("bad.hash"), "wb").write(bad)
    g = subprocess.run([exe, "--gpus", str(gpus), "-B", "20", "--readHash", "bad.hash"], cwd=workdir.path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert g.returncode == 255 and b"FATAL ERROR" in g.stderr


# ---------------------------------------------------------------------------------------- BASELINE configs[4]: split + crib accuracy on the 3 Gb-shaped set
@pytest.mark.skipif("sha256" not in MAN.get("split_digests", {}).get("genome3g-tenth-30M", {}), reason="no split digest in the manifest")
def test_config5_split_crib_report_match_reference_digests(tmp_path_factory):
    """BASELINE configs[4] (-B .. --hashDepthRange --cluster --clusterSplit with the --cribBuild accuracy check) on the 3 Gb-shaped set at 1/10
    (30 M read pairs, 160 k barcodes, 300 Mb x 2, -B 27): bin/hash10x-amd runs the chain the reference ran in the build container (make_golden.py --g3split:
    --readFQB, --hashDepthRange 6 45, --cluster 1 0, --cribBuild of both truth haplotypes, --clusterReport over every barcode, --clusterSplit — 14.6 M new
    blocks —, --cribSummary, --writeHash), on one GPU and as --gpus 2 and 3 (ranks share this box's GPU; no gather). Expected, from the REFERENCE binary:
    sha256 of the whole -o text (1.45 GB: crib matches, every CLUSTER_SUMMARY / CODE_CLUSTER line, MIN_POINT_DENSITY, the split line, the summary), the
    accuracy figures read off it, and sha256 of the canonical split .hash (3.06 GB). hash10x.c:426-521, 870-952, 956-1061."""
    import hashlib
    import subprocess
    case = MAN["split_digests"]["genome3g-tenth-30M"]
    g = case["gen2"]
    d = _roomy_dir(tmp_path_factory, "c5", 14 << 30)          # .fqb 3.6 GB, two FASTAs, report text 1.45 GB, split .hash 3.06 GB
    subprocess.run([orc.build_gen(), "-v", "2", "-P", str(g["pairs"]), "-C", str(g["barcodes"]), "-G", str(g["genome"]), "-e", str(g["err"]), "-s", str(g["seed"]),
                    "-m", str(g["mol"]), "-S", str(g["snp"]), "-L", str(g["mol_len"]), "-o", os.path.join(d, "g3t.fqb"), "-fa", os.path.join(d, "g3t")], check=True, stderr=subprocess.DEVNULL)
    for hap in ("A", "B"):
        sha = hashlib.sha256()
        with open(os.path.join(d, "g3t.%s.fa" % hap), "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                sha.update(blk)
        assert sha.hexdigest() == case["fasta_sha256"][hap], "gen_fqb -fa is not reproducing the truth genome %s" % hap
    tail = [a if not a.endswith(".fifo") else a.replace(".report.fifo", ".report.txt").replace(".split.fifo", ".split.hash") for a in case["commands_after_cluster"]]
    try:
        for gpus in (1, 2, 3):
            cmd = [os.path.join(orc.REPO, "bin", "hash10x-amd")] + (["--gpus", str(gpus)] if gpus > 1 else []) + \
                  ["-B", str(case["B"]), "--readFQB", "g3t.fqb", "--hashDepthRange", "6", "45", "--cluster", "1", "0"] + tail
            r = subprocess.run(cmd, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert r.returncode == 0, "--gpus %d: %s" % (gpus, r.stderr.decode()[-1500:])
            if gpus == 1:                                     # the accuracy figures read off the text line by line (14 M CODE_CLUSTER lines: 15 s of Python) — once:
                rep_sha, rep = orc.report_digest(os.path.join(d, "g3t.report.txt"))   # the other rank counts must produce the same BYTES, which says the same and more
                for k in ("clusters", "clusters_without_OTHER", "clusters_located", "sum_span", "sum_reads", "sum_hashes", "size"):
                    assert rep[k] == case["report"][k], "--gpus %d: report figure %s = %r, reference %r" % (gpus, k, rep[k], case["report"][k])
            else:
                sha = hashlib.sha256()
                with open(os.path.join(d, "g3t.report.txt"), "rb") as f:
                    for blk in iter(lambda: f.read(1 << 24), b""):
                        sha.update(blk)
                rep_sha = sha.hexdigest()
                assert os.path.getsize(os.path.join(d, "g3t.report.txt")) == case["report"]["size"]
            assert rep_sha == case["report_sha256"], "--gpus %d: report text differs from the reference's" % gpus
            digest, info = orc.canonical_file_digest(os.path.join(d, "g3t.split.hash"))
            assert (info["hash_number"], info["blocks_max"], info["sum_nHash"], info["size"]) == (case["hash_number"], case["blocks_max"], case["sum_nHash"], case["size"]), (gpus, info)
            assert digest == case["sha256"], "--gpus %d: split .hash differs from the reference's" % gpus
            os.remove(os.path.join(d, "g3t.split.hash")); os.remove(os.path.join(d, "g3t.report.txt"))
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)


# ---------------------------------------------------------------------------------------- the whole .hash at -B 30
@pytest.mark.skipif("genome3g-300M" not in MAN.get("strong_digests", {}) or (os.cpu_count() or 1) < 16, reason="no digest in the manifest / too few host threads to generate 300 M pairs")
def test_genome3g_full_size_matches_reference_digest(tmp_path_factory):
    """The 3 Gb workload at the size its reference run is pinned (300 M read pairs, 1.6 M barcodes, -B 30, --hashDepthRange 6 45 --cluster 1 0): the WHOLE
    canonical .hash — hashIndex[2^30] (exactly 2^32 bytes: where a 32-bit byte offset wraps), hashValue[], both Array headers, hashDepth[], the blocks and 2.2 G
    ClusterHash records, 24.7 GB — has the sha256 and the size of the REFERENCE binary's (manifest "strong_digests", make_golden.py --g3: 42 minutes there).
    writeHashFile, hash10x.c:244-267. The file goes to the test's temporary directory; where that cannot hold it the five sections are digested from
    h10x_export_slice in file order instead (the same bytes without the file)."""
    import hashlib
    import hash10x_amd
    sys.path.insert(0, orc.REPO)
    import bench
    case = MAN["strong_digests"]["genome3g-300M"]
    wl = bench.WORKLOADS["genome3g-300M"]
    recs, _first, total = bench.generate_v2(wl, wl["seed"])
    assert total == wl["pairs"] and recs.size == 30 * wl["pairs"]
    for key, sl in (("head_sha256", recs[: 30 << 20]), ("tail_sha256", recs[-(30 << 20):])):
        assert hashlib.sha256(sl.tobytes()).hexdigest() == case["input"][key], "gen_fqb v2 is not reproducing the seeded input (%s)" % key
    d = hash10x_amd.DeviceRecords(recs)
    del recs
    h = hash10x_amd.Hash10x(B=case["B"])
    h.read_fqb_device(d.ptr, d.n_records)
    d.free()
    a = case["args"]
    h.depth_range(int(a[1]), int(a[2]))
    h.cluster(int(a[4]), int(a[5]), 5)
    z = h.sizes()
    assert z["hashNumber"] == case["hash_number"] and z["nBlocks"] == case["blocks_max"] and z["nClusHash"] == case["sum_nHash"]
    tmp = _roomy_dir(tmp_path_factory, "g3", case["size"])
    out = os.path.join(tmp, "g3.hash")
    st = os.statvfs(str(tmp))
    if st.f_bavail * st.f_frsize > case["size"] + (4 << 30):
        try:
            h.write_hash(out)
            h.close()
            digest, info = orc.canonical_file_digest(out)
        finally:
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)
        assert info["sum_nSubCluster"] == case["sum_nSubCluster"] and info["size"] == case["size"]
    else:
        digest, size = orc.slice_digest(h)
        h.close()
        assert size == case["size"]
    assert digest == case["sha256"]


def test_virtual_ranks_model_and_exchange_accounting(workdir):
    """bench.virtual_ranks_block on a small generator-v2 set: 4 ranks as threads taking turns on this GPU (h10x_comm_local_serialize). Every kind of exchange of the
    sharded --readFQB / --hashDepthRange reports bytes (h10x_exchange_get), the per-peer share never exceeds a rank's total, the ranks' compute sums to no less than
    the single-GPU step's device time would allow (nothing is lost in the subtraction of the waits), and the model's figures are there. The sharded RESULTS under the
    turnstile equal the unsharded ones (checksum of all blocks and records)."""
    import sys
    sys.path.insert(0, orc.REPO)
    import bench
    import hash10x_amd
    name = "vr-test"
    bench.WORKLOADS[name] = dict(pairs=400000, barcodes=2000, genome=4000000, err=0.001, mol=10.0, snp=150, mol_len=50000.0, B=22, lo=4, hi=40, ct=3, seed=5, gen=2)
    try:
        blk = bench.virtual_ranks_block(hash10x_amd, name, 4)
    finally:
        wl = bench.WORKLOADS.pop(name)
    assert "error" not in blk, blk
    assert blk["ranks"] == 4 and len(blk["per_rank"]) == 4 and sum(r["read_pairs"] for r in blk["per_rank"]) == wl["pairs"]
    for kind in ("entries_to_hash_owners (all-to-all)", "indices_back (all-to-all)", "index_depth (allgather)", "list_data (allgather: the in-range barcode lists)"):
        e = blk["exchanges"][kind]
        assert e["sum_bytes_out"] > 0 and 0 < e["busiest_peer_share_bytes"] <= e["max_rank_bytes_out"] <= e["sum_bytes_out"], (kind, e)
    assert all(r["compute_ms"] > 0 for r in blk["per_rank"]) and blk["max_rank_compute_ms"] >= blk["mean_rank_compute_ms"] > 0
    assert blk["modelled_step_ms"] >= blk["max_rank_compute_ms"] and blk["modelled_speedup_vs_1_gpu"] > 0
    # the same 4 ranks without the turnstile, against the unsharded state
    recs, _f, _t = bench.generate_v2(wl, wl["seed"])
    h = hash10x_amd.Hash10x(B=wl["B"]); h.read_fqb(recs); h.depth_range(wl["lo"], wl["hi"]); h.cluster(1, 0, wl["ct"])
    z = h.sizes()
    exp = bench.checksum_state(h.export_slice(3, 1, z["nBlocks"] - 1), 1, h.export_slice(4, 0, z["nClusHash"]), 0)
    h.close()
    import threading
    comms = hash10x_amd.Comm.local(4); comms[0].serialize(True)
    got, errs = [None] * 4, []

    def work(r):
        try:
            rr, _first, _tot = bench.generate_v2(wl, wl["seed"], r, 4)
            hh = hash10x_amd.Hash10x(B=wl["B"])
            d = hash10x_amd.DeviceRecords(rr)
            comms[r].turn_begin()
            try:
                hh.shard_read_fqb_device(comms[r], d.ptr, rr.size // 30); hh.depth_range(wl["lo"], wl["hi"]); hh.cluster(1, 0, wl["ct"])
                got[r] = bench.sharded_state_checksum(hh, r)
            finally:
                comms[r].turn_end(0)
            hh.close(); d.free()
        except Exception as e:
            errs.append(str(e))
    th = [threading.Thread(target=work, args=(r,)) for r in range(4)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    assert got[0] == exp and all(g == got[0] for g in got)
