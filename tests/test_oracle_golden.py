"""The CPU oracle restatement against the golden vectors produced by the real reference
(tests/golden/make_golden.py). CPU only."""
import json
import os

import numpy as np
import pytest

import orc
from driver import run_commands


def _cases():
    with open(os.path.join(orc.GOLDEN, "manifest.json")) as f:
        return json.load(f)


MAN = _cases()


def test_kat1_factor1_from_glibc_random():
    # SURVEY KAT-1 / Appendix D.1 (measured on the reference): srandom(17) => factor1
    assert orc.lib().orc_factor1_from_seed(17) == 0x49308BB9003CB3AD
    with open(os.path.join(orc.GOLDEN, "kat.json")) as f:
        kat = json.load(f)
    assert int(kat["seed17_factor1"], 16) == 0x49308BB9003CB3AD


def test_kat2_seqhash_test_main():
    """Mosh lists printed by the reference's own -DTEST main (seqhash.c:199-221), k=16 w=32, default seed."""
    with open(os.path.join(orc.GOLDEN, "kat.json")) as f:
        kat = json.load(f)["seqhash_test_k16_w32_default_seed"]
    o = orc.Oracle(k=16, w=32, seed=1, B=20)   # srandom(1) == glibc's default state
    total = 0
    for s, exp in zip(kat["sequences"], kat["out"]):
        codes = np.array(["ACGT".index(c) for c in s], dtype=np.uint8)
        hs, ps = o.mosh(codes)
        assert exp["len"] == len(s)
        assert [int(h, 16) for h, _, _ in exp["moshes"]] == [int(h) for h in hs]
        assert [p for _, p, _ in exp["moshes"]] == [int(p) for p in ps]
        total += len(hs)
    assert total > 5


@pytest.mark.parametrize("case", MAN["cases"], ids=[c["name"] for c in MAN["cases"]])
def test_oracle_matches_reference_golden(case, workdir):
    workdir.need(case["input"])
    args = list(case["args"])
    out = args[-1]
    run_commands(lambda k, w, r, B: orc.Oracle(k, w, r, B), args, workdir.path)
    got = open(workdir.file(out), "rb").read()
    exp = orc.read_maybe_gz(os.path.join(orc.GOLDEN, case["output"]))
    assert orc.sha256(exp) == case["sha256"]
    assert got == exp, orc.describe_diff(got, exp)


@pytest.mark.parametrize("case", MAN["digest_cases"], ids=[c["name"] for c in MAN["digest_cases"]])
def test_oracle_matches_reference_digests(case, workdir):
    recs = orc.digest_input(workdir.file(case["name"] + ".fqb"), case["gen"])
    assert orc.sha256(recs.tobytes()) == case["input_sha256"], "gen_fqb is not reproducing the seeded input"
    extra = list(case["args"])
    pre = orc.leading_options(extra)
    args = ["-B", case["B"]] + pre + ["--readFQB", case["name"] + ".fqb"] + extra[len(pre):] + ["--writeHash", "out.hash"]
    run_commands(lambda k, w, r, B: orc.Oracle(k, w, r, B), args, workdir.path)
    got = open(workdir.file("out.hash"), "rb").read()
    hf = orc.HashFile(got)
    assert hf.hash_number == case["hash_number"] and hf.blocks_max == case["blocks_max"]
    assert int(hf.blocks["nSubCluster"].sum()) == case["sum_nSubCluster"]
    assert len(got) == case["size"] and orc.sha256(got) == case["sha256"]


def test_oracle_die_conditions():
    with pytest.raises(orc.OracleError, match="out of range 20-30"):
        orc.Oracle(B=19)
    with pytest.raises(orc.OracleError, match="must be > 0"):
        orc.Oracle(k=0)
    with pytest.raises(orc.OracleError, match="between 1 and 32"):
        orc.Oracle(k=32)
    recs = np.fromfile(os.path.join(orc.GOLDEN, "tiny.fqb"), dtype=np.uint32)
    o = orc.Oracle(B=20)
    with pytest.raises(orc.OracleError, match="chunkSize too small"):
        o.read_fqb(recs, 0, 3)      # barcode C holds 4 pairs (hash10x.c:206)
