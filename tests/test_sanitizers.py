"""The CPU C code under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on this
pool, so only host code is covered): the oracle restatement end to end, fq2b-amd on gzipped FASTQ with a whitelist, and the
argument / file handling of hash10x-amd. The GPU-marked case runs the sanitized host program through a whole command line."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import orc

REPO = orc.REPO
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=66", UBSAN_OPTIONS="halt_on_error=1:exitcode=67")


def _built(target_dir, make_target, path):
    if not os.path.exists(path):
        subprocess.run(["make", "-C", os.path.join(REPO, target_dir), make_target], check=True, stdout=subprocess.DEVNULL)
    return path


def _clean(r):
    assert r.returncode not in (66, 67) and b"ERROR: AddressSanitizer" not in r.stderr and b"runtime error:" not in r.stderr, r.stderr.decode(errors="replace")[-2000:]


def test_oracle_under_sanitizers(tmp_path):
    drv = _built("oracle", "asan_driver", os.path.join(REPO, "oracle", "asan_driver"))
    small = tmp_path / "small.fqb"
    small.write_bytes(orc.read_maybe_gz(os.path.join(orc.GOLDEN, "small.fqb.gz")))
    for fqb, lo, hi, ct in ((str(small), 3, 14, 2), (os.path.join(orc.GOLDEN, "tiny.fqb"), 1, 4, 1)):
        r = subprocess.run([drv, fqb, "20", str(lo), str(hi), str(ct), str(tmp_path / "o.hash")], env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        _clean(r)
        assert r.returncode == 0, r.stderr.decode()


def test_fq2b_under_sanitizers(tmp_path):
    exe = _built(os.path.join("hash10x_amd", "host"), "asan", os.path.join(REPO, "build", "fq2b-amd-asan"))
    rng = np.random.default_rng(9)
    wl = ["".join(rng.choice(list("ACGT"), 16)) for _ in range(30)]
    with gzip.open(tmp_path / "r1.fq.gz", "wt") as f1, gzip.open(tmp_path / "r2.fq.gz", "wt") as f2:
        for i in range(300):
            bc = list(wl[rng.integers(len(wl))])
            if rng.random() < 0.3:
                bc[rng.integers(16)] = "ACGTN"[rng.integers(5)]
            s1 = "".join(bc) + "".join(rng.choice(list("ACGTN"), 135)); s2 = "".join(rng.choice(list("acgtN"), 151))
            q = "".join(chr(int(c)) for c in rng.integers(35, 75, 151))
            f1.write("@r%d\n%s\n+\n%s\n" % (i, s1, q)); f2.write("@r%d\n%s\n+\n%s\n" % (i, s2, q))
    (tmp_path / "wl.txt").write_text("\n".join(wl) + "\n")
    for opts in ([], ["-10x", "wl.txt"]):
        r = subprocess.run([exe] + opts + ["-o", "x.fqb", "r1.fq.gz", "r2.fq.gz"], cwd=tmp_path, env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        _clean(r)
        assert r.returncode == 0 and (tmp_path / "x.fqb").stat().st_size % 120 == 0
    # malformed input: a truncated last entry and a missing '+' line must die with a message, not with a sanitizer report
    with gzip.open(tmp_path / "bad.fq.gz", "wt") as f:
        f.write("@a\nACGT\n+\nIIII\n@b\nACGT\n-\nIIII\n")
    r = subprocess.run([exe, "-o", "y.fqb", "bad.fq.gz"], cwd=tmp_path, env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    _clean(r)
    assert r.returncode != 0 and b"FATAL ERROR" in r.stderr
    with gzip.open(tmp_path / "bad2.fq.gz", "wt") as f:
        f.write("@a\nACGT\n+\nIIII\n@b\nACG")
    r = subprocess.run([exe, "-o", "y.fqb", "bad2.fq.gz"], cwd=tmp_path, env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    _clean(r)
    assert r.returncode != 0 and b"FATAL ERROR" in r.stderr


def test_cli_argument_handling_under_sanitizers(tmp_path):
    """what hash10x-amd does before it needs a GPU: the argv grammar, the die() texts, unreadable and malformed files"""
    exe = _built(os.path.join("hash10x_amd", "host"), "asan", os.path.join(REPO, "build", "hash10x-amd-asan"))
    env = dict(ENV, ASAN_OPTIONS=ENV["ASAN_OPTIONS"].replace("detect_leaks=1", "detect_leaks=0"))     # the HIP runtime keeps its own allocations
    (tmp_path / "junk.hash").write_bytes(b"10XH" + bytes(40))
    (tmp_path / "short.fqb").write_bytes(bytes(100))
    cases = [([], 0, b"Usage: hash10x-amd"), (["readFQB"], 255, b"does not start with '-'"), (["--nonsense"], 255, b"unknown option/command"),
             (["--readFQB"], 255, b"unknown option/command"), (["--readFQB", "missing.fqb"], 255, b"failed to open fqb file"),
             (["-B", "28", "--gpus", "3", "--readHash", "junk.hash"], 255, b"FATAL ERROR"), (["--gpus", "0"], 255, b"--gpus 0"),
             (["-o", "/nonexistent/dir/x", "--help"], 0, b"can't open output file")]
    for args, rc, text in cases:
        r = subprocess.run([exe] + args, cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        _clean(r)
        assert r.returncode == rc and text in r.stderr, (args, r.returncode, r.stderr[-300:])


@pytest.mark.gpu
def test_cli_whole_command_line_under_sanitizers(tmp_path):
    """the sanitized host program (ASan + UBSan on h10x_host.c / hash10x_main.c; the device library as shipped) through
    --readFQB .. --writeHash, --readHash, the reports and --gpus 2: same bytes as the golden file, no sanitizer report"""
    exe = _built(os.path.join("hash10x_amd", "host"), "asan", os.path.join(REPO, "build", "hash10x-amd-asan"))
    # use_sigaltstack=0: with the HIP runtime in the process ASan cannot unmap the alternate signal stack of a finished rank
    # thread ("failed to deallocate ... UnsetAlternateSignalStack"), an ASan-internal check, not a finding in our code
    env = dict(ENV, ASAN_OPTIONS="detect_leaks=0:exitcode=66:use_sigaltstack=0")
    (tmp_path / "small.fqb").write_bytes(orc.read_maybe_gz(os.path.join(orc.GOLDEN, "small.fqb.gz")))
    gold = orc.read_maybe_gz(os.path.join(orc.GOLDEN, "small.e2e.hash.gz"))
    for pre in ([], ["--gpus", "2"]):
        r = subprocess.run([exe] + pre + ["-B", "20", "-ct", "2", "--readFQB", "small.fqb", "--hashStats", "--hashDepthRange", "3", "14", "--cluster", "1", "0",
                                          "--codeStats", "--clusterReport", "0", "0", "--writeHash", "out.hash"], cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        _clean(r)
        assert r.returncode == 0, r.stderr.decode()[-1000:]
        assert (tmp_path / "out.hash").read_bytes() == gold
    r = subprocess.run([exe, "-B", "20", "--readHash", "out.hash", "--clusterSplit", "--codeStats", "--writeHash", "split.hash"], cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    _clean(r)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
