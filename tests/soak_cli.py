"""Randomised differential soak of the command line (test infrastructure): bin/hash10x-amd against the REFERENCE BINARY
(oracle/_ref/hash10x, run with the deterministic malloc settings of orc.run_ref) on the same random command lines — canonical
.hash bytes and every report line (--hashStats, --codeStats, --cribBuild, --clusterReport, --cribSummary). Stays inside
what the reference defines (threshold >= 1, one depth range per clustering). `python tests/soak_cli.py 100 3`."""
import os, random, shutil, subprocess, sys, tempfile, time

import orc

KEEP = ("HASH_COUNT_", "CODE_SIZE_", "CODE_CLUSTER_", "  crib matches", "    hom", "    het", "    mul", "    err", "CRIB_TABLE", "  CLUSTER_SUMMARY",
        "    CODE_CLUSTER", "  MIN_POINT_DENSITY")


def report(txt):
    return [ln for ln in txt.decode(errors="replace").splitlines()
            if ln.startswith(KEEP) or ln.startswith("  code ") or ln.startswith(" then ") or " base codes " in ln or " cluster codes " in ln or " in crib genome" in ln]


def run(n_cases, seed, verbose=False):
    rng = random.Random(seed)
    exe = os.path.join(orc.REPO, "bin", "hash10x-amd")
    bad = []
    for case in range(n_cases):
        d = tempfile.mkdtemp()
        k = rng.choice([21, 21, 16, 24, 19]); w = rng.choice([31, 31, 7, 13, 32]); r = rng.choice([17, 17, 5])
        pairs = rng.choice([500, 3000, 10000, 30000]); barcodes = min(rng.choice([3, 20, 60, 150]), pairs // 4); genome = rng.choice([12000, 60000, 300000])
        mol = rng.choice([2.0, 4.0]); mol_len = rng.choice([2500, 6000]); err = rng.choice([0.001, 0.003])
        lo = rng.choice([2, 3, 4]); hi = lo + rng.choice([3, 10, 30, 100]); ct = rng.choice([1, 2, 3, 5])
        B = 23 if w < 13 else 21
        gpus = rng.choice([1, 1, 2, 3, 4])                    # > 1: the sharded form of every command (ranks share the one GPU of the test box)
        args = ["-k", k, "-w", w, "-r", r, "-B", B, "-ct", ct] + (["--verbose"] if rng.random() < 0.4 else []) + ["--readFQB", "x.fqb"]   # --verbose: the per-barcode lines of --cluster
        if rng.random() < 0.5: args += ["--hashStats"]
        args += ["--hashDepthRange", lo, hi, "--cluster", 1, 0]
        if rng.random() < 0.5: args += ["--codeStats"]
        if rng.random() < 0.3:                                 # through a .hash file, read back on the same number of GPUs
            args += ["--writeHash", "mid.hash", "--readHash", "mid.hash", "--hashDepthRange", lo, hi]
        crib = rng.random() < 0.6
        if crib:
            if rng.random() < 0.5: args += ["--tables"]
            args += ["--cribBuild", "x.A.fa", "x.B.fa", "--clusterReport", 0, 0]
            if rng.random() < 0.5: args += ["--clusterReport", 1, max(2, barcodes // 2)]
        if rng.random() < 0.4:
            args += ["--clusterSplit"]
            if crib: args += ["--cribSummary"]
            args += ["--hashDepthRange", lo, hi, "--cluster", 1, 0]
            if rng.random() < 0.5: args += ["--codeStats"]
        args = [str(a) for a in args] + ["--writeHash", "OUT"]
        desc = dict(case=case, gpus=gpus, seed=2000 + case, pairs=pairs, barcodes=barcodes, genome=genome, mol=mol, mol_len=mol_len, err=err, args=" ".join(args))
        try:
            orc.gen_fqb(os.path.join(d, "x.fqb"), pairs, barcodes, genome, err, 2000 + case, mol, 150, mol_len, fa=os.path.join(d, "x"))
            if crib and rng.random() < 0.5:                  # cut the haplotypes into several sequences
                step = rng.choice([100, 500])
                for hap in ("A", "B"):
                    lines = open(os.path.join(d, "x.%s.fa" % hap)).read().splitlines()[1:]
                    with open(os.path.join(d, "x.%s.fa" % hap), "w") as f:
                        for i in range(0, len(lines), step):
                            f.write(">c%d\n%s\n" % (i // step + 1, "\n".join(lines[i:i + step])))
            ref = orc.run_ref([a if a != "OUT" else "ref.hash" for a in args], d)
            hip = subprocess.run([exe] + (["--gpus", str(gpus)] if gpus > 1 else []) + [a if a != "OUT" else "hip.hash" for a in args], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            why = None
            if ref.returncode != hip.returncode: why = "exit codes %d (ref) vs %d: %s | %s" % (ref.returncode, hip.returncode, ref.stderr.decode()[-200:], hip.stderr.decode()[-200:])
            elif ref.returncode == 0:
                exp = orc.canonical_hash_bytes(open(os.path.join(d, "ref.hash"), "rb").read()); got = open(os.path.join(d, "hip.hash"), "rb").read()
                if got != exp: why = ".hash: " + orc.describe_diff(got, exp)[:300]
                else:
                    a, b = report(hip.stdout), report(ref.stdout)
                    if a != b:
                        i = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
                        why = "report line %d of %d/%d:\n  hip: %s\n  ref: %s" % (i, len(a), len(b), a[i] if i < len(a) else None, b[i] if i < len(b) else None)
            if verbose: print("case %3d %s %s" % (case, "ok " if not why else "MISMATCH", desc["args"] if why else ""), flush=True)
            if why:
                desc["why"] = why; bad.append(desc)
                keep = os.environ.get("H10X_SOAK_KEEP")
                if keep:                                      # leave the evidence behind (a directory under gpurun_out/, say)
                    os.makedirs(keep, exist_ok=True)
                    for fn in ("hip.hash", "ref.hash"):
                        if os.path.exists(os.path.join(d, fn)): shutil.copy(os.path.join(d, fn), os.path.join(keep, "case%d.%s" % (case, fn)))
                    open(os.path.join(keep, "case%d.hip.out" % case), "wb").write(hip.stdout + b"\n--stderr--\n" + hip.stderr)
        except Exception as e:                                # noqa: BLE001
            desc["exception"] = repr(e); bad.append(desc)
            if verbose: print("case %3d EXCEPTION %r" % (case, e), flush=True)
        shutil.rmtree(d, ignore_errors=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    t0 = time.time()
    bad = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 3, verbose=True)
    for b in bad: print("BAD", b)
    print("cli soak: %d cases, %d bad, %.0f s" % (n, len(bad), time.time() - t0))
    sys.exit(1 if bad else 0)
