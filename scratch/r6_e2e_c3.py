"""BASELINE configs[2] at its own size, file in -> .hash out through bin/hash10x-amd (bench.cli_end_to_end), with the host layer's own timing lines:
   python scratch/r6_e2e_c3.py [runs] [H10X_READERS values, comma-separated]   (GPU box; ~15 s of generation, then ~5 s per run)"""
import json, os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import bench
man = json.load(open(os.path.join(R, "tests", "golden", "manifest.json")))["full_digest_cases"][0]; g = man["gen2"]
wl = dict(pairs=g["pairs"], barcodes=g["barcodes"], genome=g["genome"], err=g["err"], mol=g["mol"], snp=g["snp"], mol_len=g["mol_len"])
t = time.perf_counter(); recs, _f, _t = bench.generate_v2(wl, g["seed"]); print("generated in %.1f s" % (time.perf_counter() - t), flush=True)
os.environ["H10X_INGEST_TIMING"] = "1"; os.environ["H10X_HOSTPROF"] = "1"
readers = sys.argv[2].split(",") if len(sys.argv) > 2 else [""]     # reader counts, or NAME=VALUE environment settings of the host layer (H10X_NO_MMAP=1, H10X_NO_POPULATE=1)
first = True
for rd in readers:
    for k in ("H10X_READERS", "H10X_NO_MMAP", "H10X_NO_POPULATE"): os.environ.pop(k, None)
    if "=" in rd: os.environ[rd.split("=")[0]] = rd.split("=")[1]
    elif rd: os.environ["H10X_READERS"] = rd
    for run in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
        e = bench.cli_end_to_end(recs, man["B"], 30, 100, 5, expect_sha256=man["sha256"] if first else None, expect_size=man["size"])
        first = False
        print("readers=%s " % (rd or "default") + json.dumps({k: e.get(k) for k in ("wall_seconds", "per_command_wall_seconds", "hash_identical_to_reference", "host_timing_lines")}), flush=True)
