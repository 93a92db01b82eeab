"""the genome3g block of bench.py alone (with the configs[4] extras) on a named generator-v2 workload:  python scratch/r5_g3block.py [genome3g-tenth-30M]"""
import sys, os, json
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import bench, hash10x_amd
name = sys.argv[1] if len(sys.argv) > 1 else "genome3g-tenth-30M"
print(json.dumps(bench.genome3g_block(hash10x_amd, 0, steps=2, name=name), indent=1))
