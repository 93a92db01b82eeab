import sys, os, json, subprocess
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
out = subprocess.run([sys.executable, R + "/bench.py", "--steps", "5", "--warmup", "2", "--sharded", "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
d = json.loads(out.decode().strip().splitlines()[-1])
print("sharded N=1:", d["ms_per_step"], d["device_ms_per_step"], d["host_wall_ms_per_step"])
