"""Register / spill / scratch figures per kernel from hipcc's resource-usage remarks (no GPU needed):
   python scratch/kres.py hash10x_amd/csrc/stage_c.hip [name-filter]"""
import re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage",
                    "-c", src, "-o", "/dev/null"] + sys.argv[3:], stderr=subprocess.PIPE, stdout=subprocess.PIPE)
cur = None
for line in r.stderr.decode().splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": subprocess.run(["c++filt", t.split(":", 1)[1].strip()], stdout=subprocess.PIPE).stdout.decode().strip()}
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size"):
            if flt in cur["name"]:
                print("%-70s sgpr %s vgpr %s agpr %s spillS %s spillV %s scratch %s occ %s" % (cur["name"][:70], cur.get("TotalSGPRs"), cur.get("VGPRs"), cur.get("AGPRs"),
                      cur.get("SGPRs Spill"), cur.get("VGPRs Spill"), cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]")))
            cur = None
