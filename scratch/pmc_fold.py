"""Fold the passes of scratch/prof_all.sh <tag> into profiles/:
  <tag>_bench.json, <tag>_bench_under_rocprof.json, <tag>_kernel_stats.csv   (copied)
  <tag>_pmc_traffic.json   FETCH_SIZE / WRITE_SIZE bytes per step and kernel (dispatches of the profiled script's LAST step)
  <tag>_pmc_sq.json        every other counter, summed over the dispatches of each kernel in the last step, plus derived ratios
usage: pmc_fold.py <tag> [steps_in_script=2] [suffix]   (suffix: profile names become <tag><suffix>_pmc_*.json)"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
suffix = sys.argv[3] if len(sys.argv) > 3 else ""
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")


def short(name):
    return name.split("(")[0].replace("void ", "")[:110]


def load(d):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    per = collections.defaultdict(lambda: collections.defaultdict(list))       # kernel -> counter -> [(dispatch, value)]
    meta = {}
    for r in rows:
        k = short(r["Kernel_Name"])
        per[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        meta[k] = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "scratch": int(r["Scratch_Size"]), "lds": int(r["LDS_Block_Size"]), "wg": int(r["Workgroup_Size"])}
    return per, meta


def last_step(vals):
    vals = sorted(vals)
    n = len(vals) // steps
    return [v for _, v in vals[len(vals) - n:]] if n else [v for _, v in vals]


for name in ("bench.json", "bench_under_rocprof.json"):
    src = os.path.join(G, "%s_%s" % (tag, name))
    if os.path.exists(src) and not suffix:
        shutil.copy(src, os.path.join(P, "%s_%s" % (tag, name)))
for f in glob.glob(os.path.join(G, tag + "_stats", "**", "*kernel_stats.csv"), recursive=True):
    if not suffix:
        shutil.copy(f, os.path.join(P, tag + "_kernel_stats.csv"))

# what was measured: the commit the tree was at when it was sent to the GPU box, and the library's embedded source hash (h10x_build_id):
# bench.py copies them into the line and flags the traffic figure as stale when the running library is another build
meta_path = os.path.join(G, tag + "_meta.json")
META = json.load(open(meta_path)) if os.path.exists(meta_path) else {}

interesting = ("h10x::", "radix_sort", "segmented", "onesweep")
traffic = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no trace flags) -- python3 <script>; dispatches of the last step",
           "unit": "bytes (counter value x 1024)",
           "note": "raw counters; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads 1/2 of wide (16 B/lane) coalesced reads; 4-8 B/lane gathers are uncalibrated",
           "dominant": os.environ.get("H10X_DOMINANT") or ("cluster_kernel<true, 1, 1024, 0>" if "config3" in tag else "cluster_kernel<true, 0, 1024, 0>"),
           "head": META.get("head"), "build_id": META.get("build_id"), "script": META.get("script"), "kernels": {}}
for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    per, _ = load(os.path.join(G, "%s_%s" % (tag, sub)))
    for k, cs in per.items():
        if not any(x in k for x in interesting) or counter not in cs:
            continue
        v = last_step(cs[counter])
        traffic["kernels"].setdefault(k, {})[counter] = {"dispatches_per_step": len(v), "bytes_per_step": sum(v) * 1024.0}
if traffic["kernels"]:
    json.dump(traffic, open(os.path.join(P, "%s%s_pmc_traffic.json" % (tag, suffix)), "w"), indent=1)

sq = {"source": "rocprofv3 --pmc <8 SQ counters per pass> (passes sqA..sqD, tcc of scratch/prof_all.sh; no trace flags); sums over the dispatches of the last step",
      "units": "SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves (MI355X_MICROARCH.md); SQ_BUSY_CYCLES per SE; GRBM_GUI_ACTIVE summed over XCDs",
      "head": META.get("head"), "build_id": META.get("build_id"), "script": META.get("script"), "kernels": {}}
for sub in ("sqA", "sqB", "sqC", "sqD", "tcc"):
    per, meta = load(os.path.join(G, "%s_%s" % (tag, sub)))
    for k, cs in per.items():
        if not any(x in k for x in interesting):
            continue
        e = sq["kernels"].setdefault(k, {"resources": meta[k], "counters": {}})
        for c, vals in cs.items():
            v = last_step(vals)
            e["counters"][c] = sum(v)
            e["dispatches_per_step"] = len(v)
for k, e in sq["kernels"].items():
    c = e["counters"]
    d = {}
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS"):
            if n in c:
                d[n + "/WAVE_CYCLES"] = round(c[n] / wc, 4)
    if c.get("SQ_WAVES"):
        for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH", "SQ_INSTS_SMEM", "SQ_INSTS_FLAT", "SQ_INSTS_LDS_ATOMIC"):
            if n in c:
                d[n + "/wave"] = round(c[n] / c["SQ_WAVES"], 1)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_share"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 4)
    if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)):
        d["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_INSTS_VALU"):
        # VALU issue: one wave-instruction occupies a SIMD for 2 cycles (wave64 on SIMD-32... see guide: 4 for one wave alone); 256 CUs x 4 SIMDs
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        d["kernel_cycles_per_xcd_avg"] = round(cyc)
        d["valu_insts_per_simd_cycle"] = round(c["SQ_INSTS_VALU"] / (cyc * 1024), 4)
        if c.get("SQ_ACTIVE_INST_VALU"):
            d["valu_busy_share_of_simd_time"] = round(4.0 * c["SQ_ACTIVE_INST_VALU"] / (cyc * 1024), 4)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_active_share_of_cu_time"] = round(c["SQ_LDS_IDX_ACTIVE"] / (cyc * 256), 4)
    e["derived"] = d
if sq["kernels"]:
    json.dump(sq, open(os.path.join(P, "%s%s_pmc_sq.json" % (tag, suffix)), "w"), indent=1)
for k, e in sq["kernels"].items():
    if any(x in k for x in ("cluster_kernel", "mosh_lds", "lookup_pack", "probe_insert", "assign_index")):
        print(k, e["resources"], json.dumps(e["derived"]))
for k, e in traffic["kernels"].items():
    if any(x in k for x in ("cluster_kernel", "mosh_lds", "lookup_pack", "probe_insert", "assign_index")):
        print(k, {c: round(v["bytes_per_step"] / 1e6, 1) for c, v in e.items()}, "MB/step")
