import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    if not any(x in k for x in ("cluster_kernel", "mosh_lds", "lookup_pack", "probe_insert")): continue
    print(k, {c: round(v / max(cnt[(k, c)], 1)) for c, v in agg[k].items()}, "dispatches", max(cnt[(k, c)] for c in agg[k]))
