import csv, glob, sys, collections
tag = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else "gemm"
for k in (1, 2, 3):
    files = glob.glob("gpurun_out/pmc_%s_%d/**/*counter_collection.csv" % (tag, k), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, cs in agg.items():
        print(tag, k, kn, " ".join("%s=%.4g" % (c, sum(v[2:]) / max(1, len(v[2:]))) for c, v in sorted(cs.items())))
