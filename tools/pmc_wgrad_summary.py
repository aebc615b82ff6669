"""Dev tool: per-counter means of one kernel from the passes of tools/pmc_wgrad.sh / tools/pmc_gather.sh.
usage: pmc_wgrad_summary.py dec0bf3 [pmcw wgrad]   |   pmc_wgrad_summary.py dec3f pmcg gather"""
import collections, csv, glob, os, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
tag = sys.argv[1]
prefix = sys.argv[2] if len(sys.argv) > 2 else "pmcw"
pat = sys.argv[3] if len(sys.argv) > 3 else "wgrad"
agg = collections.defaultdict(list)
dur = []
for f in glob.glob(os.path.join(root, "gpurun_out", "%s_%s_*" % (prefix, tag), "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(root, "gpurun_out", "%s_%s_*" % (prefix, tag), "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(tag, "kernel us (profiled passes): median %.1f over %d launches" % (sorted(dur)[len(dur) // 2] if dur else 0, len(dur)))
for k in sorted(agg):
    v = agg[k][2:] or agg[k]
    print("  %-40s %14.4g" % (k, sum(v) / len(v)))
