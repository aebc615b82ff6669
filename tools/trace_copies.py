import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last 3 steps: find adam kernels as step delimiters
idx = [i for i, n in enumerate(names) if "adam_kernel" in n]
a, b = idx[-3], idx[-2]
prev = collections.Counter()
for i in range(a, b):
    n = names[i]
    if "copyBuffer" in n:
        r = rows[i]
        print("copy grid=%s wg=%s dur=%.1fus stream=%s | prev: %s | next: %s" % (r.get("Grid_Size", r.get("Grid_Size_X")), r.get("Workgroup_Size", r.get("Workgroup_Size_X")), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Stream_Id", r.get("Queue_Id")), names[i - 1][:50], names[i + 1][:50] if i + 1 < len(names) else ""))
print("kernels in the step:", b - a)
