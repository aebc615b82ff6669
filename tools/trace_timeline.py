"""Dev tool: timeline of ONE step from a rocprofv3 --kernel-trace CSV (the last replayed step: from one adam_kernel to the
next).  usage: python tools/trace_timeline.py <kernel_trace.csv> [step_from_end]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
a, b = adam[-back - 1], adam[-back]
t0 = int(rows[a]["End_Timestamp"])
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)[:34]
print("step wall %.1f us (adam end -> adam end)" % ((int(rows[b]["End_Timestamp"]) - t0) / 1e3))
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%8.1f %8.1f %7.1f  q%-2s grid %7d lds %6s  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r["Queue_Id"],
          int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["LDS_Block_Size"], short(r["Kernel_Name"])))
