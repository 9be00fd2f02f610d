"""Per-kernel average duration by position in the run (quantiles of launch order): python scripts/trace_batches.py <kernel_trace.csv> [parts]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 5
by = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if name.startswith("hg::"):
        by[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for name, v in by.items():
    v.sort()
    n = len(v)
    out = []
    for p in range(parts):
        seg = v[p * n // parts:(p + 1) * n // parts]
        if seg:
            out.append("%.1f" % (sum(e - s for s, e in seg) / len(seg) / 1e3))
    print("%-40s n=%4d  avg us by fifth of the run: %s" % (name[:40], n, " ".join(out)))
t0 = min(int(r["Start_Timestamp"]) for r in rows if "hg::" in r["Kernel_Name"])
t1 = max(int(r["End_Timestamp"]) for r in rows if "hg::" in r["Kernel_Name"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if "hg::" in r["Kernel_Name"])
print("span %.2f ms, kernels busy %.2f ms" % ((t1 - t0) / 1e6, busy / 1e6))
