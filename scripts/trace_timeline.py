"""Timeline of the last N kernels of a rocprofv3 kernel trace (start / end relative to the first, queue id):
python scripts/trace_timeline.py <kernel_trace.csv> [N]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "hg::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
seg = rows[-n:]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hg::", "")
    print("%9.1f %9.1f  %6.1f us  q=%s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), name[:50]))
