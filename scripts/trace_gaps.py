"""Diagnostic (not part of the product): where the GPU idles inside a registration step.

Reads a rocprofv3 kernel trace (csv) of `python3 bench.py ...` and prints, for the last STEPS steps
(a step starts at every launch of the marker kernel, default k_bin_count: the first kernel of an insertion; third argument), busy and idle time per step and the idle time by (previous
kernel -> next kernel) pair.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/t -o t -- python3 bench.py --steps 20 --no-cpu-baseline
    python scripts/trace_gaps.py gpurun_out/t/.../t_kernel_trace.csv 20
"""
import collections
import csv
import sys


def short(name):
    name = name.split("(")[0].replace("void ", "")
    return name.split("<")[0].replace("hg::", "")


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    marker = sys.argv[3] if len(sys.argv) > 3 else "k_bin_count"
    starts = [i for i, r in enumerate(rows) if r[2] == marker]
    if len(starts) < steps + 1:
        print("only", len(starts), marker, "launches found")
        return
    first = starts[-steps - 1]
    last = starts[-1]
    seg = rows[first:last]
    span = (seg[-1][1] - seg[0][0]) / 1e3
    busy = sum(e - s for s, e, _ in seg) / 1e3
    gaps = collections.defaultdict(lambda: [0.0, 0])
    per_kernel = collections.defaultdict(lambda: [0.0, 0])
    for (s0, e0, n0), (s1, e1, n1) in zip(seg, seg[1:]):
        g = gaps[(n0, n1)]
        g[0] += max(0, s1 - e0) / 1e3
        g[1] += 1
    for s, e, n in seg:
        per_kernel[n][0] += (e - s) / 1e3
        per_kernel[n][1] += 1
    print("steps %d: span %.1f us/step, kernels busy %.1f us/step, idle %.1f us/step" % (
        steps, span / steps, busy / steps, (span - busy) / steps))
    print("kernel time per step:")
    for n, (t, c) in sorted(per_kernel.items(), key=lambda kv: -kv[1][0]):
        print("  %-36s %8.1f us/step  %6.2f launches/step  %7.2f us/launch" % (n, t / steps, c / steps, t / c))
    print("idle by (previous -> next) per step:")
    for (a, b), (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]:
        print("  %-30s -> %-30s %7.1f us/step  (%5.2f per step, %5.2f us each)" % (a, b, t / steps, c / steps, t / c))


if __name__ == "__main__":
    main()
