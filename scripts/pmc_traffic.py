"""Aggregates two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide
prescribes) of `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline` into
profiles/r02_pmc_traffic.json (or the path given as third argument): HBM-side bytes per launch and kernel.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o p -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o p -- python3 bench.py ...
    python scripts/pmc_traffic.py gpurun_out/pmc_f/p_counter_collection.csv gpurun_out/pmc_w/p_counter_collection.csv
"""
import collections
import csv
import json
import os
import sys
import os as _os
sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from hectorgrapher_amd._lib import source_digest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, counter):
    tot, launches = collections.defaultdict(float), collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        launches[name].append(float(r["Counter_Value"]) * 1024.0)  # counter unit: KiB
    out = {}
    for name, vals in launches.items():
        if "k_tsdf_residuals" in name or "k_window_residuals" in name or name.endswith("k_lm"):
            # launches enqueued after the solver terminated exit at once and move (almost) nothing
            vals = [v for v in vals if v > 0.25 * max(vals)]
        out[name] = (sum(vals) / len(vals), len(vals))
    return out


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(f) | set(w)):
        if not name.startswith("hg::"):
            continue
        fs, n = f.get(name, (0.0, 0))
        ws, _ = w.get(name, (0.0, 0))
        kernels[name] = {"FETCH_SIZE": fs, "WRITE_SIZE": ws, "launches_sampled": n, "traffic_bytes": fs + ws}
    doc = {
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 "
                   + (sys.argv[4] if len(sys.argv) > 4 else "bench.py --steps 3 --warmup 1") +
                   " --no-cpu-baseline; aggregated by scripts/pmc_traffic.py",
        "unit": "bytes per launch (counter value x 1024); FETCH_SIZE NOT doubled: the gfx950 x2 correction is "
                "calibrated only for wide 16-B/lane streams, these kernels issue 4/8-byte gathers "
                "(MI355X_MICROARCH.md HBM section: other widths uncalibrated); early-exit launches of the residual "
                "kernel (solver already terminated) are excluded",
        "csrc_sha16": source_digest(),
        "kernels": kernels,
    }
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    for k, v in kernels.items():
        print("%-44s %10.0f B fetch %10.0f B write (%d launches)" % (k[:44], v["FETCH_SIZE"], v["WRITE_SIZE"], v["launches_sampled"]))


if __name__ == "__main__":
    main()
