"""Aggregates a rocprofv3 SQ-counter pass of `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline`
(scripts/r02_profile.sh) into per-kernel averages per launch: python scripts/pmc_sq.py <counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys
import os as _os
sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from hectorgrapher_amd._lib import source_digest

COMMAND = ("rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
           "SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline")


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sys.argv[1])):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for name, counters in agg.items():
        if not name.startswith("hg::"):
            continue
        d = {k: sum(v) / len(v) for k, v in counters.items()}
        d["launches"] = len(next(iter(counters.values())))
        waves = d.get("SQ_WAVES") or 1.0
        d["per_wave"] = {k: round(v / waves, 1) for k, v in d.items() if k.startswith("SQ_") and k != "SQ_WAVES"}
        out[name] = d
    json.dump({"command": COMMAND if len(sys.argv) < 4 else COMMAND.split(" -- ")[0] + " -- python3 " + sys.argv[3] + " --no-cpu-baseline",
               "unit": "counter totals per launch, averaged over the launches of the run (early-exit launches of "
                       "the residual kernel included); per_wave = total / SQ_WAVES",
               "csrc_sha16": source_digest(),
               "kernels": out}, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
