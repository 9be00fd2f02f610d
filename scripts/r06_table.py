#!/usr/bin/env python3
"""One line per r06 bench JSON under a directory (default gpurun_out/r06prof): value, per step, roofline, CPU figure, gate."""
import glob, json, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06prof"
for f in sorted(glob.glob(os.path.join(d, "r06_bench_*.json"))):
    name = os.path.basename(f)[10:-5]
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        continue
    if "value" not in j:
        continue
    r = j.get("roofline") or {}
    p = j.get("parity") or {}
    cb = j.get("cpu_baseline") or {}
    gate = p.get("bit_exact", p.get("tolerance_ok", p.get("max_dt_m")))
    ac = (cb.get("all_cores") or {})
    print("%-42s %9.1f %-9s %9.4f ms | frac %-7s launch %-8s GB/s %-7s | cpu %-7s all %-8s (%s) | gate %s" % (
        name, j["value"], j["unit"], j["ms_per_step"],
        round(r["frac"], 4) if r.get("frac") is not None else None,
        round(r["avg_launch_ms"] * 1e3, 1) if r.get("avg_launch_ms") else None,
        round(r["achieved"]) if r.get("achieved") else None,
        round(cb["value"], 2) if cb.get("value") else None,
        round(ac["value"], 1) if ac.get("value") else None, ac.get("cores"), gate))
    for k, v in (j.get("rows") or {}).items():
        print("    row %-14s %9.1f %s  cpu %.2f  gate %s" % (k, v["value"], v["unit"], v["cpu_baseline"]["value"], v["parity"]))
    if name == "default":
        for k, v in (j.get("secondary") or {}).items():
            print("    secondary %-22s %s" % (k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("value", "unit", "frac", "parity_ok", "wall_s", "error")}))
        hi = j.get("host_inclusive") or {}
        print("    host_inclusive", hi.get("value"), "| solve_form:", r.get("solve_form"), "| traffic", r.get("traffic"), r.get("traffic_source", "")[:60])
