import json, sys
d = json.loads(sys.stdin.read())
r = d.get("roofline") or {}
lm = r.get("lm_avg_launch_ms")
print(sys.argv[1], round(d["value"], 1), "ms/step %.4f" % d["ms_per_step"], "GB/s", round(r.get("achieved", 0)), "launch_ms", r.get("avg_launch_ms"),
      *(("k_lm_us", round(lm * 1e3, 2)) if lm else ()))
