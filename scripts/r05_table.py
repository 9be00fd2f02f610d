"""Regenerates the round-5 results table of DESIGN.md section 5 from profiles/r05_bench_*.json (prints the rows)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def L(n):
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench_%s.json" % n)))


def f(x, nd=1):
    return ("%." + str(nd) + "f") % x


def gaps():
    txt = open(os.path.join(ROOT, "profiles", "r05_trace_gaps.txt")).read()
    return {k: float(re.search(r"%s\s+\S+ us/step\s+\S+ launches/step\s+(\S+) us/launch" % k, txt).group(1))
            for k in ("k_bin_count", "k_bin_offsets", "k_bin_scatter", "k_bin_apply")}


def rows():
    out = []
    d = L("default"); r = d["roofline"]; ins = r["families"]["insert(expand+sort+alloc+apply, 3 levels fused)"]; g = gaps()
    traffic = (" (PMC traffic %.1f MB of %.2f MB algorithmic)" % (r["traffic"] / 1e6, r["algorithmic_bytes_per_launch"] / 1e6)) if r.get("traffic") else ""
    out.append("| **headline** (`default`): register one 100k-pt scan, 3 levels | **%s scans/s** | %s ms | `k_tsdf_residuals_single<512>` %s µs, %d GB/s, %s %%%s; insert family %d µs incl. event pairs (kernels: count %s + offsets %s + scatter %s + apply %s µs, `r05_trace_gaps.txt`), %d GB/s, %s %% | %s scans/s (1) | 3526 |" % (
        f(d["value"]), f(d["ms_per_step"], 4), f(r["avg_launch_ms"] * 1e3, 1), r["achieved"], f(r["frac"] * 100, 1), traffic, round(ins["avg_launch_ms"] * 1e3),
        f(g["k_bin_count"]), f(g["k_bin_offsets"]), f(g["k_bin_scatter"]), f(g["k_bin_apply"]), ins["achieved"], f(ins["frac"] * 100, 1), f(d["cpu_baseline"]["value"], 2)))
    s = L("steps120")
    out.append("| `steps120` (the 120-position trajectory) | %s scans/s | %s ms | %s µs | — | 3697.4 |" % (f(s["value"]), f(s["ms_per_step"], 4), f(s["roofline"]["avg_launch_ms"] * 1e3, 1)))
    hi = d["host_inclusive"]
    out.append("| host-inclusive (scans in pageable host memory) | %s scans/s | %s ms | — | — | 3424.2 |" % (f(hi["value"]), f(hi["ms_per_step"], 4)))
    w = L("window"); r = w["roofline"]
    out.append("| `window` (10 control points, 9 × 100k blocks, swept leaving scan) | **%s scans/s** | %s ms | `k_window_residuals<false>` %s µs, %d GB/s, %s %%; `k_lm` %s µs per evaluating launch (rocprof average over all launches 19.3) | %s (1) | 1354 on the static leaving scan (8.6 instead of 10.0 iterations); 1208 on this workload before the solver work |" % (
        f(w["value"]), f(w["ms_per_step"], 4), f(r["avg_launch_ms"] * 1e3, 1), r["achieved"], f(r["frac"] * 100, 1), f(r["lm_avg_launch_ms"] * 1e3, 1), f(w["cpu_baseline"]["value"], 3)))
    c = L("window_cyclic_reduction"); r = c["roofline"]
    out.append("| `window_cyclic_reduction` (`HG_LM_BTD_CR=1`: the round-4 factorisation inside the round-5 step) | %s scans/s | %s ms | `k_lm` %s µs | — | — |" % (f(c["value"]), f(c["ms_per_step"], 4), f(r["lm_avg_launch_ms"] * 1e3, 1)))
    b = L("window_batch8"); r = b["roofline"]; cb = b["cpu_baseline"]
    out.append("| `window_batch8` | %s scans/s | %s ms | `k_window_residuals_jobs` %d µs, %d GB/s, %s %%; `k_lm_jobs` %s µs | %s (1) / %s (%d threads) | 2358.8 |" % (
        f(b["value"]), f(b["ms_per_step"], 3), round(r["avg_launch_ms"] * 1e3), r["achieved"], f(r["frac"] * 100, 1), f(r["lm_avg_launch_ms"] * 1e3, 1), f(cb["value"], 3), f(cb["all_cores"]["value"], 2), cb["all_cores"]["cores"]))

    def mb(n, label, r4, bold=False):
        m = L(n); r = m["roofline"]; cb = m.get("cpu_baseline")
        cpu = "%s (1) / %s (%d)" % (f(cb["value"], 2), f(cb["all_cores"]["value"], 1), cb["all_cores"]["cores"]) if cb else "—"
        v = "%s matches/s" % f(m["value"], 0)
        if bold:
            v = "**" + v + "**"
        return "| %s | %s | %s ms | %s µs, %d GB/s, %s %% | %s | %s |" % (label, v, f(m["ms_per_step"], 3), f(r["avg_launch_ms"] * 1e3, 1), r["achieved"], f(r["frac"] * 100, 1), cpu, r4)
    out.append(mb("match_batch64", "`match_batch64` (two batches in flight: `hg_problem_solve_batch_async`)", "28.1k, 27.6 %", True))
    out.append(mb("match_batch64_blocking", "`match_batch64_blocking` (`--batches-in-flight 1`: one blocking call per batch)", "—"))
    out.append(mb("match_batch64_nopartition", "`match_batch64_nopartition` (`HG_PARTITION_MIN=0`, two batches in flight)", "—"))
    out.append(mb("match_batch16", "`match_batch16` (two batches in flight)", "20.3k"))
    for n, r4 in (("register_batch8", "9.9k"), ("register_batch16", "12.0k")):
        m = L(n); r = m["roofline"]; cb = m["cpu_baseline"]
        out.append("| `%s` | %s scans/s | %s ms | %s µs, %d GB/s, %s %% | %s / %s | %s |" % (n, f(m["value"]), f(m["ms_per_step"], 3), f(r["avg_launch_ms"] * 1e3, 1), r["achieved"], f(r["frac"] * 100, 1), f(cb["value"], 2), f(cb["all_cores"]["value"], 1), r4))
    m = L("register_filtered")
    out.append("| `register_filtered` (AdaptiveVoxelFilter matching set) | %s scans/s | %s ms | filters + match + insert, no single dominant kernel | %s (1) | 2204 |" % (f(m["value"]), f(m["ms_per_step"], 3), f(m["cpu_baseline"]["value"], 1)))
    o = L("offline8x500"); g2 = L("offline8x500_gather"); r = o["roofline"]; cb = o["cpu_baseline"]
    out.append("| `offline8x500` (configs[3] on one GPU) | %s scans/s (with the final gather %s) | %s ms | `k_tsdf_residuals_single_batch` %s µs, %d GB/s, %s %% | %s / %s | 11.3k |" % (
        f(o["value"]), f(g2["value"]), f(o["ms_per_step"], 3), f(r["avg_launch_ms"] * 1e3, 1), r["achieved"], f(r["frac"] * 100, 1), f(cb["value"], 2), f(cb["all_cores"]["value"], 1)))
    for n, label, r4 in (("insert_stream32", "`insert_stream32` (exact, 32 scans per call)", "18.5k"), ("insert_stream32hbm", "`insert_stream32hbm` (scans from HBM tensors of another allocator)", "16.6k"),
                         ("insert_stream64hbm", "`insert_stream64hbm` (wall positions)", "9.1k"), ("insert_stream64warm", "`insert_stream64warm`", "9.7k"),
                         ("insert_stream500", "`insert_stream500` (400 room copies in one pool, `--max-blocks 1048576`)", "10.4k")):
        m = L(n); r = m["roofline"]; cb = m["cpu_baseline"]
        out.append("| %s | %s scans/s | %s ms | insert family %s µs per scan, %d GB/s, %s %% | %s / %s (%d threads) | %s |" % (
            label, f(m["value"]), f(m["ms_per_step"], 3), f(r["avg_launch_ms"] * 1e3, 1), r["achieved"], f(r["frac"] * 100, 1), f(cb["value"], 1), f(cb["all_cores"]["value"], 0), cb["all_cores"]["cores"], r4))
    m = L("insert_stream_fast")
    out.append("| `insert_stream_fast` (tolerance mode) | %s scans/s | %s ms | — | — | 21.9k |" % (f(m["value"]), f(m["ms_per_step"], 3)))
    return out


if __name__ == "__main__":
    print("\n".join(rows()))
