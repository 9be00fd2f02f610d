#!/usr/bin/env python3
"""Instruction budget of the batched matcher's hot path, per phase of a return's evaluation.

Compiles csrc/hg_match.hip with -DHG_ISA_REGIONS -S (comment markers between the phases; only the 3-level doubling
variant of the direct lookup is instantiated, so the hot path is one run of instructions), takes
k_tsdf_residuals_single_batch<256> and counts the instructions between consecutive markers by class. Blocks that the
hot path only branches around (the general lookup, the single-level path) sit behind the last marker's region or in
cold blocks; instructions under a face-lane predicate are executed by every wavefront that has such a lane (nearly
all) and are counted.

Usage: python3 scripts/isa_budget.py [out.json]      (needs hipcc; no GPU)
"""
import json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN2hg29k_tsdf_residuals_single_batchILi256ELb0EEEvPKNS_9SingleJobE"


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_waitcnt") or op == "s_nop":
        return "wait/nop"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm", "s_barrier")):
        return "control"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    return None


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "m.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                               "-ffp-contract=off", "-DHG_ISA_REGIONS", "-S", "--cuda-device-only",
                               "-I" + os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "hectorgrapher_amd", "csrc", "hg_match.hip"), "-o", asm],
                              stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    start = next(i for i, ln in enumerate(lines) if ln.startswith(KERNEL + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    regions, order, cur = {}, [], "prologue"
    for ln in body:
        m = re.search(r"HG_REGION (\S+)\|(\S+)", ln)
        if m:
            cur = m.group(2)
            continue
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        c = classify(t.split()[0])
        if c is None:
            continue
        if cur not in regions:
            regions[cur] = {}
            order.append(cur)
        regions[cur][c] = regions[cur].get(c, 0) + 1
    meta = "\n".join(lines[end:end + 400])
    vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", "\n".join(lines[end:]))
    classes = ["valu", "salu", "smem", "vmem", "lds", "mfma", "wait/nop", "control"]
    print("%-22s" % "phase" + "".join("%9s" % c for c in classes))
    total = {c: 0 for c in classes}
    for r in order:
        print("%-22s" % r + "".join("%9d" % regions[r].get(c, 0) for c in classes))
        for c in classes:
            total[c] += regions[r].get(c, 0)
    print("%-22s" % "total (static)" + "".join("%9d" % total[c] for c in classes))
    res = {"kernel": "hg::k_tsdf_residuals_single_batch<256>, 3-level doubling variant, static instruction counts",
           "phases": {r: regions[r] for r in order}, "total": total, "vgprs": int(vg.group(1)) if vg else None,
           "note": "'exit' holds the cold blocks laid out behind the tile phase (general lookup call, single-level path)"}
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
