"""The C-ABI library builds, loads and exports every symbol include/hg_mi355x.h declares.
No compute calls here (no GPU needed)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "hg_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from hectorgrapher_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = header_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SYMBOLS) == names


def test_library_exports_nothing_but_the_c_abi():
    """-fvisibility=hidden + csrc/hg_exports.map: the dynamic symbol table holds the header's entry points and no
    hg:: internal, kernel handle or hg_*_big twin of the second solver build."""
    import subprocess
    from hectorgrapher_amd import _lib
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == header_functions()
    listed = re.findall(r"^\s+(hg_[a-z0-9_]+);", open(os.path.join(ROOT, "hectorgrapher_amd", "csrc",
                                                                    "hg_exports.map")).read(), flags=re.M)
    assert sorted(listed) == header_functions()


def test_two_solver_builds_define_no_common_symbol():
    """hg_match.hip is compiled twice (hg_match.o and -DHG_BIG hg_match_big.o) behind the rename list
    csrc/hg_match_big_names.h: a global the list misses would be defined by both objects (a silent ODR clash,
    weak symbols of templates and inline functions aside)."""
    import subprocess
    csrc = os.path.join(ROOT, "hectorgrapher_amd", "csrc")
    defined = []
    for obj in ("hg_match.o", "hg_match_big.o"):
        path = os.path.join(csrc, obj)
        assert os.path.exists(path), "run __graft_entry__.build() first"
        out = subprocess.check_output(["nm", "--defined-only", "-g", path], text=True)
        # strong definitions only: T/D/B/R (W/V/u = weak or unique template / inline instantiations the linker folds)
        defined.append({l.split()[-1] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] in "TDBR"})
    common = sorted(defined[0] & defined[1])
    assert not common, common[:10]
    assert len(defined[0]) > 20 and len(defined[1]) > 20
    # weak definitions the linker folds into one must be the same code in both builds: the standard library and the
    # inline helpers of hg_internal.h (which does not depend on HG_BIG), nothing that hg_match.hip itself defines
    weak = []
    for obj in ("hg_match.o", "hg_match_big.o"):
        out = subprocess.check_output(["nm", "--defined-only", "-g", os.path.join(csrc, obj)], text=True)
        weak.append({l.split()[-1] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] in "WVu"})
    shared_header = open(os.path.join(csrc, "hg_internal.h")).read()
    for sym in sorted(weak[0] & weak[1]):
        if sym.startswith(("_ZNSt", "_ZSt", "_ZNKSt", "DW.ref.", "__clang_")):
            continue
        m = re.match(r"_ZN2hg(\d+)", sym)
        assert m, sym
        name = sym[len(m.group(0)):][:int(m.group(1))]
        assert re.search(r"\b(struct|class)\s+%s\b" % name, shared_header), sym


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the product path fails loudly (HG_ERR_NO_DEVICE), it never computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hectorgrapher_amd import api
    with pytest.raises(api.HgError):
        api.Context(0)


def test_product_package_does_not_import_oracle():
    pkg = os.path.join(ROOT, "hectorgrapher_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cc")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "pyoracle" not in src and "hg_oracle" not in src, f


def test_structs_match_header_layout():
    import ctypes
    from hectorgrapher_amd import _lib
    assert ctypes.sizeof(_lib.InsertOpts) == 80
    assert ctypes.sizeof(_lib.InsertStats) == 32
    assert ctypes.sizeof(_lib.SolverOpts) == 80
    assert ctypes.sizeof(_lib.SolverSummary) == 56


def test_null_arguments_are_refused_before_any_device_work():
    """Every entry point validates its pointers first and returns HG_ERR_INVALID (-1): callable
    without a GPU, and it never aborts like the reference's CHECKs."""
    import ctypes as C
    from hectorgrapher_amd import _lib
    L = _lib.load()
    n = C.c_size_t()
    w = C.c_int32()
    assert L.hg_grid_xray(None, None, None, 0, C.byref(w), C.byref(w), None, C.byref(n)) == -1
    assert L.hg_problem_solve_batch(None, 0, None, None) == -1
    assert L.hg_problem_solve(None, None, None) == -1
    assert L.hg_grid_insert(None, None, None, None, 0, 0, None, 0, 0, None) == -1
    assert L.hg_grid_export(None, None, None, None, 0, C.byref(n)) == -1
    assert L.hg_grid_to_proto(None, None, 0, C.byref(n)) == -1
    assert L.hg_voxel_filter(None, 0.1, None, 0, 3, 0, None, C.byref(n)) == -1
    assert L.hg_adaptive_voxel_filter(None, 2.0, 150.0, 15.0, None, 0, 3, 0, None, C.byref(n)) == -1


@pytest.mark.gpu
def test_handles_of_a_destroyed_context_are_refused():
    """hg_ctx_destroy orphans the grids and problems that outlive it (a garbage-collected host destroys in any
    order): every entry point then returns HG_ERR_INVALID instead of dereferencing the dead context, and
    destroying the orphans still frees them."""
    import ctypes as C
    from hectorgrapher_amd import _lib
    L = _lib.load()
    ctx, grid, prob = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert L.hg_ctx_create(0, None, C.byref(ctx)) == 0
    assert L.hg_grid_create(ctx, 0.1, 2.5, 1000.0, 1 << 10, C.byref(grid)) == 0
    assert L.hg_problem_create(ctx, C.byref(prob)) == 0
    assert L.hg_ctx_destroy(ctx) == 0
    n = C.c_uint32()
    st = _lib.InsertStats()
    assert L.hg_grid_clear(grid) == -1
    assert L.hg_grid_num_blocks(grid, C.byref(n)) == -1
    assert L.hg_grid_status(grid, C.byref(st)) == -1
    cnt = C.c_size_t()
    assert L.hg_grid_count(grid, C.byref(cnt)) == -1
    assert L.hg_problem_reset(prob) == -1
    assert L.hg_problem_solve_async(prob, None) == -1
    assert L.hg_problem_evaluate(prob, None, None, None, None) == -1
    assert b"context" in L.hg_last_error()
    assert L.hg_problem_destroy(prob) == 0
    assert L.hg_grid_destroy(grid) == 0


def test_context_option_keys_are_documented_and_the_library_reads_no_environment_behind_them():
    """Round 6: the library's switches are per-context options (hg_ctx_set_option); the keys the
    library accepts (csrc/hg_internal.h kOptDesc) are exactly the keys include/hg_mi355x.h documents, and the only
    getenv left in the sources is the one that builds a context's defaults plus HG_STREAM_PRIORITY (read before a
    context exists)."""
    csrc = os.path.join(ROOT, "hectorgrapher_amd", "csrc")
    internal = open(os.path.join(csrc, "hg_internal.h")).read()
    table = internal[internal.index("constexpr OptDesc kOptDesc"):]
    table = table[:table.index("};")]
    accepted = re.findall(r'\{"([a-z_]+)",\s*-?\d+\}', table)
    enum = internal[internal.index("enum Opt {"):]
    enum = enum[:enum.index("OPT_COUNT")]
    assert len(accepted) == len(re.findall(r"OPT_[A-Z_]+", enum)) and len(set(accepted)) == len(accepted)
    header = open(os.path.join(ROOT, "include", "hg_mi355x.h")).read()
    doc = header[header.index("Tuning and diagnostic switches of a context"):header.index("int hg_ctx_set_option")]
    documented = set()
    for line in doc.splitlines():
        m = re.match(r"\s*\*\s{3}([a-z_]+(?:\s*/\s*[a-z_]+)*(?:,\s*[a-z_]+)*)\s{2,}", line)
        if m:
            documented.update(k.strip() for k in re.split(r"[/,]", m.group(1)))
    assert documented == set(accepted), (sorted(documented ^ set(accepted)))
    # the order of the enum is the order of the table
    assert [("OPT_" + k.upper()) for k in accepted] == re.findall(r"OPT_[A-Z_]+", enum)
    getenvs = []
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            for i, line in enumerate(open(os.path.join(csrc, name)).read().splitlines(), 1):
                code = line.split("//")[0]
                if "getenv(" in code:
                    getenvs.append((name, i, line.strip()))
    assert len(getenvs) <= 3, getenvs
    assert all(n == "hg_grid.hip" for n, _, _ in getenvs), getenvs
