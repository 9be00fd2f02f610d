"""The twisted block-tridiagonal Cholesky of the LM step (csrc/hg_btd.h) outside the solver: scripts/tw_bench.hip builds
the function into a kernel of its own, solves a random SPD block-tridiagonal band system of 2 .. 9 groups of 9 columns
(what a window of up to ten control points with velocities gives) and compares with a dense host Cholesky; the
program exits non-zero above 1e-9 relative. Window lengths that make the two chains uneven (even group counts), the
shortest chains (2, 3) and the full nine."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    exe = os.path.join(ROOT, "scripts", "tw_bench")
    src = os.path.join(ROOT, "scripts", "tw_bench.hip")
    hdr = os.path.join(ROOT, "hectorgrapher_amd", "csrc", "hg_btd.h")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                               "-I" + os.path.join(ROOT, "hectorgrapher_amd", "csrc"), src, "-o", exe])
    return exe


@pytest.mark.parametrize("groups", [2, 3, 4, 5, 6, 7, 8, 9])
def test_twisted_factorisation_against_dense_cholesky(groups):
    exe = _build()
    out = subprocess.run([exe, str(groups), "20"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ok 1" in out.stdout
