"""Randomised and full-size parity sweeps (tests/tools/*.py) at sizes that finish in seconds."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
TOOLS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools")


def _run(script, *args):
    out = subprocess.run([sys.executable, os.path.join(TOOLS, script), *map(str, args)],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return out.stdout


def test_random_insert_and_solver_configurations():
    """60 random inserter configurations (resolutions, saturation at small maximum_weight, ratios,
    dense clusters -> giant voxels) bit-exact; 60 random windows (1-10 control points, constant /
    free first state, velocities, per-scan / interpolated / unwarped blocks, odometry and IMU
    blocks) within 1e-4 m / 1e-4 rad with identical iteration counts."""
    out = _run("stress_parity.py", 60, 60, 3)
    assert "total mismatches: 0" in out


def test_full_size_scans_bit_exact_along_the_bench_trajectory():
    """100k-point scans at light and heavy positions (about 3000 updates on one voxel) into the
    three grids: counters and every voxel code identical to the oracle."""
    out = _run("stress_fullsize.py", 20, 60, 58)
    assert out.count("exact=True") == 3
