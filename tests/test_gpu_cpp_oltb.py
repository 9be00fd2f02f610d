"""The shipped C++ OptimizingLocalTrajectoryBuilder (cpp/hg_adapter.h: the reference's own window shape, device
solves and insertions through the C ABI) against an independent Python statement of
optimizing_local_trajectory_builder.cc over the CPU oracle (tests/oltb_replay.py).

cpp/example_oltb.cc feeds the builder a deterministic stream -- IMU 100 Hz, odometry 50 Hz, lidar 20 Hz whose scan
times do not fall on control points -- with the options of configuration_files/trajectory_builder_3d.lua (TSDF
grids), dumps every message bit for bit and prints the window after every step. The replay gets the same messages.
Compared, per step: whether a solve ran, its iterations and termination, the TSDF blocks of the solve (cloud size,
bracketing control points, interpolation factor, grid), the number of IMU / odometry blocks, every control point
(time exactly; pose within 1e-4 m / 1e-4 rad, the north-star tolerance; velocity) and the inserted range data; at
the end the voxels of every live submap, bit for bit."""
import os
import re
import struct
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "hectorgrapher_amd", "cpp")
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _build():
    exe = os.path.join(CPP, "example_oltb")
    src = os.path.join(CPP, "example_oltb.cc")
    hdr = os.path.join(CPP, "hg_adapter.h")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-std=c++11", "-O2", src, "-L" + os.path.join(ROOT, "hectorgrapher_amd"), "-lhg_mi355x",
                               "-Wl,-rpath," + os.path.join(ROOT, "hectorgrapher_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_cpp_oltb_reference_shape_against_oracle(tmp_path, mode):
    import oltb_replay as rp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    exe = _build()
    dump = str(tmp_path / "oltb.bin")
    scans = 44
    out = subprocess.run([exe, dump, str(mode), str(scans)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    raw = open(dump, "rb").read()
    r = rp.replay_and_compare(po, raw, out.stdout, mode, scans)
    b, solves, interpolated, single = r["b"], r["solves"], r["interpolated"], r["single"]
    max_dt, max_dr, max_dv, final_submaps = r["max_dt"], r["max_dr"], r["max_dv"], r["final_submaps"]
    assert solves >= (30 if mode == 1 else 18), solves   # CONSTANT: a control point (and a solve) every other scan
    print("mode %d: %d solves, %d interpolated / %d single blocks, max dt %.3g m dr %.3g rad dv %.3g m/s, cloud %.3g m" % (
        mode, solves, interpolated, single, max_dt, max_dr, max_dv, max(b.cloud_errors)))
    assert max_dt < 1e-4 and max_dr < 1e-4 and max_dv < 1e-3, (max_dt, max_dr, max_dv)
    if mode == 1:
        assert single > 20            # clouds on control points: single-pose blocks
    else:
        assert interpolated > 60 and single == 0   # no scan time falls on a control point
    # the range data the adapter inserted is the replay's own, to float rounding of a last-bit pose difference
    assert b.cloud_errors and max(b.cloud_errors) < 1e-5, max(b.cloud_errors)
    # the live submaps at the end: same local poses, same voxels in the same order
    # (mode 4 freezes the map at scan 30, when the motion filter has let four insertions through: one submap)
    assert len(final_submaps) == len(b.submaps) >= (1 if mode == 4 else 2), (len(final_submaps), len(b.submaps), b.num_insertions)
    for (pose, num, grids), sm in zip(final_submaps, b.submaps):
        assert num == sm.num_range_data
        np.testing.assert_allclose(pose, sm.local_pose, atol=1e-5)
        for (cells, tsd, weight), og in zip(grids, (sm.high, sm.low)):
            o_cells, o_tsd, o_weight = og.export()
            assert len(tsd) == len(o_tsd) and len(tsd) > 1000
            assert np.array_equal(cells, o_cells) and np.array_equal(tsd, o_tsd) and np.array_equal(weight, o_weight)
