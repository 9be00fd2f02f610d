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


def _parse_stdout(text):
    steps, cur = [], None
    for line in text.splitlines():
        m = re.match(r"scan (\d+) time (-?\d+) result (\d) solved (\d) iterations (\d+) termination (\d+) (\d+) queued (\d+) "
                     r"imu_blocks (\d+) odometry_blocks (\d+) residuals (\d+)", line)
        if m:
            v = [int(x) for x in m.groups()]
            cur = {"scan": v[0], "time": v[1], "result": v[2], "solved": v[3], "it": v[4], "term": (v[5], v[6]), "queued": v[7],
                   "imu_blocks": v[8], "odometry_blocks": v[9], "residuals": v[10], "blocks": [], "cps": [], "local_pose": None,
                   "inserted": 0}
            steps.append(cur)
            continue
        m = re.match(r"\s+block (\d+) (-?\d+) (-?\d+) (\S+) (\d)", line)
        if m:
            cur["blocks"].append((int(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(4)), int(m.group(5))))
            continue
        m = re.match(r"\s+cp (-?\d+) pose (.*) vel (.*)", line)
        if m:
            cur["cps"].append((int(m.group(1)), np.array([float(x) for x in m.group(2).split()]),
                               np.array([float(x) for x in m.group(3).split()])))
            continue
        m = re.match(r"\s+local_pose (-?\d+) (.*) inserted (\d+) submaps (\d+)", line)
        if m:
            cur["local_pose"] = (int(m.group(1)), np.array([float(x) for x in m.group(2).split()]))
            cur["inserted"] = int(m.group(3))
            cur["submaps"] = int(m.group(4))
    return steps


def _messages(raw):
    off = 4
    while off < len(raw):
        (kind,) = struct.unpack_from("i", raw, off); off += 4
        if kind == 0:
            (t,) = struct.unpack_from("q", raw, off); off += 8
            w = np.frombuffer(raw, np.float64, 3, off).copy(); off += 24
            yield ("imu", t, w)
        elif kind == 1:
            (t,) = struct.unpack_from("q", raw, off); off += 8
            pose = np.frombuffer(raw, np.float64, 7, off).copy(); off += 56
            yield ("odom", t, pose)
        elif kind == 2:
            (t,) = struct.unpack_from("q", raw, off); off += 8
            origin = np.frombuffer(raw, np.float32, 3, off).copy(); off += 12
            (n,) = struct.unpack_from("i", raw, off); off += 4
            pts = np.frombuffer(raw, np.float32, n * 4, off).reshape(n, 4).copy(); off += 16 * n
            yield ("scan", t, origin, pts)
        elif kind == 3:
            (n,) = struct.unpack_from("i", raw, off); off += 4
            if n:
                origin = np.frombuffer(raw, np.float32, 3, off).copy(); off += 12
                ret = np.frombuffer(raw, np.float32, n * 3, off).reshape(n, 3).copy(); off += 12 * n
                yield ("inserted", origin, ret)
            else:
                yield ("inserted", None, None)
        elif kind == 4:
            pose = np.frombuffer(raw, np.float64, 7, off).copy(); off += 56
            (num,) = struct.unpack_from("i", raw, off); off += 4
            grids = []
            for _ in range(2):
                (n,) = struct.unpack_from("i", raw, off); off += 4
                cells = np.frombuffer(raw, np.int32, n * 3, off).reshape(n, 3).copy(); off += 12 * n
                tsd = np.frombuffer(raw, np.uint16, n, off).copy(); off += 2 * n
                weight = np.frombuffer(raw, np.uint16, n, off).copy(); off += 2 * n
                grids.append((cells, tsd, weight))
            yield ("submap", pose, num, grids)
        else:
            raise AssertionError("bad record kind %d" % kind)


MODES = {
    0: dict(),                                                      # the Lua defaults
    1: dict(control_point_sampling="SYNCED_WITH_RANGE_DATA"),
    2: dict(control_point_sampling="ADAPTIVE", use_multi_resolution_matching=True, sampling_max_delta_translation=0.03),
    3: dict(use_per_point_unwarping=True),
}


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_cpp_oltb_reference_shape_against_oracle(tmp_path, mode):
    import oltb_replay as rp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    exe = _build()
    dump = str(tmp_path / "oltb.bin")
    scans = 44
    out = subprocess.run([exe, dump, str(mode), str(scans)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    gpu = _parse_stdout(out.stdout)
    assert len(gpu) == scans
    raw = open(dump, "rb").read()
    assert struct.unpack_from("i", raw, 0)[0] == mode

    opt = rp.Options()
    opt.initialization_duration = 0.2
    opt.num_range_data = 4
    for k, v in MODES[mode].items():
        setattr(opt, k, v)
    b = rp.OracleOLTB(po, opt)

    msgs = list(_messages(raw))
    max_dt = max_dr = max_dv = 0.0
    solves = interpolated = single = 0
    k = 0
    i = 0
    final_submaps = []
    while i < len(msgs):
        m = msgs[i]
        if m[0] == "imu":
            b.add_imu(m[1], m[2])
        elif m[0] == "odom":
            b.add_odometry(m[1], m[2])
        elif m[0] == "submap":
            final_submaps.append(m[1:])
        elif m[0] == "scan":
            ins = msgs[i + 1]
            assert ins[0] == "inserted"
            i += 1
            b.forced_range_data = None if ins[1] is None else (ins[1], ins[2])
            before = b.num_optimizations
            res = b.add_range_data(m[1], m[2], m[3], 12)
            g = gpu[k]
            assert g["time"] == m[1]
            solved = b.num_optimizations - before
            assert g["solved"] == solved, (k, g["solved"], solved)
            assert g["result"] == (1 if res is not None else 0), k
            assert g["queued"] == len(b.clouds), (k, g["queued"], len(b.clouds))
            if solved:
                solves += 1
                so = b.last_summary
                assert (so.num_iterations, so.termination_type, so.termination_reason) == (g["it"], g["term"][0], g["term"][1]), k
                assert g["imu_blocks"] == b.last_imu_blocks and g["odometry_blocks"] == b.last_odometry_blocks, k
                assert g["residuals"] == sum(x[0] for x in b.last_blocks) + 9 * b.last_imu_blocks + 6 * b.last_odometry_blocks, k
                if mode == 3:   # (the device merges the subdivisions between two control points into one block)
                    interpolated += len(b.last_blocks)
                else:
                    assert len(g["blocks"]) == len(b.last_blocks), k
                for gb, ob in zip(g["blocks"] if mode != 3 else [], b.last_blocks):
                    assert gb[:3] == tuple(ob[:3]) and gb[4] == ob[4], (k, gb, ob)
                    assert gb[3] == ob[3], (k, gb, ob)       # the interpolation factor: same ticks, same division
                    if gb[2] >= 0:
                        interpolated += 1
                        assert 0.0 < gb[3] < 1.0
                    else:
                        single += 1
            if res is not None:
                assert g["local_pose"][0] == res["time"]
                assert g["inserted"] == (1 if res["inserted"] else 0), k
                assert (ins[1] is not None) == res["inserted"]
            # the window as it stands after the step
            assert [c[0] for c in g["cps"]] == [c["time"] for c in b.cps], k
            for (t, gp, gv), c in zip(g["cps"], b.cps):
                max_dt = max(max_dt, float(np.linalg.norm(gp[:3] - c["t"])))
                max_dr = max(max_dr, float(2.0 * np.arccos(min(1.0, abs(float(gp[3:] @ c["q"]))))))
                max_dv = max(max_dv, float(np.linalg.norm(gv - c["v"])))
            k += 1
        i += 1
    assert k == scans
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
    assert len(final_submaps) == len(b.submaps) >= 2
    for (pose, num, grids), sm in zip(final_submaps, b.submaps):
        assert num == sm.num_range_data
        np.testing.assert_allclose(pose, sm.local_pose, atol=1e-5)
        for (cells, tsd, weight), og in zip(grids, (sm.high, sm.low)):
            o_cells, o_tsd, o_weight = og.export()
            assert len(tsd) == len(o_tsd) and len(tsd) > 1000
            assert np.array_equal(cells, o_cells) and np.array_equal(tsd, o_tsd) and np.array_equal(weight, o_weight)
