"""The header-only C++ host layer (hectorgrapher_amd/cpp/hg_adapter.h) over the C ABI: the example
drives a LocalTrajectoryBuilder3D-shaped builder (AddRangeData per scan: match + insert)."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "hectorgrapher_amd", "cpp")


def test_cpp_local_trajectory_builder_example():
    exe = os.path.join(CPP, "example_local_slam")
    if not os.path.exists(exe):   # built by __graft_entry__.build(); rebuild if the tree is fresh
        subprocess.check_call(["g++", "-std=c++11", "-O2", os.path.join(CPP, "example_local_slam.cc"),
                               "-L" + os.path.join(ROOT, "hectorgrapher_amd"), "-lhg_mi355x",
                               "-Wl,-rpath," + os.path.join(ROOT, "hectorgrapher_amd"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    poses = [tuple(float(v) for v in m.groups())
             for m in re.finditer(r"^scan \d+ pose (\S+) (\S+) (\S+)", out.stdout, re.M)]
    assert len(poses) == 5
    # the sensor moves 5 cm per scan along x; the matched poses follow it
    for k, (x, y, z) in enumerate(poses):
        assert abs(x - 0.05 * k) < 0.02 and abs(y) < 0.02 and abs(z) < 0.03
    # sliding-window builder (OptimizingLocalTrajectoryBuilder shape): odometry with an alternating
    # +-1 cm error (dead reckoning is off by up to 2 cm) and TSDF blocks in a window of 3 control
    # points; the solved poses stay between the two sources
    wposes = [tuple(float(v) for v in m.groups())
              for m in re.finditer(r"^window scan \d+ pose (\S+) (\S+) (\S+)", out.stdout, re.M)]
    assert len(wposes) == 8
    for k, (x, y, z) in enumerate(wposes):   # at rest for scans 0 and 1, then 5 cm per scan
        assert abs(x - 0.05 * max(0, k - 1)) < 0.021 and abs(y) < 0.02 and abs(z) < 0.02
    assert "(window 3," in out.stdout
    # the window carries IMU pre-integration blocks and velocity states (oltb.cc:928-1000): one block
    # per neighbouring pair of control points, and the solved velocity follows the 0.5 m/s motion
    blocks = [int(m.group(1)) for m in re.finditer(r"imu blocks (\d+)", out.stdout)]
    assert blocks[0] == 0 and blocks[1] == 1 and max(blocks) == 3 and blocks[-1] == 3
    vel = [tuple(float(v) for v in m.groups()) for m in re.finditer(r"v (\S+) (\S+) (\S+)\)", out.stdout)]
    assert len(vel) == 8
    assert max(abs(c) for c in vel[1]) < 0.25          # at rest
    for vx, vy, vz in vel[3:]:
        assert abs(vx - 0.5) < 0.3 and abs(vy) < 0.2 and abs(vz) < 0.2
    dq = re.search(r"^imu delta rotation (\S+) (\S+) (\S+) (\S+)", out.stdout, re.M)
    w, x, y, z = (float(dq.group(i)) for i in range(1, 5))
    assert abs(w - 0.70710678) < 1e-5 and abs(z - 0.70710678) < 1e-5 and abs(x) < 1e-12 and abs(y) < 1e-12
    # ActiveSubmaps3D bookkeeping (submap_3d.cc:492-514, 548-559) with num_range_data = 3
    lines = re.findall(r"^submaps after insert (\d+):(.*)$", out.stdout, re.M)
    assert [l[1].split() for l in lines] == [["1"], ["2"], ["3"], ["4", "1"], ["5", "2"],
                                             ["6(finished)", "3"], ["4", "1"]]
    # AddToTexture: the x-ray view of the first submap (ring of radius 3 m at 0.10 m: ~60 x 60 pixels)
    tex = re.search(r"^texture (\d+) x (\d+), (\d+) pixels with alpha", out.stdout, re.M)
    assert tex, out.stdout
    assert 55 <= int(tex.group(1)) <= 70 and 55 <= int(tex.group(2)) <= 70 and int(tex.group(3)) > 100
    # sensor::VoxelFilter / AdaptiveVoxelFilter wrappers: 61200 points -> 0.15 m voxels -> >= 150 points
    flt = re.search(r"^filters: (\d+) -> (\d+) -> (\d+) points", out.stdout, re.M)
    assert flt, out.stdout
    n0, n1, n2 = (int(flt.group(i)) for i in (1, 2, 3))
    assert n0 == 61200 and 300 < n1 < 2000 and 150 <= n2 <= n1


def _build_gather():
    exe = os.path.join(CPP, "example_gather")
    if not os.path.exists(exe):
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-D__HIP_PLATFORM_AMD__", "-DHG_WITH_RCCL",
                               "-I/opt/rocm/include", os.path.join(CPP, "example_gather.cc"),
                               "-L" + os.path.join(ROOT, "hectorgrapher_amd"), "-lhg_mi355x", "-L/opt/rocm/lib",
                               "-lamdhip64", "-lrccl", "-Wl,-rpath," + os.path.join(ROOT, "hectorgrapher_amd"),
                               "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_cpp_gather_two_ranks_on_one_gpu():
    """hg_gather.h (the multi-GPU exchange step for a C++ host): two rank processes on this GPU map
    their own submaps, rank 0 gathers both pyramids through the pipe transport, imports them into fresh
    grids and finds their exports equal to the owners' (count + hash of cells and codes)."""
    out = subprocess.run([_build_gather(), "pipe", "2"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"^gather ok: ranks 2 levels 3 blocks (\d+) voxels (\d+)", out.stdout, re.M)
    assert m, out.stdout
    assert int(m.group(1)) > 500 and int(m.group(2)) > 50000
    per_rank = re.findall(r"^rank (\d): (\d+) (\d+) (\d+) blocks", out.stdout, re.M)
    assert [r[0] for r in per_rank] == ["0", "1"]
    assert all(int(r[1]) > int(r[2]) > int(r[3]) > 0 for r in per_rank)   # finer levels hold more blocks


def test_cpp_gather_rccl_transport_single_rank():
    """The RCCL transport of the same example with a one-rank communicator: ncclCommInitRank,
    all-gather of the counts and the import / export check of rank 0's own pyramid run through RCCL
    (peers need more GPUs than this box has; the driver's multi-GPU node runs those)."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([_build_gather(), "rccl"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert re.search(r"^gather ok: ranks 1 levels 3", out.stdout, re.M), out.stdout


def test_cpp_gather_rccl_transport_two_ranks():
    """RcclTransport with a real peer: two rank processes, one GPU each (ncclSend / ncclRecv of the block
    payload over xGMI, counts and digests by ncclAllGather). Needs two GPUs: skipped on a one-GPU box (the
    driver's multi-GPU node runs it)."""
    import tempfile
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    exe = _build_gather()
    with tempfile.TemporaryDirectory() as d:
        id_file = os.path.join(d, "nccl_id")
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), HG_NCCL_ID_FILE=id_file,
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([exe, "rccl"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
        outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert re.search(r"^gather ok: ranks 2 levels 3 blocks (\d+) voxels (\d+)", outs[0][0], re.M), outs[0][0]


def _pose_mul(a, b):
    """Rigid3d operator* with the normalisation of rigid_transform.h:184-190 (as hg_adapter.h transform::Multiply)."""
    from hectorgrapher_amd import synth
    import numpy as np
    t = synth.quat_rotate(np.asarray(a[3:], np.float64), np.asarray(b[:3], np.float64)) + a[:3]
    q = synth.quat_mul(a[3:], b[3:])
    return np.concatenate([t, q / np.sqrt(q @ q)])


def _pose_inv(a):
    import numpy as np
    from hectorgrapher_amd import synth
    qc = np.array([a[3], -a[4], -a[5], -a[6]])
    return np.concatenate([synth.quat_rotate(qc, -np.asarray(a[:3], np.float64)), qc])


def _imu_delta_rotation(imu, start, end):
    """Rotation recurrence of IntegrateImuWithTranslationEuler (imu_integration.h:99-131): piecewise-constant
    angular velocity, delta *= AngleAxisVectorToRotationQuaternion(w dt) (transform/transform.h:121-135)."""
    import numpy as np
    q = np.array([1.0, 0.0, 0.0, 0.0])
    if not imu or not (start < end):
        return q
    it = 0
    while it + 1 < len(imu) and imu[it + 1][0] <= start:
        it += 1
    cur = start
    while cur < end:
        nxt_imu = imu[it + 1][0] if it + 1 < len(imu) else (1 << 62)
        nxt = min(nxt_imu, end)
        a = np.asarray(imu[it][1]) * ((nxt - cur) / 1e7)   # common::ToSeconds of a tick difference
        sq = float(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])
        scale, w = 0.5, 1.0
        if sq > 1e-8:
            norm = np.sqrt(sq)
            scale, w = np.sin(norm / 2.0) / norm, np.cos(norm / 2.0)
        x, y, z = scale * a
        q = np.array([q[0] * w - q[1] * x - q[2] * y - q[3] * z, q[0] * x + q[1] * w + q[2] * z - q[3] * y,
                      q[0] * y + q[2] * w + q[3] * x - q[1] * z, q[0] * z + q[3] * w + q[1] * y - q[2] * x])
        cur = nxt
        if cur == nxt_imu:
            it += 1
    return q


def test_cpp_window_builder_against_oracle(tmp_path):
    """The simplified sliding-window builder of the adapter (cpp/hg_adapter.h: SlidingWindowTrajectoryBuilder, one
    control point per scan; the reference's own shape is tested in tests/test_gpu_cpp_oltb.py) against the CPU oracle: the
    example dumps every input it feeds the adapter; this test states the window wiring a second time --
    range crop (oltb.cc:214-227), prediction from the odometry delta, first control point constant with its
    velocity (:1268-1275), IMU pre-integration blocks between neighbours (:928-1000), one multi-resolution scan
    block per free control point (:343-364), odometry blocks with delta = inverse(next) * previous (:1025-1029),
    insertion of the scans that leave the window -- over pyoracle, replays the same inputs and compares every
    control point of every step: 1e-4 m / 1e-4 rad, same iterations and termination."""
    import struct
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    exe = os.path.join(CPP, "example_parity")
    if not os.path.exists(exe):
        subprocess.check_call(["g++", "-std=c++11", "-O2", os.path.join(CPP, "example_parity.cc"),
                               "-L" + os.path.join(ROOT, "hectorgrapher_amd"), "-lhg_mi355x",
                               "-Wl,-rpath," + os.path.join(ROOT, "hectorgrapher_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    dump = str(tmp_path / "inputs.bin")
    out = subprocess.run([exe, dump, "9"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    # what the adapter reported
    gpu = {}
    step = None
    for line in out.stdout.splitlines():
        m = re.match(r"step (\d+) solved (\d+) iterations (\d+) termination (\d+) (\d+) window (\d+)", line)
        if m:
            step = int(m.group(1))
            gpu[step] = {"solved": int(m.group(2)), "it": int(m.group(3)), "term": (int(m.group(4)), int(m.group(5))), "cps": []}
            continue
        m = re.match(r"\s+inserted at (.*)", line)
        if m:
            gpu[step]["inserted_at"] = np.array([float(v) for v in m.group(1).split()])
            continue
        m = re.match(r"\s+cp \d+ pose (.*) vel (.*)", line)
        if m:
            gpu[step]["cps"].append((np.array([float(v) for v in m.group(1).split()]), np.array([float(v) for v in m.group(2).split()])))
    assert len(gpu) == 9
    # the inputs
    raw = open(dump, "rb").read()
    off = 0
    (n_scans,) = struct.unpack_from("i", raw, off)
    off += 4
    steps = []
    for _ in range(n_scans):
        (t,) = struct.unpack_from("q", raw, off); off += 8   # common::Time ticks
        (n,) = struct.unpack_from("i", raw, off); off += 4
        pts = np.frombuffer(raw, np.float32, n * 4, off).reshape(n, 4).copy(); off += 16 * n
        odom = np.frombuffer(raw, np.float64, 7, off).copy(); off += 56
        (n_imu,) = struct.unpack_from("i", raw, off); off += 4
        imu = []
        for _ in range(n_imu):
            (ti,) = struct.unpack_from("q", raw, off); off += 8
            vals = np.frombuffer(raw, np.float64, 3, off).copy(); off += 24
            imu.append((int(ti), vals))
        steps.append((t, pts, odom, imu))
    # the window builder over the oracle (options of example_parity.cc)
    res = [0.10, 0.20]
    window_max, w_imu, w_odom = 4, (1.0, 0.05, 2.0), (3.0, 5.0)
    grids = [po.Grid(r) for r in res]
    window, imu_all, map_has_data = [], [], False
    max_dt = max_dr = 0.0
    for k, (t, pts, odom, imu) in enumerate(steps):
        imu_all.extend(imu)
        r = np.sqrt((pts[:, 0] * pts[:, 0] + pts[:, 1] * pts[:, 1] + pts[:, 2] * pts[:, 2]).astype(np.float32))
        cloud = np.ascontiguousarray(pts[(r >= np.float32(1.0)) & (r <= np.float32(60.0)), :3])
        cp = {"time": t, "cloud": cloud, "odom": odom, "vel": np.zeros(3), "inserted": False}
        if not window:
            cp["pose"] = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)
        else:
            cp["pose"] = _pose_mul(window[-1]["pose"], _pose_mul(_pose_inv(window[-1]["odom"]), odom))
            if t > window[-1]["time"]:
                cp["vel"] = (cp["pose"][:3] - window[-1]["pose"][:3]) / ((t - window[-1]["time"]) / 1e7)
        window.append(cp)
        solved = 0
        if map_has_data:
            pr = po.Problem()
            for i, c in enumerate(window):
                first = i == 0 and len(window) > 1
                pr.add_pose(c["pose"], first)
                pr.set_velocity(i, c["vel"], first)
            for i in range(1, len(window)):
                dq = _imu_delta_rotation(imu_all, window[i - 1]["time"], window[i]["time"])
                pr.add_imu_block(i - 1, i, w_imu[0], w_imu[1], w_imu[2], (window[i]["time"] - window[i - 1]["time"]) / 1e7, dq)
            for i in range(1 if len(window) > 1 else 0, len(window)):
                c = window[i]["cloud"]
                pr.add_block(c, grids, 1.0 / np.sqrt(float(len(c))), i, multi_res=True)
                if i > 0:
                    delta = _pose_mul(_pose_inv(window[i]["odom"]), window[i - 1]["odom"])
                    pr.add_odometry_block(i - 1, i, w_odom[0], w_odom[1], delta)
            so = pr.solve()
            solved = 1
            for i, c in enumerate(window):
                c["pose"], c["vel"] = pr.get_pose(i), pr.get_velocity(i)
            while len(imu_all) > 1 and imu_all[1][0] <= window[0]["time"]:
                imu_all.pop(0)
            g = gpu[k]
            assert g["solved"] == 1
            assert (so.num_iterations, so.termination_type, so.termination_reason) == (g["it"], g["term"][0], g["term"][1]), k
        # the scans that leave the window are inserted at the poses the ADAPTER used (float casts of its doubles),
        # so that a last-bit difference of a pose cannot move a voxel and fork the two runs
        while window and (not map_has_data or len(window) > window_max):
            outcp = window[0]
            if not outcp["inserted"]:
                at = gpu[k]["inserted_at"]  # within 1e-4 of the oracle's own pose (asserted below), bit-identical cast
                assert np.linalg.norm(at[:3] - outcp["pose"][:3]) < 1e-4
                loc = po.transform_points(np.asarray(at, np.float64).astype(np.float32), outcp["cloud"])
                origin = po.transform_points(np.asarray(at, np.float64).astype(np.float32), np.zeros((1, 3), np.float32))[0]
                for gr in grids:
                    gr.insert(origin, loc)
                outcp["inserted"] = True
            if not map_has_data:
                map_has_data = True
                break
            window.pop(0)
        # compare the window as it stands after the step
        g = gpu[k]
        assert len(g["cps"]) == len(window), (k, len(g["cps"]), len(window))
        for (gp, gv), c in zip(g["cps"], window):
            max_dt = max(max_dt, float(np.linalg.norm(gp[:3] - c["pose"][:3])))
            max_dr = max(max_dr, float(2.0 * np.arccos(min(1.0, abs(float(gp[3:] @ c["pose"][3:]))))))
            np.testing.assert_allclose(gv, c["vel"], atol=1e-5)
    assert max_dt < 1e-4 and max_dr < 1e-4, (max_dt, max_dr)
    assert sum(g["solved"] for g in gpu.values()) == 8


def test_cpp_insert_unwarped_against_oracle(tmp_path):
    """mapping::InsertUnwarped of cpp/hg_adapter.h (the clouds that leave the window, unwarped return by return
    with the window's control poses and inserted: oltb.cc:1331-1379, :1437-1440, submap_3d.cc:436-437): the
    example dumps its inputs and the voxels of both grids; the oracle replays the inputs -- times as
    common::FromSeconds makes them, NaN returns kept, origin from the first return -- and every cell, code and
    the export order must be the same."""
    import struct
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    exe = os.path.join(CPP, "example_unwarp")
    if not os.path.exists(exe):
        subprocess.check_call(["g++", "-std=c++11", "-O2", os.path.join(CPP, "example_unwarp.cc"),
                               "-L" + os.path.join(ROOT, "hectorgrapher_amd"), "-lhg_mi355x",
                               "-Wl,-rpath," + os.path.join(ROOT, "hectorgrapher_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    dump = str(tmp_path / "unwarp.bin")
    out = subprocess.run([exe, dump], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    raw = open(dump, "rb").read()
    off = 0
    (n_cp,) = struct.unpack_from("i", raw, off); off += 4
    times, poses = [], []
    for _ in range(n_cp):
        (t,) = struct.unpack_from("q", raw, off); off += 8   # common::Time ticks
        poses.append(np.frombuffer(raw, np.float64, 7, off).copy()); off += 56
        times.append(int(t))
    submap = np.frombuffer(raw, np.float32, 7, off).copy(); off += 28
    (n_clouds,) = struct.unpack_from("i", raw, off); off += 4
    clouds = []
    for _ in range(n_clouds):
        (t,) = struct.unpack_from("q", raw, off); off += 8
        origin = np.frombuffer(raw, np.float32, 3, off).copy(); off += 12
        (n,) = struct.unpack_from("i", raw, off); off += 4
        pts = np.frombuffer(raw, np.float32, n * 4, off).reshape(n, 4).copy(); off += 16 * n
        clouds.append((int(t), origin, pts))
    assert any(np.isnan(c[2][:, 1]).any() for c in clouds)
    poses = np.asarray(poses)
    xyz, origin, ok = po.unwarp_range_data(np.asarray(times, np.int64), poses, clouds)
    assert ok
    opt = poses[0].astype(np.float32)
    xyz, origin = po.transform_points(opt, xyz), po.transform_points(opt, origin[None])[0]
    xyz, origin = po.transform_points(submap, xyz), po.transform_points(submap, origin[None])[0]
    for res in (0.10, 0.20):
        (n,) = struct.unpack_from("i", raw, off); off += 4
        cells = np.frombuffer(raw, np.int32, n * 3, off).reshape(n, 3); off += 12 * n
        tsd = np.frombuffer(raw, np.uint16, n, off); off += 2 * n
        weight = np.frombuffer(raw, np.uint16, n, off); off += 2 * n
        og = po.Grid(res)
        og.insert(origin, xyz, width=16)
        o_cells, o_tsd, o_weight = og.export()
        assert n == len(o_tsd) and n > 1000
        assert np.array_equal(cells, o_cells) and np.array_equal(tsd, o_tsd) and np.array_equal(weight, o_weight)
    assert off == len(raw)

