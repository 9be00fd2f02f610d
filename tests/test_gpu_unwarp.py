"""GPU parity of the per-point unwarping before insertion (second half;
optimizing_local_trajectory_builder.cc:1331-1379): hg_unwarp_range_data / hg_pyramid_insert_unwarped /
hg_register_scan_unwarped against the oracle's restatement of the reference loop."""
import numpy as np
import pytest

from hectorgrapher_amd import synth

pytestmark = pytest.mark.gpu

TICKS = 10_000_000  # per second (common::Time, 100 ns)


def control_points(n_cp, dt_s=0.05, t0=123_456_789_000):
    """A smooth curved motion: control point k at t0 + k dt, poses with a few cm / mrad between them."""
    times = np.array([t0 + int(round(k * dt_s * TICKS)) for k in range(n_cp)], np.int64)
    poses = []
    for k in range(n_cp):
        q = synth.quat_mul(synth.quat_from_axis_angle([0, 0, 1], 0.02 * k),
                           synth.quat_from_axis_angle([1, -1, 2], 0.004 * k * k))
        poses.append(np.concatenate([[0.06 * k, 0.03 * k - 0.002 * k * k, 0.005 * k], q / np.linalg.norm(q)]))
    return times, np.asarray(poses, np.float64)


def timed_scan(rings, cols, pose, stream, sweep_s, nan_every=0):
    """A structured scan whose columns are stamped over `sweep_s` seconds (spinning lidar): points n x 4."""
    pts = synth.generate_scan(pose, rings, cols, stream=stream)
    t = np.repeat(np.linspace(0.0, sweep_s, cols, dtype=np.float32), rings)
    out = np.concatenate([pts, t[:, None]], 1).astype(np.float32)
    if nan_every:
        out[::nan_every, 1] = np.nan
    return out


def oracle_chain(po, clouds, times, poses, pose_tq):
    """(xyz, origin) in the grid frame as the reference produces them: unwarp (:1331-1379), then
    optimized_pose.cast<float>() (:1437-1440), then the submap frame change (submap_3d.cc:436-437)."""
    xyz, origin, ok = po.unwarp_range_data(times, poses, clouds)
    assert ok
    opt = poses[0].astype(np.float32)
    xyz = po.transform_points(opt, xyz)
    origin = po.transform_points(opt, origin[None])[0]
    if pose_tq is not None:
        xyz = po.transform_points(pose_tq, xyz)
        origin = po.transform_points(pose_tq, origin[None])[0]
    return xyz, origin


def same_float_bits(a, b):
    return np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))


def assert_grids_equal(og, gg):
    o_ijk, o_t, o_w = og.export()
    g_ijk, g_t, g_w = gg.export()
    assert len(o_ijk) == len(g_ijk)
    assert np.array_equal(o_ijk, g_ijk)
    assert np.array_equal(o_t, g_t)
    assert np.array_equal(o_w, g_w)


@pytest.mark.parametrize("frame", [0, 1])
def test_unwarp_range_data_matches_oracle(po, hg, ctx, frame):
    """100k returns swept over three control-point pairs, NaN returns kept, origin from the first
    non-NaN return (the first return is NaN). Float coordinates identical to the oracle's bit for bit
    (fp64 sin / acos of the device library against glibc's go through a cast to float)."""
    times, poses = control_points(4)
    pts = timed_scan(50, 2000, synth.pose_k(2), 7, 0.149, nan_every=97)
    assert np.isnan(pts[0, 1])
    clouds = [(int(times[0]), [0.02, -0.01, 0.3], pts)]
    o_xyz, o_origin, ok = po.unwarp_range_data(times, poses, clouds)
    assert ok
    pose_tq = None
    if frame == 1:
        pose_tq = synth.pose_inverse(synth.pose_k(1)).astype(np.float32)
        o_xyz, o_origin = oracle_chain(po, clouds, times, poses, pose_tq)
    g_xyz, g_origin = hg.unwarp_range_data(ctx, clouds, times, poses, frame=frame, pose_tq=pose_tq)
    nan = np.isnan(pts[:, 1])
    assert np.array_equal(np.isnan(g_xyz), np.isnan(o_xyz))
    if frame == 0:  # kept as they are (:1342-1345)
        assert np.array_equal(g_xyz[nan][:, [0, 2]], pts[nan][:, [0, 2]])
    diff = np.nonzero((g_xyz[~nan].view(np.uint32) != o_xyz[~nan].view(np.uint32)).any(1))[0]
    assert len(diff) == 0, (len(diff), g_xyz[~nan][diff[:3]], o_xyz[~nan][diff[:3]])
    assert same_float_bits(g_origin, o_origin)
    # every pair of control points is used
    t = times[0] + (pts[~nan, 3].astype(np.float64) * 1e7).astype(np.int64)
    assert len(set(np.searchsorted(times, t, side="right"))) >= 3


def test_insert_unwarped_bit_exact_100k(po, hg, ctx):
    """Voxel codes of a 100k-return scan unwarped over three control-point pairs and inserted into the
    three resolutions: identical to the oracle's unwarp + TransformTimedRangeData + InsertData + Insert."""
    times, poses = control_points(4)
    pts = timed_scan(50, 2000, synth.pose_k(2), 11, 0.149, nan_every=53)
    clouds = [(int(times[0]), [0.0, 0.0, 0.0], pts)]
    pose_tq = synth.pose_inverse(synth.pose_k(1)).astype(np.float32)
    res = [0.05, 0.10, 0.20]
    gg = [hg.HybridGridTSDF(ctx, r, max_blocks=1 << 15) for r in res]
    ins = [hg.TSDFRangeDataInserter3D() for _ in res]
    st = hg.insert_pyramid_unwarped(ins, clouds, 50, times, poses, gg, pose_tq=pose_tq)
    o_xyz, o_origin = oracle_chain(po, clouds, times, poses, pose_tq)
    for r, g, s in zip(res, gg, st):
        og = po.Grid(r)
        n_in, u = og.insert(o_origin, o_xyz, width=50)
        assert (s.num_hits, s.num_updates) == (n_in, u)
        assert_grids_equal(og, g)
    for g in gg:
        g.close()


def test_insert_unwarped_two_clouds_and_normals(po, hg, ctx):
    """Two clouds leave the window together (one accumulation, origin from the first cloud); the
    CLOUD_STRUCTURE normal projection reads unwarped neighbours."""
    times, poses = control_points(5)
    a = timed_scan(16, 256, synth.pose_k(1), 3, 0.09)
    b = timed_scan(16, 256, synth.pose_k(2), 4, 0.09)
    clouds = [(int(times[0]) + 1234, [0.01, 0.0, 0.1], a), (int(times[0]) + 1_000_000, [0.5, 0.5, 0.5], b)]
    for kw in ({}, {"project_sdf_distance_to_scan_normal": 1}):
        gg = [hg.HybridGridTSDF(ctx, r, max_blocks=1 << 14) for r in (0.10, 0.20)]
        ins = [hg.TSDFRangeDataInserter3D(hg.InsertOpts(**kw)) for _ in gg]
        hg.insert_pyramid_unwarped(ins, clouds, 16, times, poses, gg, pose_tq=None)
        o_xyz, o_origin = oracle_chain(po, clouds, times, poses, None)
        for r, g in zip((0.10, 0.20), gg):
            og = po.Grid(r)
            og.insert(o_origin, o_xyz, po.InsertOpts(**kw), width=16)
            assert_grids_equal(og, g)
            g.close()


def test_unwarp_time_outside_control_points(hg, ctx):
    """The reference CHECK-fails when a return's time lies outside the control points (:1355-1359): here
    HG_ERR_TIME, sticky on the grids like the other insert errors."""
    times, poses = control_points(3)
    pts = timed_scan(8, 64, synth.pose_k(1), 5, 0.2)  # 0.2 s > the 0.1 s the control points span
    g = hg.HybridGridTSDF(ctx, 0.10, max_blocks=1 << 12)
    with pytest.raises(hg.HgError, match="HG_ERR_TIME"):
        hg.insert_pyramid_unwarped([hg.TSDFRangeDataInserter3D()], [(int(times[0]), [0, 0, 0], pts)], 8, times, poses, [g])
    g.clear()
    g.close()


def test_failed_unwarped_insert_leaves_the_map_untouched(po, hg, ctx):
    """The reference aborts at :1358-1359 BEFORE anything of the scan is inserted: a call that raises HG_ERR_TIME
    must not have modified the grid (the device replaces every return of the failed call by NaN, which the
    insertion's gates drop), although half of its returns lie inside the control points."""
    times, poses = control_points(3)
    good = timed_scan(8, 64, synth.pose_k(1), 5, 0.08)
    bad = timed_scan(8, 64, synth.pose_k(2), 6, 0.2)
    g = hg.HybridGridTSDF(ctx, 0.10, max_blocks=1 << 12)
    ins = [hg.TSDFRangeDataInserter3D()]
    hg.insert_pyramid_unwarped(ins, [(int(times[0]), [0, 0, 0], good)], 8, times, poses, [g])
    before = g.export()
    assert len(before[0]) > 100
    with pytest.raises(hg.HgError, match="HG_ERR_TIME"):
        hg.insert_pyramid_unwarped(ins, [(int(times[0]), [0, 0, 0], bad)], 8, times, poses, [g])
    after = g.export()
    assert all(np.array_equal(a, b) for a, b in zip(before, after))
    g.close()


def test_unwarp_range_data_time_outside_control_points(po, hg, ctx):
    """hg_unwarp_range_data alone (no grid to carry a sticky flag): HG_ERR_TIME from the call itself, and from
    hg_unwarp_status for the enqueue-only form; the device copy a caller might feed on holds NaN only. The
    oracle's restatement refuses the same input (ok = false)."""
    times, poses = control_points(3)
    pts = timed_scan(8, 64, synth.pose_k(1), 5, 0.2)
    clouds = [(int(times[0]), [0, 0, 0], pts)]
    assert not po.unwarp_range_data(times, poses, clouds)[2]
    with pytest.raises(hg.HgError, match="HG_ERR_TIME"):
        hg.unwarp_range_data(ctx, clouds, times, poses)
    n = hg.unwarp_range_data_async(ctx, clouds, times, poses, frame=1)
    with pytest.raises(hg.HgError, match="HG_ERR_TIME"):
        hg.unwarp_status(ctx)
    ptr, _, cnt = hg.unwarp_last_device(ctx)
    assert cnt == n == len(pts)
    # read the library's device copy back (plain hipMemcpy, device to host)
    import ctypes as C
    host = np.zeros((n, 3), np.float32)
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(C.c_void_p(host.ctypes.data), C.c_void_p(ptr), C.c_size_t(host.nbytes), 2) == 0
    assert np.isnan(host).all()
    # an in-range call on the same context clears the status again
    ok = timed_scan(8, 64, synth.pose_k(1), 5, 0.08)
    hg.unwarp_range_data_async(ctx, [(int(times[0]), [0, 0, 0], ok)], times, poses)
    hg.unwarp_status(ctx)


def test_register_scan_unwarped_uses_solved_poses(po, hg, ctx):
    """hg_register_scan_unwarped = hg_problem_solve, then hg_pyramid_insert_unwarped with the solved poses:
    same poses, same voxels (the device reads the control poses where the solve left them), and the
    voxels are the oracle's for those poses."""
    from conftest import build_map
    res = [0.10, 0.20]
    n_cp = 3
    times, _ = control_points(n_cp, dt_s=0.05)
    truth = [synth.pose_k(4 + k) for k in range(n_cp)]
    scans = [synth.generate_scan(truth[k], 16, 256, stream=40 + k) for k in range(n_cp)]
    pts = timed_scan(16, 256, truth[0], 50, 0.095, nan_every=31)
    clouds = [(int(times[0]), [0, 0, 0], pts)]

    def run(unwarped_call):
        _, gg = build_map(None, (ctx, hg), res, 16, 256, 4)
        pr = hg.Problem(ctx)
        ids = [pr.add_pose(truth[0], constant=True)]
        for k in range(1, n_cp):
            ids.append(pr.add_pose(synth.pose_mul(truth[k], synth.perturbation())))
        for k in range(1, n_cp):
            pr.add_block(scans[k], gg, 1.0 / np.sqrt(len(scans[k])), ids[k], multi_res=True)
        ins = [hg.TSDFRangeDataInserter3D() for _ in res]
        if unwarped_call:
            poses, summ = hg.register_scan_unwarped(pr, ins, clouds, 16, ids, times, gg)
            ctx.synchronize()
        else:
            summ = pr.solve()
            poses = np.array([pr.get_pose(i) for i in ids])
            hg.insert_pyramid_unwarped(ins, clouds, 16, times, poses, gg)
        out = [g.export() for g in gg]
        for g in gg:
            g.close()
        pr.close()
        return poses, summ, out

    p1, s1, e1 = run(True)
    p2, s2, e2 = run(False)
    assert np.array_equal(p1, p2)
    assert s1.num_iterations == s2.num_iterations and s1.num_iterations > 1
    for a, b in zip(e1, e2):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # the oracle at the GPU's poses
    og, _ = build_map(po, None, res, 16, 256, 4)
    o_xyz, o_origin = oracle_chain(po, clouds, times, p1, None)
    for g, e in zip(og, e1):
        g.insert(o_origin, o_xyz, width=16)
        assert all(np.array_equal(x, y) for x, y in zip(g.export(), e))
