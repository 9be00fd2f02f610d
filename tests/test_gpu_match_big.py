"""GPU parity of windows beyond the LDS-resident solver's limits (10 control points / 32 TSDF blocks / 20
odometry + IMU blocks): the reference's ADAPTIVE / SYNCED_WITH_RANGE_DATA control-point sampling
(optimizing_local_trajectory_builder.cc:1162-1232; trajectory_builder_3d.lua:139-143) reaches ~36 control
points and use_multi_resolution_matching = false adds two blocks per scan (oltb.cc:392-502). Such problems are
handed to the second compilation of the solver (hg_match.hip -DHG_BIG: 48 control points, 160 TSDF blocks, 96
odometry / IMU blocks, matrices in device memory) behind the same hg_problem_* calls."""
import numpy as np
import pytest

from hectorgrapher_amd import synth

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4
POSE_TOL_RAD = 1e-4


def rot_angle(qa, qb):
    return 2.0 * np.arccos(min(1.0, abs(float(np.dot(qa, qb)))))


@pytest.fixture(scope="module")
def maps(po, hg, ctx):
    from conftest import build_map
    return build_map(po, (ctx, hg), [0.05, 0.10, 0.20], 16, 625, 10, max_blocks=1 << 16)


def track(k, spacing):
    """Control point k of a slow track through the mapped room (spacing = pose_k steps per control point)."""
    return synth.pose_k(2.0 + spacing * k)


def window(po, hg, ctx, maps, n_cp, velocities, spacing=0.25, cols=60, two_pose_factor=0.6):
    og, gg = maps
    poses = [track(i, spacing) if i == 0 else synth.pose_mul(track(i, spacing), synth.perturbation()) for i in range(n_cp)]
    op, gp = po.Problem(), hg.Problem(ctx)
    for i in range(n_cp):
        for pr in (op, gp):
            assert pr.add_pose(poses[i], i == 0) == i
            if velocities:
                pr.set_velocity(i, np.array([0.4, 0.1, 0.0]), i == 0)
    for i in range(1, n_cp):
        delta = synth.pose_mul(synth.pose_inverse(track(i, spacing)), track(i - 1, spacing))
        dq = synth.pose_mul(synth.pose_inverse(track(i - 1, spacing)), track(i, spacing))[3:]
        pts = synth.generate_scan(track(i, spacing), 16, cols, stream=500 + i)
        s = 1.0 / np.sqrt(len(pts))
        for pr, g in ((op, og), (gp, gg)):
            pr.add_odometry_block(i - 1, i, 12.0, 30.0, delta)
            if velocities:
                pr.add_imu_block(i - 1, i, 3.0, 2.0, 70.0, 0.1 * spacing, dq)
                pr.add_block(pts, [g[0], g[1], g[2]], s, i, -1, 0.0, True)
            else:
                pr.add_block(pts, [g[0], g[1], g[2]], s, i - 1, i, two_pose_factor, True)
    return op, gp


def compare_solutions(op, gp, n_cp, velocities):
    so, sg = op.solve(), gp.solve()
    assert so.num_iterations == sg.num_iterations and so.termination_reason == sg.termination_reason
    assert so.num_successful_steps == sg.num_successful_steps
    assert abs(so.final_cost - sg.final_cost) <= 1e-8 * max(1e-12, so.final_cost) + 1e-15
    for i in range(n_cp):
        a, b = op.get_pose(i), gp.get_pose(i)
        assert np.linalg.norm(a[:3] - b[:3]) < POSE_TOL_M, i
        assert rot_angle(a[3:], b[3:]) < POSE_TOL_RAD, i
        if velocities:
            np.testing.assert_allclose(gp.get_velocity(i), op.get_velocity(i), atol=1e-6)
    return sg


@pytest.mark.parametrize("velocities", [True, False])
@pytest.mark.parametrize("n_cp", [12, 20, 40])
def test_large_window_against_oracle(po, hg, ctx, maps, n_cp, velocities):
    """Windows of 12, 20 and 40 control points (9-column groups with velocities and IMU blocks, 6-column groups
    coupled by two-pose scan blocks without): evaluation and solve against the oracle. 12 and 20 take the cyclic
    reduction over the workgroup, 40 the block chain of one wavefront (more than 16 groups)."""
    op, gp = window(po, hg, ctx, maps, n_cp, velocities)
    assert gp.num_columns() == (9 if velocities else 6) * (n_cp - 1)
    c0, r0, J0, g0 = op.evaluate()
    c1, r1, g1, H1 = gp.evaluate()
    assert abs(c0 - c1) <= 1e-11 * max(1.0, abs(c0))
    np.testing.assert_allclose(r1, r0, rtol=0, atol=1e-12)
    np.testing.assert_allclose(g1, g0, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(H1, J0.T @ J0, rtol=1e-9, atol=1e-10)
    sg = compare_solutions(op, gp, n_cp, velocities)
    assert sg.num_iterations > 1
    gp.close()


@pytest.mark.parametrize("path", ["btd_chain", "btd_padded", "band"])
def test_large_window_linear_solver_paths(po, hg, ctx, maps, path, request):
    """The other factorisations of the big build (block chain, padded block form, band Cholesky) on a window of
    14 control points with velocities."""
    key = {"btd_chain": "lm_btd_chain", "btd_padded": "lm_btd_generic", "band": "lm_band"}[path]
    ctx.set_option(key, 1)
    request.addfinalizer(lambda: ctx.set_option(key, 0))
    op, gp = window(po, hg, ctx, maps, 14, True)
    compare_solutions(op, gp, 14, True)
    gp.close()


def test_two_blocks_per_scan_window_of_18_scans(po, hg, ctx, maps):
    """use_multi_resolution_matching = false (the Lua default): every scan adds a high-resolution block (its
    high-resolution cloud against the high-resolution grid) and a low-resolution block, both interpolated
    between the scan's control points (oltb.cc:392-502). 18 scans between 10 control points = 36 TSDF blocks
    (more than the plain build's 32), four scan blocks + IMU + odometry on every pair of control points."""
    og, gg = maps
    n_cp = 10
    poses = [track(i, 0.5) if i == 0 else synth.pose_mul(track(i, 0.5), synth.perturbation()) for i in range(n_cp)]
    op, gp = po.Problem(), hg.Problem(ctx)
    for i in range(n_cp):
        for pr in (op, gp):
            pr.add_pose(poses[i], i == 0)
            pr.set_velocity(i, np.array([0.4, 0.1, 0.0]), i == 0)
    for i in range(1, n_cp):
        delta = synth.pose_mul(synth.pose_inverse(track(i, 0.5)), track(i - 1, 0.5))
        dq = synth.pose_mul(synth.pose_inverse(track(i - 1, 0.5)), track(i, 0.5))[3:]
        for pr in (op, gp):
            pr.add_odometry_block(i - 1, i, 12.0, 30.0, delta)
            pr.add_imu_block(i - 1, i, 3.0, 2.0, 70.0, 0.05, dq)
    scan = 0
    for i in range(1, n_cp):
        for f in (0.3, 0.8):  # two scans between control points i - 1 and i
            at = synth.pose_k(2.0 + 0.5 * (i - 1 + f))
            high = synth.generate_scan(at, 16, 50, stream=700 + scan)
            low = synth.generate_scan(at, 8, 25, stream=800 + scan)
            for pr, g in ((op, og), (gp, gg)):
                pr.add_block(high, [g[1]], 5.0 / np.sqrt(len(high)), i - 1, i, f)   # high_resolution_grid_weight / sqrt(N)
                pr.add_block(low, [g[2]], 15.0 / np.sqrt(len(low)), i - 1, i, f)    # low_resolution_grid_weight
            scan += 1
    assert scan == 18 and gp.num_columns() == 81
    c0, r0, J0, g0 = op.evaluate()
    c1, r1, g1, H1 = gp.evaluate()
    np.testing.assert_allclose(r1, r0, rtol=0, atol=1e-12)
    np.testing.assert_allclose(H1, J0.T @ J0, rtol=1e-9, atol=1e-10)
    compare_solutions(op, gp, n_cp, True)
    gp.close()


def test_far_coupling_block_takes_the_band_path(po, hg, ctx, maps):
    """A scan block between control points 1 and 9 of a 10-state window: 81 x 81 dense, beyond the plain
    build's band storage (3240 entries) -- used to be HG_ERR_CAPACITY, now solved by the big build."""
    og, gg = maps
    op, gp = po.Problem(), hg.Problem(ctx)
    for i in range(10):
        tq = track(i, 0.5) if i == 0 else synth.pose_mul(track(i, 0.5), synth.perturbation())
        for pr in (op, gp):
            pr.add_pose(tq, i == 0)
            pr.set_velocity(i, np.zeros(3), i == 0)
    for i in range(1, 10):
        delta = synth.pose_mul(synth.pose_inverse(track(i, 0.5)), track(i - 1, 0.5))
        dq = synth.pose_mul(synth.pose_inverse(track(i - 1, 0.5)), track(i, 0.5))[3:]
        for pr in (op, gp):
            pr.add_odometry_block(i - 1, i, 12.0, 30.0, delta)
            pr.add_imu_block(i - 1, i, 3.0, 2.0, 70.0, 0.05, dq)
    pts = synth.generate_scan(track(5, 0.5), 16, 40, stream=7)
    for pr, g in ((op, og), (gp, gg)):
        pr.add_block(pts, [g[1]], 1.0 / np.sqrt(len(pts)), 1, 9, 0.5)
    c0, r0, J0, g0 = op.evaluate()
    c1, r1, g1, H1 = gp.evaluate()
    np.testing.assert_allclose(H1, J0.T @ J0, rtol=1e-9, atol=1e-10)
    compare_solutions(op, gp, 10, True)
    gp.close()


def test_capacity_beyond_the_big_limits(hg, ctx, maps):
    """HG_ERR_CAPACITY only beyond 48 control points / 160 TSDF blocks / 96 odometry + IMU blocks, or a band
    wider than the big build stores."""
    _, gg = maps
    p = hg.Problem(ctx)
    for i in range(48):
        assert p.add_pose(synth.pose_k(i * 0.1), i == 0) == i
    with pytest.raises(hg.HgError, match="HG_ERR_CAPACITY"):
        p.add_pose(synth.pose_k(5.0))
    for i in range(96):
        p.add_odometry_block(i % 47, i % 47 + 1, 1.0, 1.0, np.array([0, 0, 0, 1, 0, 0, 0], np.float64))
    with pytest.raises(hg.HgError, match="HG_ERR_CAPACITY"):
        p.add_odometry_block(0, 1, 1.0, 1.0, np.array([0, 0, 0, 1, 0, 0, 0], np.float64))
    pts = synth.generate_scan(synth.pose_k(5), 8, 8, stream=7)
    for i in range(160):
        p.add_block(pts, [gg[1]], 1.0, 1 + i % 47)
    with pytest.raises(hg.HgError, match="HG_ERR_CAPACITY"):
        p.add_block(pts, [gg[1]], 1.0, 1)
    p.close()
    q = hg.Problem(ctx)
    for i in range(48):
        q.add_pose(synth.pose_k(i * 0.1), i == 0)
        q.set_velocity(i, np.zeros(3), i == 0)
    q.add_block(pts, [gg[1]], 1.0, 1, 47, 0.5)  # couples the two ends: 423 columns dense
    with pytest.raises(hg.HgError, match="HG_ERR_CAPACITY"):
        q.evaluate()
    q.close()


def test_problem_reuse_across_sizes(po, hg, ctx, maps):
    """One hg_problem reset and refilled with a large window, then a small one, then a large one again: every
    solve equals the oracle's (the big twin is kept across resets; a problem that fits runs on the plain build)."""
    og, gg = maps
    gp = hg.Problem(ctx)
    for n_cp in (14, 4, 16):
        gp.reset()
        op = po.Problem()
        poses = [track(i, 0.3) if i == 0 else synth.pose_mul(track(i, 0.3), synth.perturbation()) for i in range(n_cp)]
        for i in range(n_cp):
            for pr in (op, gp):
                pr.add_pose(poses[i], i == 0)
        for i in range(1, n_cp):
            delta = synth.pose_mul(synth.pose_inverse(track(i, 0.3)), track(i - 1, 0.3))
            pts = synth.generate_scan(track(i, 0.3), 16, 50, stream=900 + i)
            for pr, g in ((op, og), (gp, gg)):
                pr.add_odometry_block(i - 1, i, 12.0, 30.0, delta)
                pr.add_block(pts, [g[0], g[1], g[2]], 1.0 / np.sqrt(len(pts)), i, -1, 0.0, True)
        compare_solutions(op, gp, n_cp, False)
    gp.close()


def test_register_scan_on_a_large_window(po, hg, ctx):
    """hg_register_scan with a promoted problem: the insertion reads the solved pose from the big build's
    device state -- same voxels as solve, then insert at the returned pose."""
    from conftest import build_map
    res = [0.10, 0.20]
    n_cp = 12
    scans = [synth.generate_scan(track(i, 0.3), 16, 100, stream=40 + i) for i in range(n_cp)]

    def run(fused):
        _, gg = build_map(None, (ctx, hg), res, 16, 256, 4)
        pr = hg.Problem(ctx)
        for i in range(n_cp):
            pr.add_pose(track(i, 0.3) if i == 0 else synth.pose_mul(track(i, 0.3), synth.perturbation()), i == 0)
        for i in range(1, n_cp):
            pr.add_odometry_block(i - 1, i, 12.0, 30.0, synth.pose_mul(synth.pose_inverse(track(i, 0.3)), track(i - 1, 0.3)))
            pr.add_block(scans[i], gg, 1.0 / np.sqrt(len(scans[i])), i, multi_res=True)
        ins = [hg.TSDFRangeDataInserter3D() for _ in res]
        leaving = hg.RangeData([0, 0, 0], scans[1])
        if fused:
            at, summ = hg.register_scan(pr, 1, ins, leaving, gg)
            ctx.synchronize()
        else:
            summ = pr.solve()
            at = pr.get_pose(1)
            hg.insert_pyramid(ins, leaving, gg, pose_tq=at.astype(np.float32))
        out = [g.export() for g in gg]
        for g in gg:
            g.close()
        pr.close()
        return at, summ, out

    a1, s1, e1 = run(True)
    a2, s2, e2 = run(False)
    assert np.array_equal(a1, a2) and s1.num_iterations == s2.num_iterations and s1.num_iterations > 1
    for x, y in zip(e1, e2):
        assert all(np.array_equal(u, v) for u, v in zip(x, y))
