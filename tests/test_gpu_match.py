"""GPU parity: HIP residual/normal-equation kernel and on-device LM vs the CPU oracle."""
import numpy as np
import pytest

from hectorgrapher_amd import synth

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4     # BASELINE.json: pose vs CPU within 1e-4 m / 1e-4 rad
POSE_TOL_RAD = 1e-4


def rot_angle(qa, qb):
    d = abs(float(np.dot(qa, qb)))
    return 2.0 * np.arccos(min(1.0, d))


@pytest.fixture(scope="module")
def maps(po, hg, ctx):
    from conftest import build_map
    return build_map(po, (ctx, hg), [0.05, 0.10, 0.20], 16, 625, 10, max_blocks=1 << 16)


# The matcher reads voxels either by DIRECT address (all blocks in their window slot, bounding box inside
# the window: pyramid_tsd_direct) or through the hash table (pyramid_tsd_general). Which one a test
# exercised must not be luck: these fixtures pin the path per level and assert it.
@pytest.fixture(scope="module")
def maps_direct(po, hg, ctx):
    """Windows of 2^18 blocks (25.6 m at 0.05 m): the 10 x 20 x 10 m room fits at every level."""
    from conftest import build_map
    m = build_map(po, (ctx, hg), [0.05, 0.10, 0.20], 16, 625, 10, max_blocks=1 << 18)
    for g in m[1]:
        st = g.window_status()
        assert st["direct"] and st["overflow_blocks"] == 0, st
    return m


@pytest.fixture(scope="module")
def maps_overflowed(po, hg, ctx):
    """Pools whose windows (12.8 m along y at every level) are narrower than the 20 m room: blocks
    spill into the overflow area and every lookup of every level goes through the hash table."""
    from conftest import build_map
    m = build_map(po, (ctx, hg), [0.05, 0.10, 0.20], 16, 625, 10, max_blocks=[1 << 14, 1 << 12, 1 << 10])
    for g in m[1]:
        st = g.window_status()
        assert not st["direct"], st
        assert st["overflow_blocks"] > 0 or any(e > w for e, w in zip(st["extent"], st["window"])), st
    return m


@pytest.fixture(params=["direct", "overflowed"])
def maps_by_path(request, maps_direct, maps_overflowed):
    return maps_direct if request.param == "direct" else maps_overflowed


def query(K=10, rings=16, cols=625):
    pose = synth.pose_k(K)
    pts = synth.generate_scan(pose, rings, cols, stream=K)
    guess = synth.pose_mul(pose, synth.perturbation())
    return pose, pts, guess


def both_problems(po, hg, ctx, maps, levels, multi, pts, poses, constants, pose_b=-1, factor=0.0):
    og, gg = maps
    op, gp = po.Problem(), hg.Problem(ctx)
    for tq, c in zip(poses, constants):
        op.add_pose(tq, c)
        gp.add_pose(tq, c)
    s = 1.0 / np.sqrt(len(pts))
    op.add_block(pts, [og[l] for l in levels], s, 0, pose_b, factor, multi)
    gp.add_block(pts, [gg[l] for l in levels], s, 0, pose_b, factor, multi)
    return op, gp


def compare_evaluate(op, gp):
    c0, r0, J0, g0 = op.evaluate()
    c1, r1, g1, H1 = gp.evaluate()
    assert abs(c0 - c1) <= 1e-12 * max(1.0, abs(c0))
    np.testing.assert_allclose(r1, r0, rtol=0, atol=1e-13)
    np.testing.assert_allclose(g1, g0, rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(H1, J0.T @ J0, rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize("levels,multi", [([1], False), ([0], False), ([0, 1, 2], True)])
def test_evaluate_single_pose(po, hg, ctx, maps_by_path, levels, multi):
    """Residuals, cost, gradient and J^T J through BOTH voxel-lookup paths (asserted by the fixtures)."""
    _, pts, guess = query()
    op, gp = both_problems(po, hg, ctx, maps_by_path, levels, multi, pts, [guess], [False])
    compare_evaluate(op, gp)


def test_lookup_paths_agree_bit_for_bit(hg, ctx, maps_direct, maps_overflowed):
    """The same map in a direct and in an overflowed pool: identical residuals (the two lookups return
    the same voxels, the arithmetic behind them is shared)."""
    _, pts, guess = query()
    out = []
    for m in (maps_direct, maps_overflowed):
        gp = hg.Problem(ctx)
        gp.add_pose(guess)
        gp.add_block(pts, m[1], 1.0 / np.sqrt(len(pts)), 0, multi_res=True)
        out.append(gp.evaluate())
    np.testing.assert_array_equal(out[0][1], out[1][1])
    assert out[0][0] == out[1][0]


def test_evaluate_interpolated_two_poses(po, hg, ctx, maps):
    _, pts, guess = query()
    p0 = synth.pose_k(9)
    op, gp = both_problems(po, hg, ctx, maps, [1], False, pts, [p0, guess], [False, False], 1, 0.7)
    compare_evaluate(op, gp)
    op, gp = both_problems(po, hg, ctx, maps, [0, 1, 2], True, pts, [p0, guess], [True, False], 1, 0.35)
    compare_evaluate(op, gp)


@pytest.mark.parametrize("general_lm", [False, True])
@pytest.mark.parametrize("levels,multi", [([1], False), ([0, 1, 2], True)])
def test_solve_single_pose(po, hg, ctx, maps_by_path, levels, multi, general_lm, monkeypatch):
    """Both LM tails: the register-resident single-pose step and (context option lm_general) the general one;
    both voxel-lookup paths."""
    truth, pts, guess = query()
    with ctx.option("lm_general", 1 if general_lm else 0):   # (read when the solve is prepared)
        op, gp = both_problems(po, hg, ctx, maps_by_path, levels, multi, pts, [guess], [False])
        so, sg = op.solve(), gp.solve()
    a, b = op.get_pose(0), gp.get_pose(0)
    assert np.linalg.norm(a[:3] - b[:3]) < POSE_TOL_M
    assert rot_angle(a[3:], b[3:]) < POSE_TOL_RAD
    assert so.num_iterations == sg.num_iterations
    assert so.termination_reason == sg.termination_reason
    assert so.num_successful_steps == sg.num_successful_steps
    assert abs(so.final_cost - sg.final_cost) <= 1e-9 * so.final_cost
    # and the solve actually registers the scan
    assert np.linalg.norm(b[:3] - truth[:3]) < 0.02 < np.linalg.norm(guess[:3] - truth[:3])


def test_solve_two_pose_window(po, hg, ctx, maps):
    """First control point constant (oltb.cc:1268-1275), scan between the two control points."""
    truth, pts, guess = query()
    p0 = synth.pose_k(9)
    op, gp = both_problems(po, hg, ctx, maps, [1], False, pts, [p0, guess], [True, False], 1, 0.8)
    so, sg = op.solve(), gp.solve()
    for i in range(2):
        a, b = op.get_pose(i), gp.get_pose(i)
        assert np.linalg.norm(a[:3] - b[:3]) < POSE_TOL_M
        assert rot_angle(a[3:], b[3:]) < POSE_TOL_RAD
    assert so.num_iterations == sg.num_iterations
    np.testing.assert_array_equal(gp.get_pose(0), p0)


def test_empty_cloud_and_empty_grid(po, hg, ctx, maps):
    """KA-3: an empty-cloud block leaves the state untouched; an empty grid gives -tau, zero grad."""
    _, pts, guess = query()
    gp = hg.Problem(ctx)
    gp.add_pose(guess)
    gp.add_block(np.zeros((0, 3), np.float32), [maps[1][1]], 1.0, 0)
    s = gp.solve()
    np.testing.assert_array_equal(gp.get_pose(0), guess)
    empty = hg.HybridGridTSDF(ctx, 0.1, max_blocks=16)
    gp = hg.Problem(ctx)
    gp.add_pose(guess)
    gp.add_block(pts[:100], [empty], 1.0, 0)
    c, r, g, H = gp.evaluate()
    np.testing.assert_allclose(r, -0.25, atol=1e-7)
    assert np.all(g == 0) and np.all(H == 0)


def test_ceres_scan_matcher_shape(po, hg, ctx, maps):
    truth, pts, guess = query()
    m = hg.CeresScanMatcher3D(ctx, [1.0, 6.0])
    pose, summary = m.Match(guess, [(pts, maps[1][1]), (pts[::4], maps[1][2])])
    op = po.Problem()
    op.add_pose(guess)
    op.add_block(pts, [maps[0][1]], 1.0 / np.sqrt(len(pts)), 0)
    op.add_block(pts[::4], [maps[0][2]], 6.0 / np.sqrt(len(pts[::4])), 0)
    op.solve()
    a = op.get_pose(0)
    assert np.linalg.norm(a[:3] - pose[:3]) < POSE_TOL_M
    assert rot_angle(a[3:], pose[3:]) < POSE_TOL_RAD


def test_register_scan_equals_solve_then_insert(po, hg, ctx):
    """hg_register_scan (pose handed to the inserter in device memory) == solve, read the pose
    back, insert at float(pose) == the oracle doing the same on the CPU."""
    from conftest import build_map
    res = [0.05, 0.10, 0.20]
    og, gg = build_map(po, (ctx, hg), res, 16, 400, 4, max_blocks=1 << 16)
    truth = synth.pose_k(4)
    pts = synth.generate_scan(truth, 16, 400, stream=4)
    guess = synth.pose_mul(truth, synth.perturbation())
    s = 1.0 / np.sqrt(len(pts))
    op = po.Problem()
    op.add_pose(guess)
    op.add_block(pts, og, s, 0, multi_res=True)
    op.solve()
    est = op.get_pose(0)
    for g in og:
        g.insert([0, 0, 0], pts, pose_tq=est.astype(np.float32))
    gp = hg.Problem(ctx)
    gp.add_pose(guess)
    gp.add_block(pts, gg, s, 0, multi_res=True)
    ins = [hg.TSDFRangeDataInserter3D() for _ in gg]
    pose, summary = hg.register_scan(gp, 0, ins, hg.RangeData([0, 0, 0], pts), gg)
    assert np.linalg.norm(pose[:3] - est[:3]) < POSE_TOL_M
    assert rot_angle(pose[3:], est[3:]) < POSE_TOL_RAD
    if np.array_equal(pose.astype(np.float32), est.astype(np.float32)):
        # identical float pose => identical voxels
        for o, g in zip(og, gg):
            g.status()
            assert all(np.array_equal(a, b) for a, b in zip(o.export(), g.export()))
    else:
        for o, g in zip(og, gg):
            assert abs(o.count() - g.count()) < 0.01 * o.count()


def test_sliding_window_three_control_points(po, hg, ctx, maps):
    """Window of 3 control points (first constant, oltb.cc:1268-1275) with four scans: one exactly on
    a control point (TSDFSpaceCostFunction3D), three between control points (interpolated blocks),
    high- and low-resolution grids as separate blocks (oltb.cc:392-502). 12 free columns."""
    og, gg = maps
    poses = [synth.pose_k(8), synth.pose_mul(synth.pose_k(9), synth.perturbation()),
             synth.pose_mul(synth.pose_k(10), synth.perturbation())]
    const = [True, False, False]
    clouds = [synth.generate_scan(synth.pose_k(k), 16, 200, stream=40 + i)
              for i, k in enumerate((9, 9, 10, 10))]
    # (cloud, grid level, pose_a, pose_b, factor)
    blocks = [(0, 1, 1, -1, 0.0), (1, 2, 0, 1, 0.6), (2, 1, 1, 2, 0.5), (3, 2, 1, 2, 0.9)]
    op, gp = po.Problem(), hg.Problem(ctx)
    for tq, c in zip(poses, const):
        op.add_pose(tq, c)
        gp.add_pose(tq, c)
    for ci, lvl, a, b, f in blocks:
        s = (1.0 if lvl == 1 else 0.5) / np.sqrt(len(clouds[ci]))
        op.add_block(clouds[ci], [og[lvl]], s, a, b, f)
        gp.add_block(clouds[ci], [gg[lvl]], s, a, b, f)
    assert gp.num_columns() == 12 and gp.num_residuals() == op.evaluate()[1].shape[0]
    compare_evaluate(op, gp)
    so, sg = op.solve(), gp.solve()
    assert so.num_iterations == sg.num_iterations and so.termination_reason == sg.termination_reason
    for i in range(3):
        a, b = op.get_pose(i), gp.get_pose(i)
        assert np.linalg.norm(a[:3] - b[:3]) < POSE_TOL_M
        assert rot_angle(a[3:], b[3:]) < POSE_TOL_RAD
    np.testing.assert_array_equal(gp.get_pose(0), poses[0])


def _window_with_imu_and_odometry(po, hg, ctx, maps, with_tsdf):
    og, gg = maps
    n_cp = 4
    poses = [synth.pose_k(7 + i) if i == 0 else synth.pose_mul(synth.pose_k(7 + i), synth.perturbation())
             for i in range(n_cp)]
    vels = [np.array([0.5, 0.2, 0.0]) + 0.05 * i for i in range(n_cp)]
    op, gp = po.Problem(), hg.Problem(ctx)
    for i in range(n_cp):
        for pr in (op, gp):
            pr.add_pose(poses[i], i == 0)
            pr.set_velocity(i, vels[i], i == 0)        # oltb.cc:1268-1275: first state constant
    for i in range(1, n_cp):
        delta = synth.pose_mul(synth.pose_inverse(synth.pose_k(7 + i)), synth.pose_k(6 + i))
        dq = synth.pose_mul(synth.pose_inverse(synth.pose_k(6 + i)), synth.pose_k(7 + i))[3:]
        for pr in (op, gp):
            pr.add_odometry_block(i - 1, i, 12.0, 30.0, delta)
            pr.add_imu_block(i - 1, i, 3.0, 2.0, 70.0, 0.1, dq)
    if with_tsdf:
        for i in range(1, n_cp):
            pts = synth.generate_scan(synth.pose_k(7 + i), 16, 150, stream=60 + i)
            s = 1.0 / np.sqrt(len(pts))
            op.add_block(pts, [og[1]], s, i)
            gp.add_block(pts, [gg[1]], s, i)
    return op, gp, n_cp


@pytest.mark.parametrize("with_tsdf", [False, True])
def test_imu_preintegration_and_odometry_blocks(po, hg, ctx, maps, with_tsdf):
    """the non-TSDF blocks of the sliding window (oltb.cc:928-1074) on the device —
    9 columns per free control point (pose 6 + velocity 3)."""
    op, gp, n_cp = _window_with_imu_and_odometry(po, hg, ctx, maps, with_tsdf)
    assert gp.num_columns() == 9 * (n_cp - 1) == 27
    c0, r0, J0, g0 = op.evaluate()
    c1, r1, g1, H1 = gp.evaluate()
    assert abs(c0 - c1) <= 1e-11 * max(1.0, abs(c0))
    np.testing.assert_allclose(r1, r0, rtol=0, atol=1e-12)
    np.testing.assert_allclose(g1, g0, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(H1, J0.T @ J0, rtol=1e-9, atol=1e-10)
    so, sg = op.solve(), gp.solve()
    assert so.num_iterations == sg.num_iterations and so.termination_reason == sg.termination_reason
    assert abs(so.final_cost - sg.final_cost) <= 1e-8 * max(1e-12, so.final_cost) + 1e-15
    for i in range(n_cp):
        a, b = op.get_pose(i), gp.get_pose(i)
        assert np.linalg.norm(a[:3] - b[:3]) < POSE_TOL_M
        assert rot_angle(a[3:], b[3:]) < POSE_TOL_RAD
        np.testing.assert_allclose(gp.get_velocity(i), op.get_velocity(i), atol=1e-6)


def _unwarped_window(po, hg, ctx, maps, levels, multi, per_sub):
    """Three control points 0.1 s apart (first constant); two clouds whose returns are spread over
    the sweep time so that they straddle control points and partly fall outside the window."""
    from hectorgrapher_amd import api
    og, gg = maps
    tick0 = 636_000_000_000_000_000
    control = [tick0, tick0 + api.from_seconds(0.1), tick0 + api.from_seconds(0.2)]
    poses = [synth.pose_k(8), synth.pose_mul(synth.pose_k(9), synth.perturbation()),
             synth.pose_mul(synth.pose_k(10), synth.perturbation())]
    clouds = []
    for ci, (k, t_cloud, t0, t1) in enumerate([(9, 0.02, -0.05, 0.13), (10, 0.12, -0.04, 0.11)]):
        pts = synth.generate_scan(synth.pose_k(k), 16, 120, stream=80 + ci)
        times = np.linspace(t0, t1, len(pts)).astype(np.float32)
        clouds.append((tick0 + api.from_seconds(t_cloud), pts, times))
    op, gp = po.Problem(), hg.Problem(ctx)
    for i, tq in enumerate(poses):
        op.add_pose(tq, i == 0)
        gp.add_pose(tq, i == 0)
    added = api.add_per_point_matching_residuals(gp, [0, 1, 2], control, clouds, [gg[l] for l in levels],
                                                 1.3, per_sub, multi_res=multi)
    # the reference's problem: one block per subdivision, in cloud / subdivision order
    order = []
    for ci, (t_cloud, pts, times) in enumerate(clouds):
        s = 1.3 / np.sqrt(len(pts))
        for (s0, s1, a, b, ratio) in api.per_point_subdivisions(times, t_cloud, control, per_sub):
            op.add_block(pts[s0:s1], [og[l] for l in levels], s, a, b, ratio, multi)
            order.extend((ci, i) for i in range(s0, s1))
    dev_order = [(ci, int(i)) for (ci, a, b, idx, f) in added for i in idx]
    return op, gp, order, dev_order, poses


@pytest.mark.parametrize("levels,multi,per_sub", [([0, 1, 2], True, 4), ([2], False, 1), ([1], False, 7)])
def test_per_point_unwarping_blocks(po, hg, ctx, maps, levels, multi, per_sub):
    """SURVEY 8f-2 (use_per_point_unwarping, oltb.cc:513-683): subdivision blocks with their own
    interpolation ratio; per_sub == 1 on one grid is InterpolatedTSDFPerPointSpaceCostFunction3D."""
    op, gp, order, dev_order, poses = _unwarped_window(po, hg, ctx, maps, levels, multi, per_sub)
    assert sorted(order) == sorted(dev_order) and 0 < len(order) < 2 * 16 * 120
    c0, r0, J0, g0 = op.evaluate()
    c1, r1, g1, H1 = gp.evaluate()
    assert gp.num_columns() == 12 and gp.num_residuals() == len(r0) == len(order)
    pos = {key: i for i, key in enumerate(order)}
    perm = np.array([pos[key] for key in dev_order])
    np.testing.assert_allclose(r1, r0[perm], rtol=0, atol=1e-13)
    assert abs(c0 - c1) <= 1e-12 * max(1.0, abs(c0))
    np.testing.assert_allclose(g1, g0, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(H1, J0.T @ J0, rtol=1e-9, atol=1e-12)
    so, sg = op.solve(), gp.solve()
    assert so.num_iterations == sg.num_iterations and so.termination_reason == sg.termination_reason
    assert abs(so.final_cost - sg.final_cost) <= 1e-8 * max(1e-12, so.final_cost) + 1e-15
    for i in range(3):
        a, b = op.get_pose(i), gp.get_pose(i)
        assert np.linalg.norm(a[:3] - b[:3]) < POSE_TOL_M
        assert rot_angle(a[3:], b[3:]) < POSE_TOL_RAD
    np.testing.assert_array_equal(gp.get_pose(0), poses[0])


def test_unwarped_block_mixed_with_per_scan_blocks(po, hg, ctx, maps):
    """An unwarped block and ordinary per-scan blocks in one problem share the iteration's ticket."""
    og, gg = maps
    poses = [synth.pose_k(8), synth.pose_mul(synth.pose_k(9), synth.perturbation())]
    pts = synth.generate_scan(synth.pose_k(9), 16, 100, stream=91)
    f = np.linspace(0.0, 1.0, len(pts))
    op, gp = po.Problem(), hg.Problem(ctx)
    for i, tq in enumerate(poses):
        op.add_pose(tq, False)
        gp.add_pose(tq, False)
    s = 1.0 / np.sqrt(len(pts))
    gp.add_block(pts, [gg[1]], s, 1)
    gp.add_unwarped_block(pts, f, [gg[0], gg[1], gg[2]], s, 0, 1, multi_res=True)
    gp.add_block(pts, [gg[2]], 0.5 * s, 0, 1, 0.25)
    op.add_block(pts, [og[1]], s, 1)
    for i in range(len(pts)):
        op.add_block(pts[i:i + 1], [og[0], og[1], og[2]], s, 0, 1, float(f[i]), True)
    op.add_block(pts, [og[2]], 0.5 * s, 0, 1, 0.25)
    compare_evaluate(op, gp)
    so, sg = op.solve(), gp.solve()
    assert so.num_iterations == sg.num_iterations and so.termination_reason == sg.termination_reason
    for i in range(2):
        a, b = op.get_pose(i), gp.get_pose(i)
        assert np.linalg.norm(a[:3] - b[:3]) < POSE_TOL_M
        assert rot_angle(a[3:], b[3:]) < POSE_TOL_RAD


def test_unwarped_block_argument_errors(hg, ctx, maps):
    _, gg = maps
    p = hg.Problem(ctx)
    p.add_pose(synth.pose_k(0))
    p.add_pose(synth.pose_k(1))
    pts = np.zeros((4, 3), np.float32)
    with pytest.raises(hg.HgError):
        p.add_unwarped_block(pts, np.zeros(4), [gg[0]], 1.0, 0, -1)     # needs two control points
    with pytest.raises(hg.HgError):
        p.add_unwarped_block(pts, np.zeros(4), [gg[0]], 1.0, 0, 0)
    with pytest.raises(hg.HgError):
        p.add_unwarped_block(pts, np.zeros(3), [gg[0]], 1.0, 0, 1)      # one ratio per return


def test_window_of_ten_control_points(po, hg, ctx, maps):
    """ct_window_horizon / ct_window_rate ~ 9 control points -> 81 free columns
    (first state constant); block-tridiagonal normal equations in band storage on the device."""
    og, gg = maps
    n_cp = 10
    poses = [synth.pose_k(3 + i) if i == 0 else synth.pose_mul(synth.pose_k(3 + i), synth.perturbation())
             for i in range(n_cp)]
    op, gp = po.Problem(), hg.Problem(ctx)
    for i in range(n_cp):
        for pr in (op, gp):
            pr.add_pose(poses[i], i == 0)
            pr.set_velocity(i, np.array([0.5, 0.2, 0.0]) + 0.01 * i, i == 0)
    for i in range(1, n_cp):
        delta = synth.pose_mul(synth.pose_inverse(synth.pose_k(3 + i)), synth.pose_k(2 + i))
        dq = synth.pose_mul(synth.pose_inverse(synth.pose_k(2 + i)), synth.pose_k(3 + i))[3:]
        for pr in (op, gp):
            pr.add_odometry_block(i - 1, i, 12.0, 30.0, delta)
            pr.add_imu_block(i - 1, i, 3.0, 2.0, 70.0, 0.1, dq)
    for i in range(1, n_cp):   # one scan on each control point, one between each pair
        pts = synth.generate_scan(synth.pose_k(3 + i), 16, 60, stream=120 + i)
        s = 1.0 / np.sqrt(len(pts))
        for pr, g in ((op, og), (gp, gg)):
            pr.add_block(pts, [g[0], g[1], g[2]], s, i, -1, 0.0, True)
            pr.add_block(pts, [g[1]], 0.5 * s, i - 1, i, 0.4)
    assert gp.num_columns() == 81
    c0, r0, J0, g0 = op.evaluate()
    c1, r1, g1, H1 = gp.evaluate()
    assert abs(c0 - c1) <= 1e-11 * max(1.0, abs(c0))
    np.testing.assert_allclose(r1, r0, rtol=0, atol=1e-12)
    np.testing.assert_allclose(g1, g0, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(H1, J0.T @ J0, rtol=1e-9, atol=1e-10)
    so, sg = op.solve(), gp.solve()
    assert so.num_iterations == sg.num_iterations and so.termination_reason == sg.termination_reason
    assert abs(so.final_cost - sg.final_cost) <= 1e-8 * max(1e-12, so.final_cost) + 1e-15
    for i in range(n_cp):
        a, b = op.get_pose(i), gp.get_pose(i)
        assert np.linalg.norm(a[:3] - b[:3]) < POSE_TOL_M
        assert rot_angle(a[3:], b[3:]) < POSE_TOL_RAD
        np.testing.assert_allclose(gp.get_velocity(i), op.get_velocity(i), atol=1e-6)


@pytest.mark.parametrize("velocities", [True, False])
@pytest.mark.parametrize("path", ["twisted", "cyclic_reduction", "btd_chain", "btd_padded", "band"])
def test_window_linear_solver_paths(po, hg, ctx, maps, path, velocities, request):
    """The factorisations of the window's normal equations -- the twisted block factorisation from both ends of the
    chain with a wavefront per group (the default up to nine groups), block cyclic reduction over the workgroup
    (uniform 9- or 6-column groups), the block chain in one wavefront (registers, forward pass folded in),
    the padded block form, the band Cholesky -- give the oracle's solve: same iterations, poses within
    tolerance. With velocities: 9-column groups (pose + velocity, IMU blocks); without: 6-column groups
    coupled by two-pose scan blocks. (Context options, read at every solve.)"""
    key = {"cyclic_reduction": "lm_btd_cr", "btd_chain": "lm_btd_chain", "btd_padded": "lm_btd_generic", "band": "lm_band"}.get(path)
    if key:
        ctx.set_option(key, 1)
        request.addfinalizer(lambda: ctx.set_option(key, 0))
    og, gg = maps
    n_cp = 7
    poses = [synth.pose_k(3 + i) if i == 0 else synth.pose_mul(synth.pose_k(3 + i), synth.perturbation())
             for i in range(n_cp)]
    op, gp = po.Problem(), hg.Problem(ctx)
    for i in range(n_cp):
        for pr in (op, gp):
            pr.add_pose(poses[i], i == 0)
            if velocities:
                pr.set_velocity(i, np.array([0.4, 0.1, 0.0]), i == 0)
    for i in range(1, n_cp):
        delta = synth.pose_mul(synth.pose_inverse(synth.pose_k(3 + i)), synth.pose_k(2 + i))
        dq = synth.pose_mul(synth.pose_inverse(synth.pose_k(2 + i)), synth.pose_k(3 + i))[3:]
        pts = synth.generate_scan(synth.pose_k(3 + i), 16, 100, stream=300 + i)
        s = 1.0 / np.sqrt(len(pts))
        for pr, g in ((op, og), (gp, gg)):
            pr.add_odometry_block(i - 1, i, 12.0, 30.0, delta)
            if velocities:
                pr.add_imu_block(i - 1, i, 3.0, 2.0, 70.0, 0.1, dq)
                pr.add_block(pts, [g[0], g[1], g[2]], s, i, -1, 0.0, True)
            else:
                pr.add_block(pts, [g[0], g[1], g[2]], s, i - 1, i, 0.6, True)   # couples the two poses: band of 11
    assert gp.num_columns() == (9 if velocities else 6) * (n_cp - 1)
    so, sg = op.solve(), gp.solve()
    assert so.num_iterations == sg.num_iterations and so.termination_reason == sg.termination_reason
    assert so.num_successful_steps == sg.num_successful_steps
    for i in range(n_cp):
        a, b = op.get_pose(i), gp.get_pose(i)
        assert np.linalg.norm(a[:3] - b[:3]) < POSE_TOL_M
        assert rot_angle(a[3:], b[3:]) < POSE_TOL_RAD
        if velocities:
            np.testing.assert_allclose(gp.get_velocity(i), op.get_velocity(i), atol=1e-6)


def test_band_beyond_the_plain_build_is_promoted(hg, ctx, maps):
    """A block that couples the two ends of a 10-state window does not fit the LDS-resident band storage:
    the problem moves to the big build of the solver (tests/test_gpu_match_big.py compares it with the
    oracle) instead of returning HG_ERR_CAPACITY."""
    _, gg = maps
    p = hg.Problem(ctx)
    for i in range(10):
        p.add_pose(synth.pose_k(i), i == 0)
        p.set_velocity(i, np.zeros(3), i == 0)
    pts = synth.generate_scan(synth.pose_k(5), 16, 20, stream=7)
    p.add_block(pts, [gg[1]], 1.0, 1, 9, 0.5)
    cost, r, g, H = p.evaluate()
    assert np.isfinite(cost) and H.shape == (81, 81)


def test_free_velocity_without_imu_block(po, hg, ctx, maps):
    """A velocity parameter block that no residual touches (odometry only): its columns are zero and
    lie outside the band of the coupled columns; the band assembly must skip, not alias, them."""
    og, gg = maps
    op, gp = po.Problem(), hg.Problem(ctx)
    for i in range(2):
        tq = synth.pose_k(2 + i) if i == 0 else synth.pose_mul(synth.pose_k(2 + i), synth.perturbation())
        for pr in (op, gp):
            pr.add_pose(tq, i == 0)
            pr.set_velocity(i, np.array([0.5, 0.2, 0.0]), i == 0)
    pts = synth.generate_scan(synth.pose_k(3), 16, 40, stream=77)
    s = 1.0 / np.sqrt(len(pts))
    op.add_block(pts, og, s, 1, -1, 0.0, True)
    gp.add_block(pts, gg, s, 1, -1, 0.0, True)
    delta = synth.pose_mul(synth.pose_inverse(synth.pose_k(3)), synth.pose_k(2))
    for pr in (op, gp):
        pr.add_odometry_block(0, 1, 12.0, 30.0, delta)
    assert gp.num_columns() == 9
    compare_evaluate(op, gp)
    so, sg = op.solve(), gp.solve()
    assert so.num_iterations == sg.num_iterations and so.termination_reason == sg.termination_reason
    a, b = op.get_pose(1), gp.get_pose(1)
    assert np.linalg.norm(a[:3] - b[:3]) < POSE_TOL_M and rot_angle(a[3:], b[3:]) < POSE_TOL_RAD


def test_solve_batch_equals_individual_solves(po, hg, ctx):
    """hg_problem_solve_batch: independent single-pose problems share their launches; every problem
    keeps its own solver state, so poses and summaries are those of solving them one by one (and of
    the oracle). Scans of different sizes and different pyramids in one batch; a problem of another
    shape in the list makes the call fall back to sequential solves."""
    import torch
    dev = torch.device("cuda", 0)
    grids_a = [hg.HybridGridTSDF(ctx, r, max_blocks=1 << 16) for r in (0.05, 0.10, 0.20)]
    grids_b = [hg.HybridGridTSDF(ctx, r, max_blocks=1 << 16) for r in (0.10, 0.20)]
    ogrids_a = [po.Grid(r) for r in (0.05, 0.10, 0.20)]
    for k in range(4):
        pose = synth.pose_k(k)
        loc = synth.transform_points(pose, synth.generate_scan(pose, 32, 900, stream=k))
        for g in grids_a + grids_b:
            hg.TSDFRangeDataInserter3D().Insert(hg.RangeData(pose[:3], loc), g)
        for g in ogrids_a:
            g.insert(pose[:3], loc)
    cases = []
    for j, (rings, cols) in enumerate([(32, 900), (16, 625), (50, 1200), (8, 100), (32, 901), (20, 333)]):
        pose = synth.pose_k(4 + j % 3)
        pts = synth.generate_scan(pose, rings, cols, stream=40 + j)
        guess = synth.pose_mul(pose, synth.perturbation())
        cases.append((pts, guess, grids_a if j % 2 == 0 else grids_b))

    def build():
        ps = []
        for pts, guess, gl in cases:
            p = hg.Problem(ctx)
            i = p.add_pose(guess)
            p.add_block(torch.from_numpy(pts).to(dev), gl, 1.0 / np.sqrt(len(pts)), i, multi_res=True)
            ps.append(p)
        return ps

    single = build()
    s_single = [p.solve() for p in single]
    batch = build()
    s_batch = hg.solve_batch(batch)
    # the batched residual pass sums the normal equations over 256-return workgroups, the single
    # solve over 512-return ones: same terms, another association, so the last bits may differ
    for a, b, sa, sb in zip(single, batch, s_single, s_batch):
        np.testing.assert_allclose(b.get_pose(0), a.get_pose(0), rtol=0, atol=1e-9)
        assert (sa.num_iterations, sa.termination_type, sa.termination_reason) == \
               (sb.num_iterations, sb.termination_type, sb.termination_reason)
        assert abs(sa.final_cost - sb.final_cost) <= 1e-12 * max(1.0, abs(sa.final_cost))
        assert abs(sa.initial_cost - sb.initial_cost) <= 1e-12 * max(1.0, abs(sa.initial_cost))
    # against the oracle for the first case
    pr = po.Problem()
    i = pr.add_pose(cases[0][1])
    pr.add_block(cases[0][0], ogrids_a, 1.0 / np.sqrt(len(cases[0][0])), i, multi_res=True)
    so = pr.solve()
    assert so.num_iterations == s_batch[0].num_iterations
    assert np.abs(pr.get_pose(i) - batch[0].get_pose(0)).max() < 1e-9
    # mixed shapes: a two-pose problem in the list -> sequential fallback, same results
    mixed = build()[:2]
    p2 = hg.Problem(ctx)
    a_ = p2.add_pose(cases[0][1], True)
    b_ = p2.add_pose(cases[1][1])
    p2.add_block(torch.from_numpy(cases[0][0]).to(dev), grids_a, 1e-2, a_, b_, 0.4, multi_res=True)
    s_mixed = hg.solve_batch(mixed + [p2])
    assert np.array_equal(mixed[0].get_pose(0), single[0].get_pose(0))
    assert s_mixed[2].num_iterations >= 1


@pytest.mark.parametrize("count,fold,part_at", [(9, 1, 2), (50, 1, 2), (50, 0, 2), (50, 1, 1), (50, 1, 0)])
def test_solve_batch_with_several_tiles_per_workgroup(po, hg, ctx, count, fold, part_at):
    """From 8 problems on the batched residual pass gives a workgroup two tiles of 256 returns, from 48 on four
    (accumulators run through, next return prefetched): ragged sizes -- fewer returns than one tile, one return more
    than a workgroup's share, sizes that leave the last workgroup's later tiles empty -- against solving one by one,
    and the first two problems against the oracle. From 32 problems on the returns are level-partitioned: classified
    by the residual launch in front (`partition_fold`, the default) or by a lookup pass of its own, at LM iteration
    `partition_at` (0: no launch in front, the classify kernel)."""
    import torch
    with ctx.option("partition_fold", fold), ctx.option("partition_at", part_at):
        _several_tiles_case(po, hg, ctx, count)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_level_partition_keeps_iterations_and_termination_over_seeds(po, hg, ctx, seed):
    """the partition re-orders a problem's returns in the middle of its solve, so the accept / reject and
    tolerance tests of that iteration compare sums of different association. 4 x 50 more problems (other scans, other
    guesses): every one ends with the iteration count and termination of its own solve."""
    _several_tiles_case(po, hg, ctx, 50, seed=seed)


def _several_tiles_case(po, hg, ctx, count, seed=0):
    import torch
    dev = torch.device("cuda", 0)
    res = (0.05, 0.10, 0.20)
    grids = [hg.HybridGridTSDF(ctx, r, max_blocks=1 << 16) for r in res]
    ogrids = [po.Grid(r) for r in res]
    for k in range(4):
        pose = synth.pose_k(k)
        loc = synth.transform_points(pose, synth.generate_scan(pose, 32, 900, stream=k))
        for g in grids:
            hg.TSDFRangeDataInserter3D().Insert(hg.RangeData(pose[:3], loc), g)
        for g in ogrids:
            g.insert(pose[:3], loc)
    shapes = [(8, 25), (16, 64), (16, 65), (8, 257), (4, 513), (32, 33), (2, 1025), (1, 255), (8, 129), (3, 683)]
    cases = []
    for j in range(count):
        rings, cols = shapes[j % len(shapes)]
        pose = synth.pose_k(4 + j % 3)
        pts = synth.generate_scan(pose, rings, cols, stream=400 + j + 1000 * seed)
        guess = synth.pose_mul(pose, synth.perturbation())
        if seed:  # (another guess per problem: the fixed perturbation scaled and turned)
            rng = np.random.default_rng(77 + 100 * seed + j)
            guess = guess.copy()
            guess[:3] += rng.uniform(-0.02, 0.02, 3)
        cases.append((pts, guess))

    def build():
        ps = []
        for pts, guess in cases:
            p = hg.Problem(ctx)
            i = p.add_pose(guess)
            p.add_block(torch.from_numpy(pts).to(dev), grids, 1.0 / np.sqrt(len(pts)), i, multi_res=True)
            ps.append(p)
        return ps

    single = build()
    s_single = [p.solve() for p in single]
    batch = build()
    s_batch = hg.solve_batch(batch)
    for a, b, sa, sb in zip(single, batch, s_single, s_batch):
        np.testing.assert_allclose(b.get_pose(0), a.get_pose(0), rtol=0, atol=1e-9)
        assert (sa.num_iterations, sa.termination_type, sa.termination_reason) == \
               (sb.num_iterations, sb.termination_type, sb.termination_reason)
        assert abs(sa.final_cost - sb.final_cost) <= 1e-12 * max(1.0, abs(sa.final_cost))
    for j in range(2):
        pr = po.Problem()
        i = pr.add_pose(cases[j][1])
        pr.add_block(cases[j][0], ogrids, 1.0 / np.sqrt(len(cases[j][0])), i, multi_res=True)
        so = pr.solve()
        assert so.num_iterations == s_batch[j].num_iterations
        assert np.abs(pr.get_pose(i) - batch[j].get_pose(0)).max() < 1e-9


def test_solve_batch_async_keeps_several_batches_in_flight(po, hg, ctx):
    """hg_problem_solve_batch_async: three batches (50 problems -- the level partition is on --, 9 and 5, the last one
    with a two-pose problem that makes it fall back to per-problem enqueues) are enqueued one behind the other before
    any of them is fetched, twice over so that the job-table staging ring wraps; every problem ends where
    hg_problem_solve_batch put it (bitwise: same launches, same order of sums)."""
    import torch
    dev = torch.device("cuda", 0)
    res = (0.05, 0.10, 0.20)
    grids = [hg.HybridGridTSDF(ctx, r, max_blocks=1 << 16) for r in res]
    for k in range(4):
        pose = synth.pose_k(k)
        loc = synth.transform_points(pose, synth.generate_scan(pose, 32, 900, stream=k))
        for g in grids:
            hg.TSDFRangeDataInserter3D().Insert(hg.RangeData(pose[:3], loc), g)
    shapes = [(8, 25), (16, 64), (16, 65), (8, 257), (4, 513), (32, 33), (2, 1025), (1, 255), (8, 129), (3, 683)]
    cases = []
    for j in range(64):
        rings, cols = shapes[j % len(shapes)]
        pose = synth.pose_k(4 + j % 3)
        pts = synth.generate_scan(pose, rings, cols, stream=900 + j)
        cases.append((torch.from_numpy(pts).to(dev), len(pts), synth.pose_mul(pose, synth.perturbation())))

    def build(lo, hi, with_two_pose=False):
        ps = []
        for d, n, guess in cases[lo:hi]:
            p = hg.Problem(ctx)
            i = p.add_pose(guess)
            p.add_block(d, grids, 1.0 / np.sqrt(n), i, multi_res=True)
            ps.append(p)
        if with_two_pose:
            p2 = hg.Problem(ctx)
            a_ = p2.add_pose(cases[0][2], True)
            b_ = p2.add_pose(cases[1][2])
            p2.add_block(cases[0][0], grids, 1e-2, a_, b_, 0.4, multi_res=True)
            ps.append(p2)
        return ps

    cuts = [(0, 50, False), (50, 59, False), (59, 63, True)]
    ref = [build(*c) for c in cuts]
    s_ref = [hg.solve_batch(b) for b in ref]
    for _ in range(2):
        flight = [build(*c) for c in cuts]
        for b in flight:
            hg.solve_batch_async(b)
        s_flight = [hg.fetch_batch(b) for b in flight]
        for rb, fb, rs, fs in zip(ref, flight, s_ref, s_flight):
            for rp, fp, a, b in zip(rb, fb, rs, fs):
                for i in range(2 if rp is rb[-1] and len(rb) == 5 else 1):
                    assert np.array_equal(rp.get_pose(i), fp.get_pose(i))
                assert (a.num_iterations, a.termination_type, a.termination_reason, a.final_cost) == \
                       (b.num_iterations, b.termination_type, b.termination_reason, b.final_cost)
    # a problem that has not been fetched cannot be fetched twice
    with pytest.raises(Exception):
        hg.fetch_batch(flight[1][:1])


def test_register_scan_batch_equals_individual_registrations(po, hg, ctx):
    """hg_register_scan_batch: the registration step of several independent submaps with shared
    launches (batched matcher + insert kernels over a table of pyramids) gives every submap what
    hg_register_scan gives it on its own: same iteration counts and termination, poses equal to the
    rounding of the normal-equation sums, and -- the float casts of those poses being equal --
    bit-identical voxel codes; submap 0 is also checked against the oracle."""
    import torch
    dev = torch.device("cuda", 0)
    res = [0.05, 0.10, 0.20]
    S, rings, cols, steps = 4, 16, 625, 3   # 4 submaps: the batched insertion cuts large bins into 2048-record slices
    sets = {}
    for name in ("batch", "single"):
        sets[name] = [[hg.HybridGridTSDF(ctx, r, max_blocks=1 << 15) for r in res] for _ in range(S)]
    ins = [hg.TSDFRangeDataInserter3D() for _ in res]
    og = [po.Grid(r) for r in res]
    for j in range(S):
        for k in range(4):
            pose = synth.pose_k(7 * j + k)
            pts = synth.generate_scan(pose, rings, cols, stream=100 * j + k)
            for name in sets:
                hg.insert_pyramid(ins, hg.RangeData([0, 0, 0], pts), sets[name][j], pose_tq=pose.astype(np.float32))
            if j == 0:
                loc = synth.transform_points(pose, pts)
                for g in og:
                    g.insert(pose[:3], loc)
    pb = [hg.Problem(ctx) for _ in range(S)]
    ps = [hg.Problem(ctx) for _ in range(S)]
    scale = 1.0 / np.sqrt(float(rings * cols))
    for step in range(steps):
        scans, guesses = [], []
        for j in range(S):
            pose = synth.pose_k(7 * j + 4 + step)
            pts = synth.generate_scan(pose, rings, cols, stream=100 * j + 4 + step)
            scans.append((pts, torch.from_numpy(pts).to(dev)))
            guesses.append(synth.pose_mul(pose, synth.perturbation()))
        for j in range(S):
            for p, grids in ((pb[j], sets["batch"][j]), (ps[j], sets["single"][j])):
                p.reset()
                p.add_pose(guesses[j])
                p.add_block(scans[j][1], grids, scale, 0, multi_res=True)
        poses, summ = hg.register_scan_batch(pb, [0] * S, ins, [hg.RangeData([0, 0, 0], d) for _, d in scans],
                                             sets["batch"])
        for j in range(S):
            est, s1 = hg.register_scan(ps[j], 0, ins, hg.RangeData([0, 0, 0], scans[j][1]), sets["single"][j])
            np.testing.assert_allclose(poses[j], est, rtol=0, atol=1e-9)
            assert (summ[j].num_iterations, summ[j].termination_type, summ[j].termination_reason) == \
                   (s1.num_iterations, s1.termination_type, s1.termination_reason)
            assert np.array_equal(poses[j].astype(np.float32), est.astype(np.float32))
        # oracle: submap 0
        op = po.Problem()
        op.add_pose(guesses[0])
        op.add_block(scans[0][0], og, scale, 0, multi_res=True)
        so = op.solve()
        ref = op.get_pose(0)
        assert np.abs(ref - poses[0]).max() < 1e-9 and so.num_iterations == summ[0].num_iterations
        at = ref if np.array_equal(ref.astype(np.float32), poses[0].astype(np.float32)) else poses[0]
        loc = synth.transform_points(at, scans[0][0])
        for g in og:
            g.insert(at[:3].astype(np.float32), loc)
    ctx.synchronize()
    for j in range(S):
        for a, b in zip(sets["batch"][j], sets["single"][j]):
            a.status()
            for x, y in zip(a.export(), b.export()):
                assert np.array_equal(x, y)
    for o, g in zip(og, sets["batch"][0]):
        for x, y in zip(o.export(), g.export()):
            assert np.array_equal(x, y)


def test_register_scan_sequence_equals_step_by_step(hg, ctx):
    """hg_register_scan_sequence (the per-scan loop of a trajectory builder in one call) gives exactly what
    calling the steps one by one gives: the same poses bit for bit, the same summaries, the same maps."""
    import torch
    dev = torch.device("cuda", 0)
    res = [0.05, 0.10, 0.20]
    rings, cols, steps = 16, 625, 5
    sets = {name: [hg.HybridGridTSDF(ctx, r, max_blocks=1 << 15) for r in res] for name in ("seq", "loop")}
    ins = [hg.TSDFRangeDataInserter3D() for _ in res]
    for k in range(3):
        pose = synth.pose_k(k)
        pts = synth.generate_scan(pose, rings, cols, stream=k)
        for grids in sets.values():
            hg.insert_pyramid(ins, hg.RangeData([0, 0, 0], pts), grids, pose_tq=pose.astype(np.float32))
    scans, guesses = [], []
    for k in range(3, 3 + steps):
        pose = synth.pose_k(k)
        scans.append(hg.RangeData([0, 0, 0], torch.from_numpy(synth.generate_scan(pose, rings, cols, stream=k)).to(dev)))
        guesses.append(synth.pose_mul(pose, synth.perturbation()))
    scale = 1.0 / np.sqrt(float(rings * cols))
    ps, pl = hg.Problem(ctx), hg.Problem(ctx)
    poses, summ = hg.register_scan_sequence(ps, ins, scans, guesses, sets["seq"], scale, multi_res=True)
    for k in range(steps):
        pl.reset()
        pi = pl.add_pose(guesses[k])
        pl.add_block(scans[k].returns, sets["loop"], scale, pi, multi_res=True)
        est, s1 = hg.register_scan(pl, pi, ins, scans[k], sets["loop"])
        assert np.array_equal(poses[k], est)
        assert (summ[k].num_iterations, summ[k].num_successful_steps, summ[k].termination_type,
                summ[k].termination_reason, summ[k].final_cost) == \
               (s1.num_iterations, s1.num_successful_steps, s1.termination_type, s1.termination_reason, s1.final_cost)
    for a, b in zip(sets["seq"], sets["loop"]):
        a.status()
        for x, y in zip(a.export(), b.export()):
            assert np.array_equal(x, y)


def test_register_scan_sequence_with_host_scans_equals_device_scans(hg, ctx):
    """Scans handed over in HOST memory (what a drop-in receives from sensor::RangeData) travel through two
    device slots on a copy stream, scan k + 1 while step k runs: same poses bit for bit, same maps as the
    sequence over device-resident scans -- also for scans of different sizes (a slot grows) and more steps
    than slots."""
    import torch
    dev = torch.device("cuda", 0)
    res = [0.05, 0.10, 0.20]
    sets = {name: [hg.HybridGridTSDF(ctx, r, max_blocks=1 << 15) for r in res] for name in ("host", "device")}
    ins = [hg.TSDFRangeDataInserter3D() for _ in res]
    for k in range(3):
        pose = synth.pose_k(k)
        pts = synth.generate_scan(pose, 16, 625, stream=k)
        for grids in sets.values():
            hg.insert_pyramid(ins, hg.RangeData([0, 0, 0], pts), grids, pose_tq=pose.astype(np.float32))
    sizes = [(16, 400), (16, 625), (12, 500), (16, 700), (16, 625), (8, 300), (16, 625)]
    host, device, guesses, scale = [], [], [], []
    for j, (rings, cols) in enumerate(sizes):
        pose = synth.pose_k(3 + j)
        pts = synth.generate_scan(pose, rings, cols, stream=3 + j)
        host.append(hg.RangeData([0, 0, 0], pts))
        device.append(hg.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)))
        guesses.append(synth.pose_mul(pose, synth.perturbation()))
        scale.append(1.0 / np.sqrt(float(len(pts))))
    ph, pd = hg.Problem(ctx), hg.Problem(ctx)
    poses_h, summ_h = hg.register_scan_sequence(ph, ins, host, guesses, sets["host"], scale, multi_res=True)
    poses_d, summ_d = hg.register_scan_sequence(pd, ins, device, guesses, sets["device"], scale, multi_res=True)
    assert np.array_equal(poses_h, poses_d)
    assert [s.num_iterations for s in summ_h] == [s.num_iterations for s in summ_d]
    for a, b in zip(sets["host"], sets["device"]):
        a.status()
        for x, y in zip(a.export(), b.export()):
            assert np.array_equal(x, y)


def test_solve_batch_of_windows_equals_individual_solves(po, hg, ctx, maps):
    """hg_problem_solve_batch on GENERAL problems: sliding windows of different submaps share the launches of
    k_window_residuals / k_lm (grid row = problem). Every window keeps its own state: iterations and
    termination are those of solving it alone, poses agree to the rounding of the normal-equation sums (a
    window sizes its residual pass for its share of the chip: other workgroup partials, same terms). Windows of
    different sizes in one batch, one of them with a per-point-unwarped block; window 0 against the oracle."""
    og, gg = maps

    def build(api_problem, grids, first, n_cp, unwarped):
        pr = api_problem
        poses = [synth.pose_k(first + i) if i == 0 else synth.pose_mul(synth.pose_k(first + i), synth.perturbation())
                 for i in range(n_cp)]
        for i in range(n_cp):
            pr.add_pose(poses[i], i == 0)
            pr.set_velocity(i, np.array([0.5, 0.2, 0.0]), i == 0)
        for i in range(1, n_cp):
            k = first + i
            delta = synth.pose_mul(synth.pose_inverse(synth.pose_k(k)), synth.pose_k(k - 1))
            dq = synth.pose_mul(synth.pose_inverse(synth.pose_k(k - 1)), synth.pose_k(k))[3:]
            pr.add_odometry_block(i - 1, i, 12.0, 30.0, delta)
            pr.add_imu_block(i - 1, i, 3.0, 2.0, 70.0, 0.1, dq)
            pts = synth.generate_scan(synth.pose_k(k), 16, 120 + 7 * i, stream=1000 + 10 * first + i)
            s = 1.0 / np.sqrt(len(pts))
            if unwarped and i == 1 and hasattr(pr, "add_unwarped_block"):
                f = np.linspace(0.1, 0.9, len(pts))
                pr.add_unwarped_block(pts, f, [grids[0], grids[1], grids[2]], s, 0, 1, multi_res=True)
            elif unwarped and i == 1:
                continue
            else:
                pr.add_block(pts, [grids[0], grids[1], grids[2]], s, i, -1, 0.0, True)
        return pr

    shapes = [(2, 6, False), (3, 4, False), (1, 7, True), (4, 3, False), (2, 5, False)]
    single = [build(hg.Problem(ctx), gg, f, n, u) for f, n, u in shapes]
    s_single = [p.solve() for p in single]
    batch = [build(hg.Problem(ctx), gg, f, n, u) for f, n, u in shapes]
    s_batch = hg.solve_batch(batch)
    for (f, n, u), a, b, sa, sb in zip(shapes, single, batch, s_single, s_batch):
        assert (sa.num_iterations, sa.termination_type, sa.termination_reason, sa.num_successful_steps) == \
               (sb.num_iterations, sb.termination_type, sb.termination_reason, sb.num_successful_steps)
        assert sa.num_iterations > 1
        for i in range(n):
            np.testing.assert_allclose(b.get_pose(i), a.get_pose(i), rtol=0, atol=1e-9)
            np.testing.assert_allclose(b.get_velocity(i), a.get_velocity(i), rtol=0, atol=1e-8)
        assert abs(sa.final_cost - sb.final_cost) <= 1e-10 * max(1.0, abs(sa.final_cost))
    op = build(po.Problem(), og, *shapes[0])
    so = op.solve()
    assert so.num_iterations == s_batch[0].num_iterations and so.termination_reason == s_batch[0].termination_reason
    for i in range(shapes[0][1]):
        a, b = op.get_pose(i), batch[0].get_pose(i)
        assert np.linalg.norm(a[:3] - b[:3]) < POSE_TOL_M and rot_angle(a[3:], b[3:]) < POSE_TOL_RAD
    for p in single + batch:
        p.close()


@pytest.mark.parametrize("shape", ["single", "two_pose", "unwarped"])
def test_structured_lane_order_is_a_permutation_of_the_sums(po, hg, ctx, maps, shape):
    """hg_problem_set_block_width lets adjacent lanes take horizontally adjacent returns of a structured scan
    (four columns at a time): the residuals keep their positions and values, the normal equations are the same
    sums in another order -- evaluation within rounding, solve with the same iterations and poses within 1e-9;
    a return count that is not a multiple of the width keeps the plain order."""
    og, gg = maps
    rings, cols = 16, 203  # 203 columns: 50 groups of four + 3 columns that keep their order
    pts = synth.generate_scan(synth.pose_k(5), rings, cols, stream=77)
    guess = synth.pose_mul(synth.pose_k(5), synth.perturbation())

    def build(width):
        pr = hg.Problem(ctx)
        if shape == "single":
            i = pr.add_pose(guess)
            pr.add_block(pts, gg, 1.0 / np.sqrt(len(pts)), i, multi_res=True, width=width)
        else:
            a = pr.add_pose(synth.pose_k(4), True)
            b = pr.add_pose(guess)
            if shape == "two_pose":
                pr.add_block(pts, gg, 1.0 / np.sqrt(len(pts)), a, b, 0.7, multi_res=True, width=width)
            else:
                k = pr.add_unwarped_block(pts, np.linspace(0.2, 1.0, len(pts)), gg, 1.0 / np.sqrt(len(pts)), a, b, multi_res=True)
                pr.set_block_width(k, width)
        return pr

    plain, structured = build(0), build(rings)
    c0, r0, g0, H0 = plain.evaluate()
    c1, r1, g1, H1 = structured.evaluate()
    np.testing.assert_array_equal(r1, r0)  # per-return values do not depend on which lane computes them
    assert abs(c0 - c1) <= 1e-13 * max(1.0, abs(c0))
    np.testing.assert_allclose(g1, g0, rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(H1, H0, rtol=1e-11, atol=1e-14)
    s0, s1 = plain.solve(), structured.solve()
    assert (s0.num_iterations, s0.termination_reason) == (s1.num_iterations, s1.termination_reason)
    last = 0 if shape == "single" else 1
    np.testing.assert_allclose(structured.get_pose(last), plain.get_pose(last), rtol=0, atol=1e-9)
    odd = hg.Problem(ctx)
    i = odd.add_pose(guess)
    odd.add_block(pts[:-5], gg, 1.0, i, multi_res=True, width=rings)  # not a multiple: plain order, still correct
    ref = hg.Problem(ctx)
    ref.add_pose(guess)
    ref.add_block(pts[:-5], gg, 1.0, 0, multi_res=True)
    np.testing.assert_array_equal(odd.evaluate()[2], ref.evaluate()[2])
    for pr in (plain, structured, odd, ref):
        pr.close()
