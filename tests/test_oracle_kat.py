"""Pins the CPU oracle against the reference's own known-answer tests.

Each test names the reference test it restates (paths relative to /root/reference/cartographer/).
"""
import numpy as np
import pytest


def test_tsd_value_converter_defaults_and_roundtrips(po):
    """mapping/2d/tsd_value_converter_test.cc:38-84 (tau = 0.1, max weight 10)."""
    c = po.Converter(0.1, 10.0)
    for i in range(1, 32768):
        assert c.tsd_to_value(c.value_to_tsd(i)) == i
        assert c.tsd_to_value(c.value_to_tsd(i + 32768)) == i          # update marker ignored
        assert c.weight_to_value(c.value_to_weight(i)) == i
        assert c.weight_to_value(c.value_to_weight(i + 32768)) == i


def test_tsd_value_converter_tolerances_and_clamping(po):
    """tsd_value_converter_test.cc:86-122."""
    tau, wmax = np.float32(0.1), np.float32(10.0)
    c = po.Converter(tau, wmax)
    tol_t = tau * np.float32(2.0) / np.float32(32767.0)
    tol_w = wmax / np.float32(32767.0)
    for i in range(1000):
        s = -tau + np.float32(i) * np.float32(2.0) * tau / np.float32(1000)
        assert abs(c.value_to_tsd(c.tsd_to_value(s)) - s) <= tol_t
        w = np.float32(i) * wmax / np.float32(1000)
        assert abs(c.value_to_weight(c.weight_to_value(w)) - w) <= tol_w
    assert abs(c.value_to_weight(c.weight_to_value(2 * wmax)) - wmax) <= tol_w
    assert abs(c.value_to_weight(c.weight_to_value(-wmax)) - 0.0) <= tol_w
    assert abs(c.value_to_tsd(c.tsd_to_value(2 * tau)) - tau) <= tol_t
    assert abs(c.value_to_tsd(c.tsd_to_value(-2 * tau)) + tau) <= tol_t


def test_value_conversion_table_formula_is_exact(po):
    """mapping/value_conversion_tables_test.cc:45-66: LUT[i] == i*scale + (lb - scale) exactly,
    LUT[0] == unknown. (The reference draws bounds from std::mt19937; any bounds pin the formula.)"""
    rng = np.random.default_rng(42)
    for _ in range(100):
        a, b, u = rng.uniform(-10, 10, 3).astype(np.float32)
        lb, ub = min(a, b), max(a, b)
        t = po.conversion_table(u, lb, ub)
        assert t[0] == u and t[32768] == u
        scale = (ub - lb) / np.float32(32766.0)
        i = np.arange(1, 32768, dtype=np.int32).astype(np.float32)
        expect = i * scale + (lb - scale)
        assert np.array_equal(t[1:32768], expect)
        assert np.array_equal(t[32769:], expect)                     # marker bit masked


def test_hybrid_grid_tsdf_apply(po):
    """mapping/3d/hybrid_grid_tsdf_test.cc:31-53."""
    g = po.Grid(1.0, 0.5, 1.0)
    cells = [[0, 0, 0], [0, 1, 0], [1, 0, 0], [1, 1, 0], [0, 0, 1], [0, 1, 1], [1, 0, 1], [1, 1, 1]]
    assert not g.get_float(cells)[2].any()
    g.set_cell([1, 0, 1], 0.1, 0.5)
    t, w, k = g.get_float([[1, 0, 1], [0, 0, 1]])
    assert k[0] and abs(t[0] - 0.1) < 1e-4 and abs(w[0] - 0.5) < 1e-4
    assert abs(t[1] + 0.5) < 1e-4 and abs(w[1]) < 1e-4


def test_get_cell_index_rounding(po):
    """mapping/3d/hybrid_grid_test.cc:91-125 (resolution 2)."""
    g = po.Grid(2.0)
    pts = [[0, 0, 0], [0, 26, 10], [14, 0, 10], [14, 26, 0], [8.5, 11.5, 0.5], [7.5, 12.5, 1.5],
           [6.5, 14.5, 2.5], [5.5, 13.5, 3.5]]
    want = [[0, 0, 0], [0, 13, 5], [7, 0, 5], [7, 13, 0], [4, 6, 0], [4, 6, 1], [3, 7, 1], [3, 7, 2]]
    assert g.cell_index(pts).tolist() == want
    c = g.center_of_cell([[3, 2, 1]])
    np.testing.assert_allclose(c, [[6, 4, 2]], atol=1e-6)
    assert g.cell_index(c).tolist() == [[3, 2, 1]]
    # exact .5 ties round half away from zero (std::lround, common/port.h:40)
    assert g.cell_index([[1.0, -1.0, 3.0]]).tolist() == [[1, -1, 2]]


def test_iteration_covers_random_cells_in_tree_order(po):
    """hybrid_grid_test.cc:127-184: iteration visits exactly the set cells; order = meta cell,
    leaf, voxel, each z-major."""
    rng = np.random.default_rng(1285120005)
    ijk = np.unique(rng.integers(-3000, 3000, (10000, 3)).astype(np.int32), axis=0)
    g = po.Grid(2.0, 2.5, 100.0)
    for c in ijk:
        g.set_cell(c, 0.3, 7.0)
    e_ijk, e_t, e_w = g.export()
    assert len(e_ijk) == len(ijk)
    assert set(map(tuple, e_ijk.tolist())) == set(map(tuple, ijk.tolist()))
    s = e_ijk.astype(np.int64) + 8192
    key = ((((((s[:, 2] >> 6) << 8 | (s[:, 1] >> 6)) << 8 | (s[:, 0] >> 6)) << 3 | ((s[:, 2] >> 3) & 7)) << 3
             | ((s[:, 1] >> 3) & 7)) << 3 | ((s[:, 0] >> 3) & 7)) << 9 | ((s[:, 2] & 7) << 6) | ((s[:, 1] & 7) << 3) | (s[:, 0] & 7)
    assert np.all(np.diff(key) > 0)


def test_interpolate_transform(po):
    """transform/timestamped_transform_test.cc:37-88: lerp + slerp, 21 factors, 1e-6."""
    t0 = np.array([0, 0, 0, 1, 0, 0, 0], float)
    t1 = np.array([10, 10, 10, np.cos(1.0), 0, 0, np.sin(1.0)], float)   # Rz(2 rad)
    np.testing.assert_allclose(po.interpolate_transform(t0, t1, 0.0), t0, atol=1e-6)
    np.testing.assert_allclose(po.interpolate_transform(t0, t1, 1.0), t1, atol=1e-6)
    for i in range(21):
        f = i / 20.0
        want = np.array([10 * f, 10 * f, 10 * f, np.cos(f), 0, 0, np.sin(f)])
        np.testing.assert_allclose(po.interpolate_transform(t0, t1, f), want, atol=1e-6)


def test_ka1_insert_hit_axis_aligned(po):
    """SURVEY Appendix B KA-1 (derived from tsdf_range_data_inserter_3d.cc:294-342,725-737)."""
    g = po.Grid(0.1)
    assert g.insert([0, 0, 0], [[1, 0, 0]]) == (1, 6)
    ijk = np.array([[x, 0, 0] for x in range(7, 15)], np.int32)
    t, w = g.read_cells(ijk)
    assert t.tolist() == [0, 62258, 55705, 49152, 42599, 36046, 32769, 0]
    assert w.tolist() == [0, 34, 34, 34, 34, 34, 34, 0]
    tf, wf, _ = g.get_float(ijk[1:7])
    np.testing.assert_allclose(tf, [0.1999939, 0.0999970, 0.0, -0.0999970, -0.1999939, -0.25], atol=1e-7)
    np.testing.assert_allclose(wf, 1.0071415, rtol=1e-7)
    g.insert([0, 0, 0], [[1, 0, 0]])
    t2, w2 = g.read_cells(ijk)
    assert t2.tolist() == t.tolist() and w2[1:7].tolist() == [67] * 6
    np.testing.assert_allclose(g.get_float(ijk[1:2])[1], 2.0142829, rtol=1e-7)


def test_ka2_ka3_unknown_and_empty_grid(po):
    """KA-2: untouched cell reads (-tau, 0, unknown). KA-3: lookups on an empty grid return -tau
    with zero gradient (interpolated_tsdf.h:86-89); an empty-cloud block converges at iteration 0
    (interpolated_tsdf_space_cost_function_3d_test.cc:236-240: summary.iterations.size() == 1)."""
    g = po.Grid(0.1)
    t, w, k = g.get_float([[5, -7, 2]])
    assert t[0] == np.float32(-0.25) and w[0] == 0 and not k[0]
    val, grad = po.interp_tsd([g], [[0.31, -0.2, 0.77]])
    assert val[0] == np.float64(np.float32(-0.25)) and np.all(grad == 0)
    pr = po.Problem()
    pose = [0.1, 0.2, 0.3, 1, 0, 0, 0]
    pr.add_pose(pose)
    pr.add_block(np.zeros((0, 3), np.float32), [g], 1.0, 0)
    s = pr.solve()
    assert s.num_iterations <= 1 and s.termination_type == 0
    np.testing.assert_array_equal(pr.get_pose(0), pose)


def test_ka4_branch_constants(po):
    """KA-4: single-res lookup mixes the hard-coded -0.3 (interpolated_tsdf.h:34); the multi-res
    lookup skips a level with any invalid corner (interpolated_multi_resolution_tsdf.h:99-106)."""
    g = po.Grid(0.1)
    # only the 4 voxels with x index 0 valid (tsd 0.1), the x=1 face unknown
    for y in (0, 1):
        for z in (0, 1):
            g.set_cell([0, y, z], 0.1, 5.0)
    p = [[0.03, 0.04, 0.05]]
    val, grad = po.interp_tsd([g], p)
    t = g.get_float([[0, 0, 0]])[0][0]
    assert abs(val[0] - float(t)) < 1e-12 and np.allclose(grad, 0)      # copy-valid-side branch
    g2 = po.Grid(0.1)
    g2.set_cell([0, 0, 0], 0.1, 5.0)
    g2.set_cell([1, 1, 0], 0.2, 5.0)
    val2, _ = po.interp_tsd([g2], p)
    # z-pairs (0,0,*): copy; (0,1,*): both invalid -> -0.3; lerp in y mixes -0.3? no: w12 == 0 -> copy
    # x-pair: q1 from (0,*,*), q2 from (1,*,*) both valid -> lerp; result lies between the two values
    assert 0.09 < val2[0] < 0.21
    coarse = po.Grid(0.2)
    for x in (-1, 0, 1, 2):
        for y in (-1, 0, 1, 2):
            for z in (-1, 0, 1, 2):
                coarse.set_cell([x, y, z], 0.05, 3.0)
    vm, _ = po.interp_tsd([g, coarse], p, multi_res=True)
    assert abs(vm[0] - float(coarse.get_float([[0, 0, 0]])[0][0])) < 1e-9   # fine level skipped
    ve, _ = po.interp_tsd([g, po.Grid(0.2)], p, multi_res=True)
    assert ve[0] == np.float64(np.float32(-0.25))                          # none valid: finest -tau


def test_insertion_ratio_recurrence(po):
    """tsdf_range_data_inserter_3d.cc:703-710: deterministic decimation, first point always kept."""
    g = po.Grid(0.2)
    pts = np.tile(np.array([[2.0, 0.1, 0.2]], np.float32), (100, 1))
    n_in, _ = g.insert([0, 0, 0], pts, po.InsertOpts(insertion_ratio=0.1))
    assert n_in == 10


def test_jet_jacobian_matches_finite_differences(po):
    from hectorgrapher_amd import synth
    g = po.Grid(0.1)
    for k in range(3):
        pose = synth.pose_k(k)
        g.insert(pose[:3], synth.transform_points(pose, synth.generate_scan(pose, 8, 200, stream=k)))
    pose = synth.pose_k(3)
    pts = synth.generate_scan(pose, 8, 200, stream=3)
    pr = po.Problem()
    pr.add_pose(synth.pose_k(2))
    pr.add_pose(pose)
    pr.add_block(pts, [g], 0.05, 0, 1, 0.6)
    c, r, J, grad = pr.evaluate()
    np.testing.assert_allclose(grad, J.T @ r, rtol=1e-12, atol=1e-15)
    eps = 1e-7
    for col in range(12):
        i, k = divmod(col, 6)
        base = pr.get_pose(i)
        d = np.zeros(6)
        d[k] = eps
        tq = base.copy()
        tq[:3] += d[:3]
        tq[3:] = po.quaternion_plus(base[3:], d[3:])
        pr.set_pose(i, tq)
        _, r2, _, _ = pr.evaluate(False)
        pr.set_pose(i, base)
        err = np.abs((r2 - r) / eps - J[:, col])
        assert np.median(err) < 1e-5 and (err > 1e-3).mean() < 0.02      # isolated branch flips only


def test_voxel_filter_reference_kats(po):
    """sensor/internal/voxel_filter_test.cc:30-56."""
    pc = np.array([[0, 0, 0], [0.1, -0.1, 0.1], [0.3, -0.1, 0], [0, 0, 0.1]], np.float32)
    assert po.voxel_filter(0.3, pc).tolist() == [0, 2]                         # first point per voxel
    pc = np.array([[100000., 0, 0], [100000.001, -0.0001, 0.0001], [100000.003, -0.0001, 0],
                   [-200000., 0, 0]], np.float32)
    assert po.voxel_filter(0.01, pc).tolist() == [0, 3]                        # large coordinates
    timed = np.array([[-100.0, 0.3, 0.4, float(i)] for i in range(100)], np.float32)
    assert po.voxel_filter(0.3, timed).tolist() == [0]                         # ignores time


def test_adaptive_voxel_filter_properties(po):
    """adaptive_voxel_filter.h:46-86: sparse clouds pass through, dense ones keep >= min points."""
    from hectorgrapher_amd import synth
    pts = synth.generate_scan(synth.pose_k(0), 16, 625)
    keep = po.adaptive_voxel_filter(2.0, 150, 15.0, pts)
    r = np.linalg.norm(pts[keep], axis=1)
    assert 150 <= len(keep) < 400 and np.all(r <= 15.0) and np.all(np.diff(keep.astype(np.int64)) > 0)
    few = pts[:100]
    assert po.adaptive_voxel_filter(2.0, 150, 60.0, few).tolist() == list(range(100))


def test_odometry_and_imu_blocks_jacobian_and_minimum(po):
    """oltb.cc:928-1074 blocks restated with Jets: Jacobian vs finite differences; a window tied only
    by odometry + IMU blocks converges to the poses / velocities those measurements imply."""
    from hectorgrapher_amd import synth
    pr = po.Problem()
    p0 = synth.pose_k(3)
    p1 = synth.pose_mul(synth.pose_k(4), synth.perturbation())
    pr.add_pose(p0, True)
    pr.add_pose(p1)
    pr.set_velocity(0, [0.5, 0.2, 0.0], True)
    pr.set_velocity(1, [0.4, 0.3, 0.1], False)
    delta = synth.pose_mul(synth.pose_inverse(synth.pose_k(4)), synth.pose_k(3))
    dq = synth.pose_mul(synth.pose_inverse(synth.pose_k(3)), synth.pose_k(4))[3:]
    pr.add_odometry_block(0, 1, 10.0, 5.0, delta)
    pr.add_imu_block(0, 1, 3.0, 2.0, 7.0, 0.1, dq)
    c, r, J, g = pr.evaluate()
    assert J.shape == (15, 9)
    eps = 1e-7
    x1, v1 = pr.get_pose(1), pr.get_velocity(1)
    for col in range(9):
        d = np.zeros(9)
        d[col] = eps
        tq = x1.copy()
        tq[:3] += d[:3]
        tq[3:] = po.quaternion_plus(x1[3:], d[3:6])
        pr.set_pose(1, tq)
        pr.set_velocity(1, v1 + d[6:9])
        _, r2, _, _ = pr.evaluate(False)
        pr.set_pose(1, x1)
        pr.set_velocity(1, v1)
        assert np.abs((r2 - r) / eps - J[:, col]).max() < 1e-5
    s = pr.solve()
    assert s.termination_type == 0 and s.final_cost < 1e-12
    np.testing.assert_allclose(pr.get_pose(1), synth.pose_k(4), atol=1e-6)
    np.testing.assert_allclose(pr.get_velocity(1), [0.5, 0.2, 0.0], atol=1e-6)


def test_xray_texture_known_answers(po):
    """submap_3d.cc:80-105,142-214 on hand-built columns (identity pose, r = 0.1 m, tau = 0.25 m).
    tsd = 0 -> probability 1 - 0/tau = 1, stored as ProbabilityToValue(clamp 0.9) = 32767, read back
    as 0.9. Expected bytes derived by hand:
      full column z = 0..4:  count 5, z_difference 4, no free space, mean 0.9 -> log-odds integer
                             255 -> delta -127 -> (value 0, alpha 127)
      gapped column z = 0,4: count 2, free space 2 * 0.15, mean (1.8 + 0.1 * 0.3) / 2.3 = 0.79565
                             -> RoundToInt((1.35949 + 2.19722) * 254 / 4.39445) + 1 = 207 -> (0, 79)
      short column z = 0..2: z_difference 2 < 3 -> (0, 0)
      far voxel |tsd| = 0.2: probability 0.2 < 0.501, ignored (would otherwise widen the box)"""
    g = po.Grid(0.1)
    for z in range(5):
        g.set_cell([0, 0, z], 0.0, 1.0)
    for z in (0, 4):
        g.set_cell([2, 1, z], 0.0, 1.0)
    for z in range(3):
        g.set_cell([1, 3, z], 0.0, 1.0)
    g.set_cell([9, 9, 0], 0.2, 1.0)
    cells, mx = g.xray(np.array([0, 0, 0, 1, 0, 0, 0.0]))
    assert mx.tolist() == [2, 3]
    assert cells.shape == (3, 4, 2)                 # height = x extent 0..2, width = y extent 0..3
    want = np.zeros((3, 4, 2), np.uint8)
    want[2 - 0, 3 - 0] = (0, 127)                   # pixel (max_x - x, max_y - y)
    want[2 - 2, 3 - 1] = (0, 79)
    assert np.array_equal(cells, want)
    # a free-space-dominated column turns into "value": two voxels at probability 0.52 spanning z 0..30
    g2 = po.Grid(0.1)
    for z in (0, 30):
        g2.set_cell([0, 0, z], 0.12, 1.0)           # probability 1 - 0.12 / 0.25 = 0.52
    cells2, _ = g2.xray(np.array([0, 0, 0, 1, 0, 0, 0.0]))
    # count 2, free space 28 * 0.15 = 4.2, max_probability 0.52 -> mean (1.04 + 0.48 * 4.2) / 6.2 = 0.4929
    # -> logit -0.0284 -> RoundToInt(2.16883 * 57.8002) + 1 = 126 -> delta 2 -> (value 2, alpha 0)
    assert cells2.shape == (1, 1, 2) and cells2[0, 0].tolist() == [2, 0]


def test_lm_reaches_the_minimum_an_independent_solver_finds(po):
    """The restated Ceres loop against scipy.optimize.least_squares (trust-region reflective, its own
    finite-difference Jacobian, no shared code with the oracle's solver): same local minimum of the
    same residual function from the same start."""
    scipy_opt = pytest.importorskip("scipy.optimize")
    from hectorgrapher_amd import synth
    g = po.Grid(0.1)
    for k in range(4):
        pose = synth.pose_k(k)
        g.insert(pose[:3], synth.transform_points(pose, synth.generate_scan(pose, 16, 400, stream=k)))
    pose = synth.pose_k(4)
    pts = synth.generate_scan(pose, 16, 400, stream=4)
    start = synth.pose_mul(pose, synth.perturbation())
    pr = po.Problem()
    pr.add_pose(start)
    pr.add_block(pts, [g], 1.0 / np.sqrt(len(pts)), 0)
    summ = pr.solve(max_num_iterations=50, function_tolerance=1e-12, gradient_tolerance=1e-14,
                    parameter_tolerance=1e-12)
    solved = pr.get_pose(0)

    ev = po.Problem()
    ev.add_pose(start)
    ev.add_block(pts, [g], 1.0 / np.sqrt(len(pts)), 0)

    def residuals(x):  # x: translation + rotation-vector increment on the start pose
        tq = start.copy()
        tq[:3] = start[:3] + x[:3]
        tq[3:] = po.quaternion_plus(start[3:], x[3:])
        ev.set_pose(0, tq)
        return ev.evaluate(False)[1].copy()

    ref = scipy_opt.least_squares(residuals, np.zeros(6), method="trf", xtol=1e-12, ftol=1e-12, gtol=1e-12,
                                  diff_step=1e-6)
    assert ref.cost <= summ.initial_cost
    # the cost is piecewise smooth (validity branches of the interpolation flip between voxels), so two
    # optimisers stop a few branch cells apart: both remove the same share of the cost, within 1 % of
    # the initial cost, at poses within 3 mm / 3 mrad
    tq = start.copy()
    tq[:3] = start[:3] + ref.x[:3]
    tq[3:] = po.quaternion_plus(start[3:], ref.x[3:])
    print("cost oracle %.6e scipy %.6e initial %.6e; dt %.2e dq %.2e" % (
        summ.final_cost, ref.cost, summ.initial_cost, np.abs(tq[:3] - solved[:3]).max(),
        min(np.abs(tq[3:] - solved[3:]).max(), np.abs(tq[3:] + solved[3:]).max())))
    assert summ.final_cost < 0.7 * summ.initial_cost
    assert abs(summ.final_cost - ref.cost) <= 1e-2 * summ.initial_cost
    assert np.abs(tq[:3] - solved[:3]).max() < 3e-3
    assert min(np.abs(tq[3:] - solved[3:]).max(), np.abs(tq[3:] + solved[3:]).max()) < 3e-3


def test_oracle_under_ubsan():
    """Sanitizers run on the CPU build only: the oracle restatement compiled with -fsanitize=undefined
    (-fno-sanitize-recover: any finding aborts) runs this file's known-answer tests, the ray-walk identities and a
    small insert + two-pose solve -- signed overflow, bad shifts, out-of-range float-to-int conversions, misaligned
    or null accesses and array bounds in the codec, the tree, the ray walk, the Jet arithmetic and the solver would
    end the subprocess with a sanitizer report."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HG_ORACLE_SANITIZE="1")
    code = r"""
import sys, os
sys.path.insert(0, os.path.join(%r, "oracle")); sys.path.insert(0, %r)
import numpy as np
import pyoracle as po
assert po.lib()._name.endswith("libhg_oracle_ubsan.so")
from hectorgrapher_amd import synth
grids = [po.Grid(r) for r in (0.05, 0.10, 0.20)]
for k in range(3):
    pose = synth.pose_k(k)
    pts = synth.generate_scan(pose, 8, 128, stream=k)
    loc = synth.transform_points(pose, pts)
    for g in grids:
        g.insert(pose[:3], loc, width=8)
pr = po.Problem()
a = pr.add_pose(synth.pose_k(2), True)
b = pr.add_pose(synth.pose_mul(synth.pose_k(3), synth.perturbation()))
pts = synth.generate_scan(synth.pose_k(3), 8, 128, stream=3)
pr.add_block(pts, grids, 1.0 / np.sqrt(len(pts)), b, multi_res=True)
pr.add_block(pts, [grids[1]], 0.5 / np.sqrt(len(pts)), a, b, 0.7)
pr.set_velocity(a, [0, 0, 0], True); pr.set_velocity(b, [0.1, 0, 0])
pr.add_odometry_block(a, b, 2.0, 3.0, synth.pose_mul(synth.pose_inverse(synth.pose_k(3)), synth.pose_k(2)))
pr.add_imu_block(a, b, 1.0, 1.0, 1.0, 0.1, [1.0, 0.0, 0.0, 0.0])
s = pr.solve()
assert s.num_iterations >= 1
ijk, t, w = grids[0].export()
assert len(t) > 1000
ct = np.array([0, 1000000, 2000000], np.int64)
cp = np.array([synth.pose_k(0), synth.pose_k(1), synth.pose_k(2)])
tp = np.concatenate([pts, np.linspace(0.0, 0.19, len(pts), dtype=np.float32)[:, None]], 1)
xyz, org, ok = po.unwarp_range_data(ct, cp, [(0, [0, 0, 0], tp)])
assert ok
print("ubsan ok", s.num_iterations, len(t))
""" % (root, root)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ubsan ok" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])
    assert "runtime error" not in out.stderr, out.stderr[-3000:]
    # and the reference's known-answer vectors through the sanitized build
    kat = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_oracle_kat.py"), "-q", "-x",
                          "-k", "not ubsan", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    assert kat.returncode == 0, (kat.stdout[-1500:], kat.stderr[-1500:])
    assert "runtime error" not in kat.stderr
