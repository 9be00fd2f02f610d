"""Full-size parity gate for the headline configuration (BASELINE.json configs[1]): a 100 000-point
scan (50 rings x 2000 columns) registered into the 3-resolution TSDF (0.05 / 0.10 / 0.20 m) --
multi-resolution LM match from the bench's perturbed guess, then exact insertion at the solved
pose -- HIP path (through hg_register_scan, what bench.py times) against the CPU oracle:
pose within 1e-4 m / 1e-4 rad (north_star tolerance) with identical iteration counts and
termination, then every voxel code of the three grids identical."""
import numpy as np
import pytest

import bench
from hectorgrapher_amd import synth

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4
POSE_TOL_RAD = 1e-4
RINGS, COLS = 50, 2000
MAP_SCANS = 10


def rot_angle(qa, qb):
    return 2.0 * np.arccos(min(1.0, abs(float(np.dot(qa, qb)))))


def test_register_100k_point_scans_three_resolutions(po, hg, ctx):
    import torch
    dev = torch.device("cuda", 0)
    og = [po.Grid(r) for r in bench.RESOLUTIONS]
    gg = [hg.HybridGridTSDF(ctx, r, max_blocks=1 << 18) for r in bench.RESOLUTIONS]
    ins = [hg.TSDFRangeDataInserter3D() for _ in gg]
    for pose, pts in bench.make_scans(RINGS, COLS, 0, MAP_SCANS, 0):
        loc = synth.transform_points(pose, pts)
        for g in og:
            g.insert(pose[:3], loc)
        hg.insert_pyramid(ins, hg.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), gg,
                          pose_tq=pose.astype(np.float32))
    n_pts = RINGS * COLS
    scale = 1.0 / np.sqrt(float(n_pts))
    problem = hg.Problem(ctx)
    # consecutive steps, as the bench runs them: scan k is matched against the map that holds scan k - 1
    for pose, pts in bench.make_scans(RINGS, COLS, MAP_SCANS, 3, 0):
        assert len(pts) == n_pts
        guess = synth.pose_mul(pose, synth.perturbation())
        d = torch.from_numpy(pts).to(dev)
        problem.reset()
        pi = problem.add_pose(guess)
        problem.add_block(d, gg, scale, pi, multi_res=True)
        est, sg = hg.register_scan(problem, pi, ins, hg.RangeData([0, 0, 0], d), gg)
        op = po.Problem()
        oi = op.add_pose(guess)
        op.add_block(pts, og, scale, oi, multi_res=True)
        so = op.solve()
        ref = op.get_pose(oi)
        assert np.linalg.norm(est[:3] - ref[:3]) < POSE_TOL_M
        assert rot_angle(est[3:], ref[3:]) < POSE_TOL_RAD
        assert (sg.num_iterations, sg.num_successful_steps, sg.termination_type, sg.termination_reason) == \
               (so.num_iterations, so.num_successful_steps, so.termination_type, so.termination_reason)
        assert abs(sg.final_cost - so.final_cost) <= 1e-9 * max(1.0, abs(so.final_cost))
        # Submap3D::InsertData at optimized_pose.cast<float>(): the oracle inserts at ITS solved pose;
        # the maps stay bit-identical as long as both float casts agree (they do unless a component
        # sits within 1e-12 of a float rounding boundary, in which case the oracle follows the device)
        at = ref if np.array_equal(ref.astype(np.float32), est.astype(np.float32)) else est
        loc = synth.transform_points(at, pts)
        for g in og:
            g.insert(at[:3].astype(np.float32), loc)
    for o, g in zip(og, gg):
        g.status()  # raises on sticky capacity / range flags
        eo, eg = o.export(), g.export()
        assert len(eo[1]) > 20000
        for x, y in zip(eo, eg):
            assert np.array_equal(x, y)


def test_stream_of_100k_point_scans_equals_sequential_reference(po, hg, ctx):
    """BASELINE.json configs[2]: a stream of 100 000-point scans handed over in ONE batched call
    (hg_pyramid_insert_batch) equals the oracle inserting the scans one after the other: every voxel
    code of the three grids, and the counters."""
    import ctypes as C
    import torch
    from hectorgrapher_amd import _lib
    dev = torch.device("cuda", 0)
    B = 12
    scans = bench.make_scans(RINGS, COLS, 50, B, 0)   # around the heavy end of the trajectory
    og = [po.Grid(r) for r in bench.RESOLUTIONS]
    ref = [[0, 0] for _ in og]
    for pose, pts in scans:
        loc = synth.transform_points(pose, pts)
        for g, acc in zip(og, ref):
            n_in, u = g.insert(pose[:3].astype(np.float32), loc)
            acc[0] += n_in
            acc[1] += u
    gg = [hg.HybridGridTSDF(ctx, r, max_blocks=1 << 16) for r in bench.RESOLUTIONS]
    xyz = torch.from_numpy(np.concatenate([p for _, p in scans])).to(dev)
    poses = np.array([pose for pose, _ in scans], np.float32)
    origins = np.zeros((B, 3), np.float32)
    offs = np.arange(B + 1, dtype=np.uint64) * (RINGS * COLS)
    L = _lib.load()
    garr = (C.c_void_p * 3)(*[g._h for g in gg])
    opts = (hg.InsertOpts * 3)(*[hg.InsertOpts() for _ in gg])
    st = (hg.InsertStats * 3)()
    hg.check(L.hg_pyramid_insert_batch(garr, opts, 3, origins.ctypes.data_as(C.c_void_p), xyz.data_ptr(),
                                       offs.ctypes.data_as(C.c_void_p), B, 0, poses.ctypes.data_as(C.c_void_p),
                                       _lib.HG_INSERT_EXACT, 1, st), "hg_pyramid_insert_batch")
    for o, g, s, r in zip(og, gg, st, ref):
        assert (s.num_hits, s.num_updates) == tuple(r)
        for x, y in zip(o.export(), g.export()):
            assert np.array_equal(x, y)


def test_stream_of_small_scans_shares_passes(po, hg, ctx):
    """Small scans of one batched call are binned together (up to 2^17 returns per pass, seq = return
    index * 8 + sample keeps the reference's update order across the scans of a pass): 40 scans of
    10 000 points = 4 passes, bit-exact against the oracle inserting them one after the other."""
    B, rings, cols = 40, 16, 625
    og = [po.Grid(r) for r in bench.RESOLUTIONS]
    gg = [hg.HybridGridTSDF(ctx, r, max_blocks=1 << 16) for r in bench.RESOLUTIONS]
    origins, clouds, offs = [], [], [0]
    for k in range(B):
        pose = synth.pose_k(k)
        loc = synth.transform_points(pose, synth.generate_scan(pose, rings, cols, stream=k))
        origins.append(pose[:3])
        clouds.append(loc)
        offs.append(offs[-1] + len(loc))
        for g in og:
            g.insert(pose[:3], loc)
    ins = hg.TSDFRangeDataInserter3D()
    for g in gg:
        ins.InsertBatch(np.array(origins), np.concatenate(clouds), offs, g)
    for o, g in zip(og, gg):
        for x, y in zip(o.export(), g.export()):
            assert np.array_equal(x, y)


def test_window_of_nine_100k_point_scans_against_oracle(po, hg, ctx):
    """The OptimizingLocalTrajectoryBuilder shape at the size bench.py --workload window quotes: ten
    control points (81 free columns), nine 100 000-point multi-resolution scan blocks, IMU + odometry
    blocks -- k_window_residuals (lean tiles, per-block tails) + k_lm (block-tridiagonal solve) against the
    oracle: every control point within 1e-4 m / 1e-4 rad, same iterations and termination."""
    import torch
    dev = torch.device("cuda", 0)
    n_cp = 10
    og = [po.Grid(r) for r in bench.RESOLUTIONS]
    gg = [hg.HybridGridTSDF(ctx, r, max_blocks=1 << 18) for r in bench.RESOLUTIONS]
    ins = [hg.TSDFRangeDataInserter3D() for _ in gg]
    for pose, pts in bench.make_scans(RINGS, COLS, 0, MAP_SCANS, 0):
        loc = synth.transform_points(pose, pts)
        for g in og:
            g.insert(pose[:3], loc)
        hg.insert_pyramid(ins, hg.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), gg,
                          pose_tq=pose.astype(np.float32))
    for g in gg:
        assert g.window_status()["direct"]
    scans = [synth.generate_scan(synth.pose_k(MAP_SCANS + j), RINGS, COLS, stream=MAP_SCANS + j) for j in range(n_cp - 1)]
    spec = bench.window_spec(synth, MAP_SCANS - 1, n_cp)
    op, gp = po.Problem(), hg.Problem(ctx)
    bench.window_build(op, spec, scans, og, RINGS * COLS)
    bench.window_build(gp, spec, [torch.from_numpy(s).to(dev) for s in scans], gg, RINGS * COLS)
    assert gp.num_columns() == 81
    so, sg = op.solve(), gp.solve()
    assert (sg.num_iterations, sg.num_successful_steps, sg.termination_type, sg.termination_reason) == \
           (so.num_iterations, so.num_successful_steps, so.termination_type, so.termination_reason)
    assert abs(sg.final_cost - so.final_cost) <= 1e-9 * max(1.0, abs(so.final_cost))
    for i in range(n_cp):
        a, b = op.get_pose(i), gp.get_pose(i)
        assert np.linalg.norm(a[:3] - b[:3]) < POSE_TOL_M
        assert rot_angle(a[3:], b[3:]) < POSE_TOL_RAD
        np.testing.assert_allclose(gp.get_velocity(i), op.get_velocity(i), atol=1e-6)


def test_register_scan_batch_of_100k_point_scans_against_oracle(po, hg, ctx):
    """hg_register_scan_batch at full size: four independent submaps, one 100 000-point registration step
    each (batched matcher + job-table insert kernels); submaps 0 and 3 replayed by the oracle: pose
    within 1e-4 m / 1e-4 rad, same iterations, and every voxel code of their three grids."""
    import torch
    dev = torch.device("cuda", 0)
    S, map_scans = 4, 3
    ins = [hg.TSDFRangeDataInserter3D() for _ in bench.RESOLUTIONS]
    pyramids, oracles, queries, guesses = [], {}, [], []
    for j in range(S):
        sb = 100000 * (j + 1)
        grids = [hg.HybridGridTSDF(ctx, r, max_blocks=1 << 16) for r in bench.RESOLUTIONS]
        if j in (0, 3):
            oracles[j] = [po.Grid(r) for r in bench.RESOLUTIONS]
        for pose, pts in bench.make_scans(RINGS, COLS, 0, map_scans, sb):
            hg.insert_pyramid(ins, hg.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), grids,
                              pose_tq=pose.astype(np.float32))
            if j in oracles:
                loc = synth.transform_points(pose, pts)
                for g in oracles[j]:
                    g.insert(pose[:3], loc)
        pose, pts = bench.make_scans(RINGS, COLS, map_scans, 1, sb)[0]
        pyramids.append(grids)
        queries.append((pts, torch.from_numpy(pts).to(dev)))
        guesses.append(synth.pose_mul(pose, synth.perturbation()))
    scale = 1.0 / np.sqrt(float(RINGS * COLS))
    problems = [hg.Problem(ctx) for _ in range(S)]
    for j in range(S):
        problems[j].add_pose(guesses[j])
        problems[j].add_block(queries[j][1], pyramids[j], scale, 0, multi_res=True)
    poses, summ = hg.register_scan_batch(problems, [0] * S, ins, [hg.RangeData([0, 0, 0], d) for _, d in queries], pyramids)
    ctx.synchronize()
    for j, og in oracles.items():
        op = po.Problem()
        op.add_pose(guesses[j])
        op.add_block(queries[j][0], og, scale, 0, multi_res=True)
        so = op.solve()
        ref = op.get_pose(0)
        assert np.linalg.norm(ref[:3] - poses[j][:3]) < POSE_TOL_M
        assert rot_angle(ref[3:], poses[j][3:]) < POSE_TOL_RAD
        assert (so.num_iterations, so.termination_type, so.termination_reason) == \
               (summ[j].num_iterations, summ[j].termination_type, summ[j].termination_reason)
        at = ref if np.array_equal(ref.astype(np.float32), poses[j].astype(np.float32)) else poses[j]
        loc = synth.transform_points(at, queries[j][0])
        for o, g in zip(og, pyramids[j]):
            o.insert(at[:3].astype(np.float32), loc)
            g.status()
            for x, y in zip(o.export(), g.export()):
                assert np.array_equal(x, y)


def test_both_hand_overs_of_the_partial_sums_pass_the_bench_gate():
    """The single-pose chain hands its partial sums to the LM tail as tagged granules (round 5) or, with
    HG_TICKET_HANDOVER=1 (the default of the context option ticket_handover), by acknowledged stores and a ticket (rounds 1 - 4). The switch is read when a context is created, so
    each form runs the headline command in a process of its own: bench.py aborts unless the first timed steps equal the
    oracle's (poses within 1e-4 m / 1e-4 rad, same iterations and termination), and the two forms must agree with
    each other far below that."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = []
    # (round 6) and the third form: the whole solve as ONE persistent launch, which bench.py's headline asks for
    # (hg_ctx_set_option "persistent_solve"); --no-persistent-solve keeps a launch per evaluation
    for env_extra, extra_args in (({}, []), ({}, ["--no-persistent-solve"]), ({"HG_TICKET_HANDOVER": "1"}, ["--no-persistent-solve"])):
        env = dict(os.environ, **env_extra)
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "1", "--cpu-scans", "2",
                              "--no-secondary", "--host-steps", "0"] + extra_args, capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        lines.append(json.loads(out.stdout.strip().splitlines()[-1]))
    for d in lines:
        assert d["parity"]["same_iterations_and_termination"] and d["parity"]["max_dt_m"] < 1e-9
    assert lines[0]["roofline"]["solve_form"].startswith("persistent") and lines[1]["roofline"]["solve_form"].startswith("one ")
    assert lines[0]["config"]["mean_pose_error_m"] == lines[1]["config"]["mean_pose_error_m"]   # the same arithmetic
    assert abs(lines[0]["config"]["mean_pose_error_m"] - lines[2]["config"]["mean_pose_error_m"]) < 1e-9
