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
