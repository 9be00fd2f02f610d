"""Diagnostic (not part of the product): wall time of a sliding-window solve (OLTB shape)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hectorgrapher_amd import api, synth
import bench

ctx = api.Context(0)
dev = torch.device("cuda", 0)
grids = [api.HybridGridTSDF(ctx, r, max_blocks=1 << 18) for r in bench.RESOLUTIONS]
ins = [api.TSDFRangeDataInserter3D() for _ in grids]
for pose, pts in bench.make_scans(50, 2000, 0, 10, 0):
    api.insert_pyramid(ins, api.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), grids, pose_tq=pose.astype(np.float32))
n_cp = int(sys.argv[1]) if len(sys.argv) > 1 else 6
scans = bench.make_scans(50, 2000, 10, n_cp - 1, 0)
d = [torch.from_numpy(p).to(dev) for _, p in scans]
torch.cuda.synchronize()
for rep in range(3):
    p = api.Problem(ctx)
    for i in range(n_cp):
        tq = synth.pose_k(9 + i) if i == 0 else synth.pose_mul(synth.pose_k(9 + i), synth.perturbation())
        p.add_pose(tq, i == 0)
        p.set_velocity(i, np.array([0.5, 0.2, 0.0]), i == 0)
    for i in range(1, n_cp):
        delta = synth.pose_mul(synth.pose_inverse(synth.pose_k(9 + i)), synth.pose_k(8 + i))
        dq = synth.pose_mul(synth.pose_inverse(synth.pose_k(8 + i)), synth.pose_k(9 + i))[3:]
        p.add_odometry_block(i - 1, i, 12.0, 30.0, delta)
        p.add_imu_block(i - 1, i, 3.0, 2.0, 70.0, 0.1, dq)
        p.add_block(d[i - 1], grids, 1.0 / np.sqrt(100000.0), i, multi_res=True)
    ctx.synchronize()
    t0 = time.perf_counter()
    s = p.solve()
    t1 = time.perf_counter()
    print("window %d control points, %d scans x 100k: solve %.3f ms, %d iterations, cost %.3e -> %.3e" % (
        n_cp, n_cp - 1, (t1 - t0) * 1e3, s.num_iterations, s.initial_cost, s.final_cost), flush=True)
ctx.prof_enable(True); ctx.prof_reset()
for _ in range(5):
    p.evaluate(want_residuals=False)
ctx.synchronize()
pr = ctx.prof_read(); ctx.prof_enable(False)
print("evaluate(): residual launch %.1f us (x%d)" % (pr["residuals"][1] / pr["residuals"][0] * 1e3, pr["residuals"][0]))
