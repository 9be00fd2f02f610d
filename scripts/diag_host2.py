"""Diagnostic (not part of the product): Python-side time split of the bench step."""
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
q = bench.make_scans(50, 2000, 10, 45, 0)
d = [torch.from_numpy(p).to(dev) for _, p in q]
guess = [synth.pose_mul(pose, synth.perturbation()) for pose, _ in q]
torch.cuda.synchronize()
problem = api.Problem(ctx)
scale = 1.0 / np.sqrt(100000.0)
T = [0.0] * 6
errs = []
for i in range(45):
    t0 = time.perf_counter()
    ctx.prof_enable(False)
    t1 = time.perf_counter()
    problem.reset()
    pi = problem.add_pose(guess[i])
    problem.add_block(d[i], grids, scale, pi, multi_res=True)
    t2 = time.perf_counter()
    rd = api.RangeData([0, 0, 0], d[i])
    t3 = time.perf_counter()
    est, s = api.register_scan(problem, pi, ins, rd, grids)
    t4 = time.perf_counter()
    errs.append(float(np.linalg.norm(est[:3] - q[i][0][:3])))
    t5 = time.perf_counter()
    if i >= 5:
        for k, (a, b) in enumerate([(t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5)]):
            T[k] += b - a
print("us per step: prof_enable %.1f, problem setup %.1f, RangeData %.1f, register_scan %.1f, err %.1f" % tuple(x / 40 * 1e6 for x in T[:5]))
