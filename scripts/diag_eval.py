"""Diagnostic (not part of the product): residual-kernel time vs pyramid levels."""
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
pose, pts = bench.make_scans(50, 2000, 10, 1, 0)[0]
d = torch.from_numpy(pts).to(dev)
guess = synth.pose_mul(pose, synth.perturbation())
for name, gl, multi in [("1 level (0.05)", [grids[0]], False), ("1 level (0.20)", [grids[2]], False),
                        ("2 levels", grids[:2], True), ("3 levels", grids, True)]:
    p = api.Problem(ctx)
    i = p.add_pose(guess)
    p.add_block(d, gl, 1e-2, i, multi_res=multi)
    p.evaluate(want_residuals=False)
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(20):
        p.evaluate(want_residuals=False)
    ctx.synchronize()
    pr = ctx.prof_read(); ctx.prof_enable(False)
    print("%-16s residual kernel %.2f us" % (name, pr["residuals"][1] / pr["residuals"][0] * 1e3), flush=True)
