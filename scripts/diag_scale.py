"""Diagnostic (not part of the product): residual kernel time vs number of returns (one block)."""
import sys, os
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
guess = synth.pose_mul(pose, synth.perturbation())
for mult in (1, 2, 4, 9):
    d = torch.from_numpy(np.concatenate([pts] * mult)).to(dev)
    p = api.Problem(ctx)
    i = p.add_pose(guess)
    p.add_block(d, grids, 1e-2, i, multi_res=True)
    p.evaluate(want_residuals=False)
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(10):
        p.evaluate(want_residuals=False)
    ctx.synchronize()
    pr = ctx.prof_read(); ctx.prof_enable(False)
    print("%7d returns: residual kernel %.1f us" % (len(pts) * mult, pr["residuals"][1] / pr["residuals"][0] * 1e3), flush=True)
