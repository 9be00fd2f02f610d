"""Diagnostic (not part of the product): in-kernel timeline of one registration solve. Needs a
library built with EXTRA=-DHG_EVAL_STAMPS (prints the stamps of the last evaluation to stderr)."""
import os, sys
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
p = api.Problem(ctx)
for rep in range(4):
    p.reset()
    i = p.add_pose(guess)
    p.add_block(d, grids, 1.0 / np.sqrt(len(pts)), i, multi_res=True)
    s = p.solve()
print("iterations", s.num_iterations)
