"""Diagnostic (not part of the product): insert kernel times per level (single-level inserts)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hectorgrapher_amd import api, synth
import bench
ctx = api.Context(0)
dev = torch.device("cuda", 0)
for res in (0.05, 0.10, 0.20):
    g = [api.HybridGridTSDF(ctx, res, max_blocks=1 << 18)]
    ins = [api.TSDFRangeDataInserter3D()]
    for pose, pts in bench.make_scans(50, 2000, 0, 10, 0):
        api.insert_pyramid(ins, api.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), g, pose_tq=pose.astype(np.float32))
    pose, pts = bench.make_scans(50, 2000, 20, 1, 0)[0]
    d = torch.from_numpy(pts).to(dev)
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(5):
        api.insert_pyramid(ins, api.RangeData([0, 0, 0], d), g, pose_tq=pose.astype(np.float32), want_stats=False)
    ctx.synchronize()
    pr = ctx.prof_read(); ctx.prof_enable(False)
    print("res %.2f: count %.1f offsets %.1f scatter %.1f apply %.1f us" % (res, pr["ray_count"][1] / 5 * 1e3, pr["scan"][1] / 5 * 1e3, pr["ray_expand"][1] / 5 * 1e3, pr["apply"][1] / 5 * 1e3), flush=True)
