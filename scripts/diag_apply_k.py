"""Diagnostic (not part of the product): HIP-event time of the four binned insert kernels for single scans at chosen
trajectory positions (light positions, next to a wall), for the library HG_LIB_PATH points to.
Usage: python scripts/diag_apply_k.py 10 40 50 60"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hectorgrapher_amd import api
import bench
ctx = api.Context(0)
dev = torch.device("cuda", 0)
grids = [api.HybridGridTSDF(ctx, r, max_blocks=1 << 18) for r in bench.RESOLUTIONS]
ins = [api.TSDFRangeDataInserter3D() for _ in grids]
for rep in range(2):  # (the second visit of a position finds saturated weights on the heavy voxels)
    for k in [int(a) for a in sys.argv[1:]]:
        pose, pts = bench.make_scans(50, 2000, k, 1, 0)[0]
        d = torch.from_numpy(pts).to(dev)
        torch.cuda.synchronize()
        ctx.prof_reset()
        ctx.prof_enable(True)
        for _ in range(3):
            api.insert_pyramid(ins, api.RangeData([0, 0, 0], d), grids, pose_tq=pose.astype(np.float32), want_stats=False)
        ctx.synchronize()
        p = ctx.prof_read()
        ctx.prof_enable(False)
        print("visit %d k=%d" % (rep, k), {n: round(v[1] / max(1, v[0]) * 1e3, 1) for n, v in p.items() if v[0]}, flush=True)
