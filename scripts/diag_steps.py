"""Diagnostic (not part of the product): per-step insert kernel times along the bench trajectory."""
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
ks = [int(a) for a in sys.argv[1:]] or list(range(0, 121, 10))
for k in ks:
    scans = bench.make_scans(50, 2000, k, 1, 0)
    pose, pts = scans[0]
    d = torch.from_numpy(pts).to(dev)
    torch.cuda.synchronize()
    api.insert_pyramid(ins, api.RangeData([0, 0, 0], d), grids, pose_tq=pose.astype(np.float32))  # warm
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(5):
        api.insert_pyramid(ins, api.RangeData([0, 0, 0], d), grids, pose_tq=pose.astype(np.float32), want_stats=False)
    ctx.synchronize()
    prof = ctx.prof_read(); ctx.prof_enable(False)
    # longest per-voxel chain at the coarsest level (host estimate: hits per 0.2 m voxel of the hit cell x ~5)
    w = synth.transform_points(pose, pts)
    cells = np.round(w / 0.2).astype(np.int64)
    _, cnt = np.unique(cells, axis=0, return_counts=True)
    print("k=%3d pos=(%.2f,%.2f) count=%.1f scan=%.1f scatter=%.1f apply=%.1f us  max hits/0.2m voxel=%d" % (
        k, pose[0], pose[1], prof["ray_count"][1] / 5 * 1e3, prof["scan"][1] / 5 * 1e3,
        prof["ray_expand"][1] / 5 * 1e3, prof["apply"][1] / 5 * 1e3, cnt.max()), flush=True)
