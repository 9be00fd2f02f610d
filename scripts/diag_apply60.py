"""Diagnostic (not part of the product): k_bin_apply time for the k=60 scan in different map states."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hectorgrapher_amd import api, synth
import bench
ctx = api.Context(0)
dev = torch.device("cuda", 0)
def timed_insert(grids, ins, k, label):
    pose, pts = bench.make_scans(50, 2000, k, 1, 0)[0]
    d = torch.from_numpy(pts).to(dev)
    torch.cuda.synchronize()
    ctx.prof_enable(True); ctx.prof_reset()
    st = api.insert_pyramid(ins, api.RangeData([0, 0, 0], d), grids, pose_tq=pose.astype(np.float32))
    pr = ctx.prof_read(); ctx.prof_enable(False)
    print("%s: apply %.1f us, updates %s" % (label, pr["apply"][1] * 1e3, [s.num_updates for s in st]), flush=True)
grids = [api.HybridGridTSDF(ctx, r, max_blocks=1 << 18) for r in bench.RESOLUTIONS]
ins = [api.TSDFRangeDataInserter3D() for _ in grids]
timed_insert(grids, ins, 5, "fresh map, k=5 ")
timed_insert(grids, ins, 60, "then k=60      ")
timed_insert(grids, ins, 60, "k=60 again     ")
timed_insert(grids, ins, 60, "k=60 third time")
for k in range(50, 60):
    pose, pts = bench.make_scans(50, 2000, k, 1, 0)[0]
    api.insert_pyramid(ins, api.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), grids, pose_tq=pose.astype(np.float32), want_stats=False)
timed_insert(grids, ins, 60, "after k=50..59 ")
