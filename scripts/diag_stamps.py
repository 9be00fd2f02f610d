"""Diagnostic (not part of the product): per-item phase times of k_bin_apply (library built with
-DHG_BIN_STAMPS, HG_LIB_PATH=scripts/libhg_stamps.so) for the bench map at step K."""
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
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
os.environ["HG_QUIET"] = "1"
devnull = os.open(os.devnull, os.O_WRONLY)
saved = os.dup(2)
os.dup2(devnull, 2)
for pose, pts in bench.make_scans(50, 2000, 0, K, 0):
    api.insert_pyramid(ins, api.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), grids, pose_tq=pose.astype(np.float32), want_stats=False)
ctx.synchronize()
os.dup2(saved, 2)
pose, pts = bench.make_scans(50, 2000, K, 1, 0)[0]
api.insert_pyramid(ins, api.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), grids, pose_tq=pose.astype(np.float32), want_stats=False)
ctx.synchronize()
