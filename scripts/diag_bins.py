"""Diagnostic (not part of the product): k_bin_apply phase timings at chosen trajectory positions
(library built with -DHG_BIN_STAMPS prints them to stderr)."""
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
for k in [int(a) for a in sys.argv[1:]]:
    pose, pts = bench.make_scans(50, 2000, k, 1, 0)[0]
    d = torch.from_numpy(pts).to(dev)
    torch.cuda.synchronize()
    sys.stderr.write("=== k=%d\n" % k); sys.stderr.flush()
    api.insert_pyramid(ins, api.RangeData([0, 0, 0], d), grids, pose_tq=pose.astype(np.float32), want_stats=False)
    ctx.synchronize()
