import sys, numpy as np
sys.path.insert(0, '/root/repo')
import torch, bench
from hectorgrapher_amd import api, synth
dev = torch.device("cuda", 0)
ctx = api.Context(0)
gg = [api.HybridGridTSDF(ctx, r, max_blocks=1 << 18) for r in bench.RESOLUTIONS]
ins = [api.TSDFRangeDataInserter3D() for _ in gg]
for pose, pts in bench.make_scans(50, 2000, 0, 10, 0):
    api.insert_pyramid(ins, api.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), gg, pose_tq=pose.astype(np.float32))
for g in gg:
    print(g.window_status())
