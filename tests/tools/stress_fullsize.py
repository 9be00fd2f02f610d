"""Full-size exactness check (test infrastructure, uses the oracle; not part of the product): 100k-point scans along the bench trajectory,
including the positions with ~3000-update chains, GPU pyramid insert vs oracle, bit-exact."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po
from hectorgrapher_amd import api as hg, synth
import bench
ctx = hg.Context(0)
ogs = [po.Grid(r) for r in bench.RESOLUTIONS]
ggs = [hg.HybridGridTSDF(ctx, r, max_blocks=1 << 18) for r in bench.RESOLUTIONS]
ins = [hg.TSDFRangeDataInserter3D() for _ in ggs]
bad = 0
for k in [int(a) for a in sys.argv[1:]] or [0, 20, 40, 55, 60, 60, 58, 30]:
    pose, pts = bench.make_scans(50, 2000, k, 1, 0)[0]
    loc = synth.transform_points(pose, pts)
    ref = [g.insert(pose[:3].astype(np.float32), loc) for g in ogs]
    st = hg.insert_pyramid(ins, hg.RangeData([0, 0, 0], pts), ggs, pose_tq=pose.astype(np.float32))
    same = all((s.num_hits, s.num_updates) == tuple(r) for s, r in zip(st, ref))
    for o, g in zip(ogs, ggs):
        same = same and all(np.array_equal(x, y) for x, y in zip(o.export(), g.export()))
    print("k=%d exact=%s" % (k, same), flush=True)
    bad += 0 if same else 1
sys.exit(1 if bad else 0)
