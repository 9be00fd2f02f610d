"""Tuning scratch: host / device split of one window step (bench.py --workload window shape)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from hectorgrapher_amd import api, synth
args = bench.parse_args()
dev = torch.device("cuda", 0)
ctx = api.Context(0)
n_pts = args.rings * args.cols
n_cp = args.window
grids = [api.HybridGridTSDF(ctx, r, max_blocks=args.max_blocks) for r in bench.RESOLUTIONS]
ins = [api.TSDFRangeDataInserter3D() for _ in grids]
for pose, pts in bench.make_scans(args.rings, args.cols, 0, args.map_scans, 0):
    api.insert_pyramid(ins, api.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), grids, pose_tq=pose.astype(np.float32))
scans = [torch.from_numpy(synth.generate_scan(synth.pose_k(args.map_scans + j), args.rings, args.cols, stream=args.map_scans + j)).to(dev)
         for j in range(n_cp + 8)]
torch.cuda.synchronize()
pr = api.Problem(ctx)
tb = ts = 0.0
for s in range(8):
    t0 = time.perf_counter()
    pr.reset()
    bench.window_problem(pr, synth, args.map_scans - 1 + s, n_cp, scans[s:s + n_cp - 1], grids, n_pts)
    t1 = time.perf_counter()
    ctx.prof_reset(); ctx.prof_enable(s == 7)
    summ = pr.solve()
    t2 = time.perf_counter()
    if s >= 2:
        tb += t1 - t0; ts += t2 - t1
    print("step", s, "build %.3f ms, solve %.3f ms, iterations %d, cost evals %d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, summ.num_iterations, summ.num_cost_evaluations))
prof = ctx.prof_read()
print({k: (v[0], round(v[1], 4)) for k, v in prof.items()})
