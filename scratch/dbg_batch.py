import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import torch, bench
from hectorgrapher_amd import api, synth
dev = torch.device("cuda", 0)
ctx = api.Context(0)
S = int(sys.argv[1]); steps = 24
ins = [api.TSDFRangeDataInserter3D() for _ in bench.RESOLUTIONS]
pyr, q, gs, pr = [], [], [], []
for j in range(S):
    sb = 100000 * (j + 1)
    grids = [api.HybridGridTSDF(ctx, r, max_blocks=1 << 17) for r in bench.RESOLUTIONS]
    for pose, pts in bench.make_scans(50, 2000, 0, 10, sb):
        api.insert_pyramid(ins, api.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), grids, pose_tq=pose.astype(np.float32))
    qq = bench.make_scans(50, 2000, 10, steps, sb)
    pyr.append(grids); q.append([(pose, torch.from_numpy(pts).to(dev)) for pose, pts in qq])
    gs.append([synth.pose_mul(pose, synth.perturbation()) for pose, _ in qq]); pr.append(api.Problem(ctx))
torch.cuda.synchronize()
scale = 1.0 / np.sqrt(100000.0)
for i in range(steps):
    t0 = time.perf_counter()
    for j in range(S):
        p = pr[j]; p.reset(); p.add_pose(gs[j][i]); p.add_block(q[j][i][1], pyr[j], scale, 0, multi_res=True)
    t1 = time.perf_counter()
    poses, summ = api.register_scan_batch(pr, [0] * S, ins, [api.RangeData([0, 0, 0], q[j][i][1]) for j in range(S)], pyr)
    t2 = time.perf_counter()
    ctx.synchronize()
    t3 = time.perf_counter()
    print("step %d prep %.3f ms call %.3f ms drain %.3f ms its %s" % (i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, [s.num_iterations for s in summ]))
for j in range(min(S, 2)):
    for g in pyr[j]:
        print(j, g.window_status())
