"""Where the host time of one match_batch step goes (round 5): rebuild of the problems in Python, the enqueue part of
hg_problem_solve_batch, the wait, the fetches. Run on the GPU box: HG_HOST_TIMES=1 python scripts/host_gap_probe.py [B]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hectorgrapher_amd import api, synth
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
ctx = api.Context(0)
rings, cols = 50, 2000
map_scans = bench.make_scans(rings, cols, 0, 10, 0)
grids = [api.HybridGridTSDF(ctx, r, max_blocks=1 << 18) for r in bench.RESOLUTIONS]
ins = [api.TSDFRangeDataInserter3D() for _ in grids]
for pose, pts in map_scans:
    api.insert_pyramid(ins, api.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), grids, pose_tq=pose.astype(np.float32))
queries = []
for j in range(B):
    pose = synth.pose_k(j % 10)
    pts = synth.generate_scan(pose, rings, cols, stream=5000 + j)
    queries.append((torch.from_numpy(pts).to(dev), synth.pose_mul(pose, synth.perturbation())))
torch.cuda.synchronize()
scale = 1.0 / np.sqrt(float(rings * cols))
problems = [api.Problem(ctx) for _ in range(B)]
for rep in range(6):
    t0 = time.perf_counter()
    for p, (d, guess) in zip(problems, queries):
        p.reset()
        i = p.add_pose(guess)
        p.add_block(d, grids, scale, i, multi_res=True, width=rings)
    t1 = time.perf_counter()
    summ = api.solve_batch(problems)
    t2 = time.perf_counter()
    print("step %d: rebuild %.1f us, solve_batch %.1f us" % (rep, (t1 - t0) * 1e6, (t2 - t1) * 1e6), file=sys.stderr)
