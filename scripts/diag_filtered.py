"""Where the register_filtered step spends its time: filter / gather / solve / insert (host clock)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hectorgrapher_amd import api, synth

dev = torch.device("cuda", 0)
ctx = api.Context(0)
RES = (0.05, 0.10, 0.20)
grids = [api.HybridGridTSDF(ctx, r, max_blocks=1 << 18) for r in RES]
ins = [api.TSDFRangeDataInserter3D() for _ in grids]
for k in range(10):
    pose = synth.pose_k(k)
    pts = synth.generate_scan(pose, 50, 2000, stream=k)
    api.insert_pyramid(ins, api.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), grids, pose_tq=pose.astype(np.float32))
pose = synth.pose_k(10)
pts = synth.generate_scan(pose, 50, 2000, stream=10)
d = torch.from_numpy(pts).to(dev)
guess = synth.pose_mul(pose, synth.perturbation())
avf = api.AdaptiveVoxelFilter(ctx, 2.0, 150, 15.0)
pr = api.Problem(ctx)
torch.cuda.synchronize()


def timed(f, reps=20):
    f()
    ctx.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    ctx.synchronize(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


idx = avf.Filter(d)
print("filter            %.1f us (kept %d)" % (timed(lambda: avf.Filter(d)), len(idx)))
sel = d[torch.from_numpy(idx.astype(np.int64)).to(dev)].contiguous()
print("torch gather      %.1f us" % timed(lambda: d[torch.from_numpy(idx.astype(np.int64)).to(dev)].contiguous()))


def solve():
    pr.reset()
    pi = pr.add_pose(guess)
    pr.add_block(sel, grids, 1.0 / np.sqrt(len(idx)), pi, multi_res=True)
    return pr.solve()


s = solve()
print("solve (%d it)      %.1f us" % (s.num_iterations, timed(solve)))
print("insert            %.1f us" % timed(lambda: api.insert_pyramid(ins, api.RangeData([0, 0, 0], d), grids, pose_tq=pose.astype(np.float32), want_stats=False)))
