"""Diagnostic (not part of the product): step time of a submap as a function of which context / stream of the process it uses."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hectorgrapher_amd import api, synth
import bench

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)


class Job:
    def __init__(self, j):
        self.ctx = api.Context(0)
        self.grids = [api.HybridGridTSDF(self.ctx, r, max_blocks=1 << 18) for r in bench.RESOLUTIONS]
        self.ins = [api.TSDFRangeDataInserter3D() for _ in self.grids]
        for pose, pts in bench.make_scans(50, 2000, 0, 10, 1000 * j):
            api.insert_pyramid(self.ins, api.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), self.grids,
                               pose_tq=pose.astype(np.float32))
        self.q = bench.make_scans(50, 2000, 10, STEPS + 2, 1000 * j)
        self.d = [torch.from_numpy(p).to(dev) for _, p in self.q]
        self.g = [synth.pose_mul(p, synth.perturbation()) for p, _ in self.q]
        self.pr = api.Problem(self.ctx)
        self.t = []

    def step(self, i):
        t0 = time.perf_counter()
        self.pr.reset()
        pi = self.pr.add_pose(self.g[i])
        self.pr.add_block(self.d[i], self.grids, 1.0 / np.sqrt(100000.0), pi, multi_res=True)
        api.register_scan(self.pr, pi, self.ins, api.RangeData([0, 0, 0], self.d[i]), self.grids)
        self.t.append(time.perf_counter() - t0)

    def run(self):
        for i in range(2, STEPS + 2):
            self.step(i)
        self.ctx.synchronize()



jobs = [Job(j) for j in range(S)]
for jb in jobs:
    jb.step(0); jb.step(1); jb.ctx.synchronize(); jb.t.clear()
order = list(range(S)) + [0]
for k in order:
    jb = jobs[k]
    jb.t.clear()
    t0 = time.perf_counter()
    jb.run()
    el = time.perf_counter() - t0
    tt = np.array(jb.t) * 1e3
    print("context %d alone: %.3f ms per step (median %.3f, max %.3f at step %d)" % (k, el / STEPS * 1e3, np.median(tt), tt.max(), int(tt.argmax())), flush=True)
