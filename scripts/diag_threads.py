"""Diagnostic (not part of the product): S submaps registered by S host threads of ONE process, one
context / stream each, against the same work done by one thread."""
import gc, os, sys, threading, time
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


gc.collect()
gc.disable()   # a generational collection with torch loaded takes 30-60 ms: it would be measured as a stall
jobs = [Job(j) for j in range(S)]
for jb in jobs:
    jb.step(0); jb.step(1); jb.ctx.synchronize(); jb.t.clear()
t0 = time.perf_counter()
for jb in jobs:
    jb.run()
seq = time.perf_counter() - t0
jobs = [Job(j) for j in range(S)]
for jb in jobs:
    jb.step(0); jb.step(1); jb.ctx.synchronize(); jb.t.clear()
th = [threading.Thread(target=jb.run) for jb in jobs]
t0 = time.perf_counter()
for t in th:
    t.start()
for t in th:
    t.join()
par = time.perf_counter() - t0
print("S=%d steps=%d: one thread %.1f ms, %d threads %.1f ms (x%.2f)" % (S, STEPS, seq * 1e3, S, par * 1e3, seq / par))
tt = np.array(jobs[0].t) * 1e3
print("thread 0 per-step ms: first 10 mean %.3f, last 10 mean %.3f, max %.3f" % (tt[:10].mean(), tt[-10:].mean(), tt.max()))
