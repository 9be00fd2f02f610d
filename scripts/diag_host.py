"""Diagnostic (not part of the product): host-side time of one registration step."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hectorgrapher_amd import api, synth
import bench
ctx = api.Context(0)
dev = torch.device("cuda", 0)
grids = [api.HybridGridTSDF(ctx, r, max_blocks=1 << 18) for r in bench.RESOLUTIONS]
ins = [api.TSDFRangeDataInserter3D() for _ in grids]
for pose, pts in bench.make_scans(50, 2000, 0, 10, 0):
    api.insert_pyramid(ins, api.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), grids, pose_tq=pose.astype(np.float32))
q = bench.make_scans(50, 2000, 10, 30, 0)
d = [torch.from_numpy(p).to(dev) for _, p in q]
guess = [synth.pose_mul(pose, synth.perturbation()) for pose, _ in q]
torch.cuda.synchronize()
problem = api.Problem(ctx)
scale = 1.0 / np.sqrt(100000.0)
T = {"setup": 0.0, "register": 0.0}
for i in range(30):
    t0 = time.perf_counter()
    problem.reset()
    pi = problem.add_pose(guess[i])
    problem.add_block(d[i], grids, scale, pi, multi_res=True)
    t1 = time.perf_counter()
    est, s = api.register_scan(problem, pi, ins, api.RangeData([0, 0, 0], d[i]), grids)
    t2 = time.perf_counter()
    if i >= 5:
        T["setup"] += t1 - t0; T["register"] += t2 - t1
print("per step: python problem setup %.1f us, register_scan call (enqueue + wait for pose) %.1f us" % (T["setup"] / 25 * 1e6, T["register"] / 25 * 1e6))
# enqueue-only cost: solve_async + insert without waiting
import ctypes as C
L = api._lib.load()
o = api.SolverOpts(); L.hg_solver_default_opts(C.byref(o))
ctx.synchronize()
t0 = time.perf_counter()
for i in range(20):
    problem.reset(); pi = problem.add_pose(guess[i]); problem.add_block(d[i], grids, scale, pi, multi_res=True)
    api.check(L.hg_problem_solve_async(problem._h, C.byref(o)))
    tq = time.perf_counter()
    api.check(L.hg_problem_fetch(problem._h, None))
t1 = time.perf_counter()
ctx.synchronize()
print("solve_async + fetch loop: %.1f us per solve" % ((t1 - t0) / 20 * 1e6))
ctx.synchronize()
t0 = time.perf_counter()
problem.reset(); pi = problem.add_pose(guess[0]); problem.add_block(d[0], grids, scale, pi, multi_res=True)
api.check(L.hg_problem_solve_async(problem._h, C.byref(o)))
t1 = time.perf_counter()
api.check(L.hg_problem_fetch(problem._h, None))
t2 = time.perf_counter()
print("one solve: enqueue %.1f us, wait %.1f us" % ((t1 - t0) * 1e6, (t2 - t1) * 1e6))
