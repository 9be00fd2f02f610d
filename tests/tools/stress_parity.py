"""Randomised parity sweep (GPU vs oracle), broader than the unit tests. Test infrastructure (uses the oracle); not part of the product.
usage: python tests/tools/stress_parity.py [n_insert_cases] [n_solver_cases] [seed]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyoracle as po
from hectorgrapher_amd import api as hg, synth
from conftest import build_map

n_ins = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_sol = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
ctx = hg.Context(0)
bad = 0

def grids_equal(og, gg):
    a = og.export(); b = gg.export()
    return all(np.array_equal(x, y) for x, y in zip(a, b))

for case in range(n_ins):
    res = float(rng.choice([0.05, 0.1, 0.2, 0.33]))
    maxw = float(rng.choice([3.0, 25.0, 1000.0]))
    rtd = float(rng.choice([1.5, 2.5, 3.0]))
    rings, cols = int(rng.choice([8, 16, 32])), int(rng.choice([100, 400, 900]))
    kw = dict(relative_truncation_distance=rtd, maximum_weight=maxw, min_range=float(rng.choice([0.0, 0.4, 2.0])),
              max_range=float(rng.choice([6.0, 15.0, 60.0])), insertion_ratio=float(rng.choice([1.0, 0.37])))
    og = po.Grid(res, rtd, maxw)
    gg = hg.HybridGridTSDF(ctx, res, rtd, maxw, max_blocks=1 << 15)
    oi = po.InsertOpts(**kw); gi = hg.TSDFRangeDataInserter3D(hg.InsertOpts(**kw))
    ok = True
    for rep in range(int(rng.integers(1, 6))):
        k = int(rng.integers(0, 60))
        pose = synth.pose_k(k)
        pts = synth.generate_scan(pose, rings, cols, stream=1000 * case + rep)
        if rng.random() < 0.3:   # dense cluster -> long chains / giant voxels
            hot = pts[int(rng.integers(0, len(pts)))]
            pts = np.concatenate([pts, hot + (rng.standard_normal((int(rng.integers(500, 6000)), 3)) * 2e-3).astype(np.float32)])
        loc = synth.transform_points(pose, pts)
        a = og.insert(pose[:3].astype(np.float32), loc, oi)
        st = gi.Insert(hg.RangeData(pose[:3].astype(np.float32), loc), gg)
        ok = ok and (st.num_hits, st.num_updates) == tuple(a)
    ok = ok and grids_equal(og, gg)
    if not ok:
        bad += 1
        print("INSERT MISMATCH case", case, res, kw, flush=True)
    gg.close() if hasattr(gg, "close") else None
print("insert cases: %d, mismatches so far: %d" % (n_ins, bad), flush=True)

og, gg = build_map(po, (ctx, hg), [0.05, 0.10, 0.20], 16, 400, 8, max_blocks=1 << 16)
def rot_angle(qa, qb):
    return 2.0 * np.arccos(min(1.0, abs(float(np.dot(qa, qb)))))
for case in range(n_sol):
    n_cp = int(rng.integers(1, 11))
    op, gp = po.Problem(), hg.Problem(ctx)
    const0 = n_cp > 1 and rng.random() < 0.8
    use_vel = n_cp > 1 and rng.random() < 0.6
    for i in range(n_cp):
        tq = synth.pose_k(2 + i)
        if not (i == 0 and const0):
            tq = synth.pose_mul(tq, synth.perturbation())
        for pr in (op, gp):
            pr.add_pose(tq, i == 0 and const0)
            if use_vel:
                pr.set_velocity(i, np.array([0.5, 0.2, 0.0]) + 0.02 * i, i == 0 and const0)
    nblk = 0
    desc = []
    for i in range(n_cp):
        if i == 0 and const0 and n_cp > 1:
            continue
        kind = int(rng.integers(0, 3)) if i > 0 else 0
        pts = synth.generate_scan(synth.pose_k(2 + i), 16, int(rng.choice([40, 150])), stream=5000 + 20 * case + i)
        lv = [int(rng.integers(0, 3))]
        multi = rng.random() < 0.5
        if multi: lv = [0, 1, 2]
        s = float(rng.choice([0.5, 1.0])) / np.sqrt(len(pts))
        if kind == 0:
            op.add_block(pts, [og[l] for l in lv], s, i, -1, 0.0, multi); gp.add_block(pts, [gg[l] for l in lv], s, i, -1, 0.0, multi)
        elif kind == 1:
            f = float(rng.random())
            op.add_block(pts, [og[l] for l in lv], s, i - 1, i, f, multi); gp.add_block(pts, [gg[l] for l in lv], s, i - 1, i, f, multi)
        else:
            f = np.sort(rng.random(len(pts)))
            gp.add_unwarped_block(pts, f, [gg[l] for l in lv], s, i - 1, i, multi_res=multi)
            for j in range(len(pts)):
                op.add_block(pts[j:j + 1], [og[l] for l in lv], s, i - 1, i, float(f[j]), multi)
        nblk += 1
        desc.append(("tsdf", i, kind, lv, multi))
    for i in range(1, n_cp):
        if rng.random() < 0.7:
            desc.append(("odom", i))
            delta = synth.pose_mul(synth.pose_inverse(synth.pose_k(2 + i)), synth.pose_k(1 + i))
            for pr in (op, gp): pr.add_odometry_block(i - 1, i, 12.0, 30.0, delta)
        if use_vel and rng.random() < 0.7:
            desc.append(("imu", i))
            dq = synth.pose_mul(synth.pose_inverse(synth.pose_k(1 + i)), synth.pose_k(2 + i))[3:]
            for pr in (op, gp): pr.add_imu_block(i - 1, i, 3.0, 2.0, 70.0, 0.1, dq)
    if gp.num_columns() == 0 or gp.num_residuals() == 0:
        continue
    c0, r0, J0, g0 = op.evaluate()
    c1, r1, g1, H1 = gp.evaluate()
    eval_ok = np.allclose(H1, J0.T @ J0, rtol=1e-8, atol=1e-9) and np.allclose(g1, g0, rtol=1e-8, atol=1e-10)
    so, sg = op.solve(), gp.solve()
    ok = so.num_iterations == sg.num_iterations and so.termination_reason == sg.termination_reason
    for i in range(n_cp):
        a, b = op.get_pose(i), gp.get_pose(i)
        ok = ok and np.linalg.norm(a[:3] - b[:3]) < 1e-4 and rot_angle(a[3:], b[3:]) < 1e-4
    if not ok:
        bad += 1
        print("SOLVER MISMATCH case", case, "n_cp", n_cp, "cols", gp.num_columns(), "it", so.num_iterations, sg.num_iterations,
              "reason", so.termination_reason, sg.termination_reason, "const0", const0, "vel", use_vel,
              "eval_ok", eval_ok, desc, flush=True)
print("solver cases: %d; total mismatches: %d" % (n_sol, bad), flush=True)
sys.exit(1 if bad else 0)
