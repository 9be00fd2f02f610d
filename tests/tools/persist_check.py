"""Child process of tests/test_gpu_persistent.py: ONE context in the process, so the library may take the persistent
single-launch solve (hg_ctx_set_option "persistent_solve"). Runs a short registration trajectory twice -- a launch per
evaluation, then the persistent form -- from identical maps and prints one JSON line with both pose sequences.
usage: python tests/tools/persist_check.py <steps> <rings> <cols>"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hectorgrapher_amd import api as hg, synth  # noqa: E402

steps, rings, cols = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
RES = (0.05, 0.10, 0.20)
ctx = hg.Context(0)


def trajectory(persistent):
    ctx.set_option("persistent_solve", 1 if persistent else 0)
    grids = [hg.HybridGridTSDF(ctx, r, max_blocks=1 << 16) for r in RES]
    ins = [hg.TSDFRangeDataInserter3D() for _ in RES]
    for k in range(4):
        pose = synth.pose_k(k)
        hg.insert_pyramid(ins, hg.RangeData([0, 0, 0], synth.generate_scan(pose, rings, cols, stream=k)), grids,
                          pose_tq=pose.astype(np.float32))
    out = []
    p = hg.Problem(ctx)
    for k in range(4, 4 + steps):
        pose = synth.pose_k(k)
        pts = synth.generate_scan(pose, rings, cols, stream=k)
        guess = synth.pose_mul(pose, synth.perturbation())
        p.reset()
        i = p.add_pose(guess)
        p.add_block(pts, grids, 1.0 / np.sqrt(len(pts)), i, multi_res=True)
        est, summ = hg.register_scan(p, i, ins, hg.RangeData([0, 0, 0], pts, width=rings), grids)
        out.append((est.tolist(), summ.num_iterations, summ.termination_type, summ.termination_reason))
    codes = [[a.tolist() for a in g.export()[1:]] for g in grids[2:]]  # the coarsest level's voxel codes
    p.close()
    for g in grids:
        g.close()
    return out, codes


a, ca = trajectory(False)
b, cb = trajectory(True)
print(json.dumps({"per_evaluation": a, "persistent": b, "voxels_equal": ca == cb,
                  "persistent_option": ctx.get_option("persistent_solve")}))
