#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/hg_oracle.hpp).

The reference itself cannot be built in this image (Eigen/Ceres/glog/protobuf are absent,
SURVEY.md §8c), so these vectors come from the restatement; they pin the oracle and the HIP path
against regressions and carry the derived known answers of SURVEY.md Appendix B.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402
from hectorgrapher_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    res = [0.10, 0.20]
    grids = [po.Grid(r) for r in res]
    scans, origins = [], []
    for k in range(3):
        pose = synth.pose_k(k)
        pts = synth.transform_points(pose, synth.generate_scan(pose, 8, 96, stream=k))
        scans.append(pts)
        origins.append(pose[:3].astype(np.float32))
        for g in grids:
            g.insert(origins[-1], pts)
    out = {"resolutions": np.array(res, np.float32), "origins": np.array(origins),
           "scans": np.array(scans)}
    for i, g in enumerate(grids):
        ijk, t, w = g.export()
        out["ijk%d" % i], out["tsd%d" % i], out["w%d" % i] = ijk, t, w
    pose = synth.pose_k(3)
    q = synth.generate_scan(pose, 8, 96, stream=3)
    guess = synth.pose_mul(pose, synth.perturbation())
    out["query"], out["guess"] = q, guess
    for name, multi, gl in (("single", False, [grids[0]]), ("multi", True, grids)):
        pr = po.Problem()
        pr.add_pose(guess)
        pr.add_block(q, gl, 1.0 / np.sqrt(len(q)), 0, multi_res=multi)
        c, r, J, g = pr.evaluate()
        s = pr.solve()
        out[name + "_cost"] = np.array([c])
        out[name + "_residuals"] = r
        out[name + "_gradient"] = g
        out[name + "_JtJ"] = J.T @ J
        out[name + "_pose"] = pr.get_pose(0)
        out[name + "_summary"] = np.array([s.num_iterations, s.num_successful_steps,
                                           s.termination_type, s.termination_reason])
    np.savez_compressed(os.path.join(OUT, "small_map.npz"), **out)
    print("wrote", os.path.join(OUT, "small_map.npz"), {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
