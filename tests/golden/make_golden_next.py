#!/usr/bin/env python3
"""Generates tests/golden/next_rows.npz from the CPU oracle: fixtures for the rows either side of
the hot path — voxel filters and the x-ray texture — over the grids of
small_map.npz. Run from the repo root:  python tests/golden/make_golden_next.py"""
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
    G = np.load(os.path.join(OUT, "small_map.npz"))
    grid = po.Grid(float(G["resolutions"][0]))
    for o, s in zip(G["origins"], G["scans"]):
        grid.insert(o, s)
    pose = np.array([0.4, -0.3, 0.2, 0.98480775, 0.0, 0.0, 0.17364818])  # 20 degrees about z
    cells, max_index = grid.xray(pose)
    cloud = synth.generate_scan(synth.pose_k(5), 16, 625, stream=55)
    out = {
        "xray_pose": pose, "xray_cells": cells, "xray_max_index": max_index,
        "cloud": cloud,
        "voxel_filter_015": po.voxel_filter(0.15, cloud),
        "adaptive_high": po.adaptive_voxel_filter(2.0, 150, 15.0, cloud),
        "adaptive_low": po.adaptive_voxel_filter(4.0, 200, 60.0, cloud),
    }
    np.savez_compressed(os.path.join(OUT, "next_rows.npz"), **out)
    print("wrote next_rows.npz", {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
