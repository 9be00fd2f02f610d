"""Diagnostic: worst voxels of the tolerance insert mode against the exact oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import pyoracle as po
from hectorgrapher_amd import api as hg, synth
from test_gpu_insert_fast import decode, sorted_cells

ctx = hg.Context(0)
res = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
tau = float(np.float32(2.5 * res))
pose = synth.pose_k(2)
loc = synth.transform_points(pose, synth.generate_scan(pose, 50, 2000, stream=2))
og = po.Grid(res); og.insert(pose[:3], loc)
g = hg.HybridGridTSDF(ctx, res, max_blocks=1 << 16)
hg.TSDFRangeDataInserter3D(mode=hg._lib.HG_INSERT_FAST).Insert(hg.RangeData(pose[:3], loc), g)
a = sorted_cells(*og.export()); b = sorted_cells(*g.export())
ta, wa = decode(*a, tau, 1000.0); tb, wb = decode(*b, tau, 1000.0)
m = np.maximum(1, np.round(wb)); dt = np.abs(ta - tb)
idx = np.argsort(-(dt / m))[:12]
for i in idx:
    print(a[0][i], "m=%d w_exact=%.3f w_fast=%.3f tsd_exact=%.6f tsd_fast=%.6f d=%.2e" % (m[i], wa[i], wb[i], ta[i], tb[i], dt[i]))
print("quantum", 2 * tau / 32766, "max dt", dt.max(), "mean dt", dt.mean(), "share above 1e-4*m", np.mean(dt > 1e-4 * m))
