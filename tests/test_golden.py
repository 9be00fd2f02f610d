"""Committed golden vectors (tests/golden/small_map.npz, made by tests/golden/make_golden.py):
the oracle must reproduce them on CPU; the HIP path must reproduce them on the GPU."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "small_map.npz"))


def test_oracle_reproduces_golden_insert_and_match(po):
    grids = [po.Grid(float(r)) for r in G["resolutions"]]
    for o, s in zip(G["origins"], G["scans"]):
        for g in grids:
            g.insert(o, s)
    for i, g in enumerate(grids):
        ijk, t, w = g.export()
        assert np.array_equal(ijk, G["ijk%d" % i]) and np.array_equal(t, G["tsd%d" % i]) and np.array_equal(w, G["w%d" % i])
    pr = po.Problem()
    pr.add_pose(G["guess"])
    pr.add_block(G["query"], grids, 1.0 / np.sqrt(len(G["query"])), 0, multi_res=True)
    c, r, J, g = pr.evaluate()
    np.testing.assert_allclose(r, G["multi_residuals"], atol=1e-15)
    pr.solve()
    np.testing.assert_allclose(pr.get_pose(0), G["multi_pose"], atol=1e-12)


@pytest.mark.gpu
def test_hip_reproduces_golden(hg, ctx):
    grids = [hg.HybridGridTSDF(ctx, float(r), max_blocks=1 << 13) for r in G["resolutions"]]
    ins = [hg.TSDFRangeDataInserter3D() for _ in grids]
    for o, s in zip(G["origins"], G["scans"]):
        hg.insert_pyramid(ins, hg.RangeData(o, s), grids)
    for i, g in enumerate(grids):
        ijk, t, w = g.export()
        assert np.array_equal(ijk, G["ijk%d" % i]) and np.array_equal(t, G["tsd%d" % i]) and np.array_equal(w, G["w%d" % i])
    for name, multi, gl in (("single", False, grids[:1]), ("multi", True, grids)):
        p = hg.Problem(ctx)
        p.add_pose(G["guess"])
        p.add_block(G["query"], gl, 1.0 / np.sqrt(len(G["query"])), 0, multi_res=multi)
        c, r, g, H = p.evaluate()
        assert abs(c - G[name + "_cost"][0]) < 1e-12
        np.testing.assert_allclose(r, G[name + "_residuals"], atol=1e-13)
        np.testing.assert_allclose(g, G[name + "_gradient"], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(H, G[name + "_JtJ"], rtol=1e-9, atol=1e-13)
        s = p.solve()
        pose = p.get_pose(0)
        assert np.linalg.norm(pose[:3] - G[name + "_pose"][:3]) < 1e-4       # north_star tolerance
        assert 2 * np.arccos(min(1.0, abs(float(pose[3:] @ G[name + "_pose"][3:])))) < 1e-4
        assert [s.num_iterations, s.num_successful_steps, s.termination_type, s.termination_reason] == G[name + "_summary"].tolist()


N = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "next_rows.npz"))


def test_oracle_reproduces_golden_filters_and_xray(po):
    grid = po.Grid(float(G["resolutions"][0]))
    for o, s in zip(G["origins"], G["scans"]):
        grid.insert(o, s)
    cells, mx = grid.xray(N["xray_pose"])
    assert np.array_equal(cells, N["xray_cells"]) and np.array_equal(mx, N["xray_max_index"])
    assert np.array_equal(po.voxel_filter(0.15, N["cloud"]), N["voxel_filter_015"])
    assert np.array_equal(po.adaptive_voxel_filter(2.0, 150, 15.0, N["cloud"]), N["adaptive_high"])
    assert np.array_equal(po.adaptive_voxel_filter(4.0, 200, 60.0, N["cloud"]), N["adaptive_low"])


@pytest.mark.gpu
def test_hip_reproduces_golden_filters_and_xray(hg, ctx):
    grid = hg.HybridGridTSDF(ctx, float(G["resolutions"][0]), max_blocks=1 << 13)
    for o, s in zip(G["origins"], G["scans"]):
        hg.TSDFRangeDataInserter3D().Insert(hg.RangeData(o, s), grid)
    cells, mx = grid.xray(N["xray_pose"])
    assert np.array_equal(cells, N["xray_cells"]) and np.array_equal(mx, N["xray_max_index"])
    assert np.array_equal(hg.VoxelFilter(ctx, 0.15).Filter(N["cloud"]), N["voxel_filter_015"])
    assert np.array_equal(hg.AdaptiveVoxelFilter(ctx, 2.0, 150, 15.0).Filter(N["cloud"]), N["adaptive_high"])
    assert np.array_equal(hg.AdaptiveVoxelFilter(ctx, 4.0, 200, 60.0).Filter(N["cloud"]), N["adaptive_low"])
