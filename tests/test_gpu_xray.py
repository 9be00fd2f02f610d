"""X-ray texture of a TSDF submap (submap_3d.cc:245-276): device bytes vs the oracle."""
import numpy as np
import pytest

from hectorgrapher_amd import synth

pytestmark = pytest.mark.gpu


def quat_axis_angle(axis, angle):
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    return np.concatenate([[np.cos(angle / 2)], np.sin(angle / 2) * axis])


def build(po, hg, ctx, res, scans, rings, cols):
    g = hg.HybridGridTSDF(ctx, res, max_blocks=1 << 16)
    og = po.Grid(res)
    for k in range(scans):
        pose = synth.pose_k(k)
        loc = synth.transform_points(pose, synth.generate_scan(pose, rings, cols, stream=k))
        hg.TSDFRangeDataInserter3D().Insert(hg.RangeData(pose[:3], loc), g)
        og.insert(pose[:3], loc)
    return g, og


@pytest.mark.parametrize("res", [0.05, 0.10, 0.20])
def test_xray_matches_oracle(po, hg, ctx, res):
    g, og = build(po, hg, ctx, res, 3, 32, 900)
    poses = [np.array([0, 0, 0, 1, 0, 0, 0], np.float64),
             np.concatenate([[0.3, -0.2, 0.1], quat_axis_angle([0, 0, 1], 0.35)]),
             np.concatenate([[-4.0, 7.5, 1.2], quat_axis_angle([1, -1, 2], 0.8)])]
    for pose in poses:
        want, want_max = og.xray(pose)
        got, got_max = g.xray(pose)
        assert got.shape == want.shape and got.shape[0] > 50 and got.shape[1] > 50
        assert np.array_equal(got_max, want_max)
        assert np.array_equal(got, want), (res, pose, int((got != want).sum()))
        assert (want[..., 1] > 0).sum() > 100  # the walls are there


def test_xray_empty_and_capacity(po, hg, ctx):
    g = hg.HybridGridTSDF(ctx, 0.1, max_blocks=1 << 10)
    cells, mx = g.xray(np.array([0, 0, 0, 1, 0, 0, 0.0]))
    assert cells.shape == (0, 0, 2)
    # voxels far from the surface only (|tsd| = tau): nothing above the obstruction limit
    g.set_cells(np.array([[1, 2, 3], [4, 5, 6]], np.int32), np.array([0.25, -0.25], np.float32),
                np.array([1.0, 1.0], np.float32))
    cells, mx = g.xray(np.array([0, 0, 0, 1, 0, 0, 0.0]))
    assert cells.shape == (0, 0, 2)
    import ctypes as C
    g2, _ = build(po, hg, ctx, 0.2, 1, 16, 300)
    L = hg._lib.load()
    pose = np.array([0, 0, 0, 1, 0, 0, 0.0])
    w, h, n = C.c_int32(), C.c_int32(), C.c_size_t()
    mx = np.zeros(2, np.int32)
    small = np.zeros(16, np.uint8)
    rc = L.hg_grid_xray(g2._h, pose.ctypes.data_as(C.c_void_p), small.ctypes.data_as(C.c_void_p), 16,
                        C.byref(w), C.byref(h), mx.ctypes.data_as(C.c_void_p), C.byref(n))
    assert rc == -4  # HG_ERR_CAPACITY, sizes still reported
    assert n.value == 2 * w.value * h.value > 16
