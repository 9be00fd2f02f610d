"""GPU parity: HIP TSDF inserter vs the CPU oracle, bit-exact on the uint16 voxel codes."""
import os
import numpy as np
import pytest

from hectorgrapher_amd import synth

pytestmark = pytest.mark.gpu


def assert_grids_equal(og, gg):
    o_ijk, o_t, o_w = og.export()
    g_ijk, g_t, g_w = gg.export()
    assert len(o_ijk) == len(g_ijk)
    # same cells in the same (reference iterator) order, identical codes
    assert np.array_equal(o_ijk, g_ijk)
    assert np.array_equal(o_t, g_t)
    assert np.array_equal(o_w, g_w)


def test_ka1_axis_aligned_hit(po, hg, ctx):
    """SURVEY Appendix B KA-1 (x.5 rounding ties at 7.5 and 12.5)."""
    g = hg.HybridGridTSDF(ctx, 0.1, max_blocks=64)
    ins = hg.TSDFRangeDataInserter3D()
    st = ins.Insert(hg.RangeData([0, 0, 0], np.array([[1, 0, 0]], np.float32)), g)
    assert (st.num_hits, st.num_updates) == (1, 6)
    ijk = np.array([[x, 0, 0] for x in range(7, 15)], np.int32)
    t, w = g.read_cells(ijk)
    assert t.tolist() == [0, 62258, 55705, 49152, 42599, 36046, 32769, 0]
    assert w.tolist() == [0, 34, 34, 34, 34, 34, 34, 0]
    ins.Insert(hg.RangeData([0, 0, 0], np.array([[1, 0, 0]], np.float32)), g)
    t, w = g.read_cells(ijk)
    assert t.tolist() == [0, 62258, 55705, 49152, 42599, 36046, 32769, 0]
    assert w.tolist() == [0, 67, 67, 67, 67, 67, 67, 0]
    np.testing.assert_allclose(g.GetWeight(ijk[1:2]), [2.0142829], rtol=1e-6)


def test_unknown_cells(hg, ctx):
    """KA-2 / hybrid_grid_tsdf_test.cc:31-53."""
    g = hg.HybridGridTSDF(ctx, 1.0, 0.5, 1.0, max_blocks=64)
    ijk = np.array([[0, 0, 0], [0, 1, 0], [1, 0, 0], [1, 1, 0], [0, 0, 1], [0, 1, 1], [1, 0, 1],
                    [1, 1, 1]], np.int32)
    assert not g.IsKnown(ijk).any()
    g.SetCell([1, 0, 1], 0.1, 0.5)
    assert g.IsKnown([[1, 0, 1]])[0]
    assert abs(g.GetTSD([[1, 0, 1]])[0] - 0.1) < 1e-4
    assert abs(g.GetWeight([[1, 0, 1]])[0] - 0.5) < 1e-4
    assert abs(g.GetTSD([[0, 0, 1]])[0] + 0.5) < 1e-4
    assert abs(g.GetWeight([[0, 0, 1]])[0]) < 1e-4


@pytest.mark.parametrize("res", [0.05, 0.10, 0.20, 0.45])
def test_single_scan_bit_exact(po, hg, ctx, res):
    pose = synth.pose_k(3)
    pts = synth.generate_scan(pose, 16, 256, stream=3)
    loc = synth.transform_points(pose, pts)
    og = po.Grid(res)
    gg = hg.HybridGridTSDF(ctx, res, max_blocks=1 << 15)
    n_in, u = og.insert(pose[:3], loc)
    st = hg.TSDFRangeDataInserter3D().Insert(hg.RangeData(pose[:3], loc), gg)
    assert (st.num_hits, st.num_updates) == (n_in, u)
    assert_grids_equal(og, gg)


def test_accumulated_scans_bit_exact(po, hg, ctx):
    """10 scans into one grid: order-dependent re-quantisation must match after every scan."""
    from conftest import build_map
    og, gg = build_map(po, (ctx, hg), [0.10], 16, 625, 10)
    assert_grids_equal(og[0], gg[0])


def test_weight_saturation(po, hg, ctx):
    """Same hit 1100 times: weights saturate at maximum_weight (2-D precedent:
    tsdf_range_data_inserter_2d_test.cc:118-143)."""
    og = po.Grid(0.1)
    gg = hg.HybridGridTSDF(ctx, 0.1, max_blocks=64)
    pts = np.tile(np.array([[1.03, 0.02, -0.01]], np.float32), (1100, 1))
    og.insert([0, 0, 0], pts)
    hg.TSDFRangeDataInserter3D().Insert(hg.RangeData([0, 0, 0], pts), gg)
    assert_grids_equal(og, gg)
    _, w = gg.export()[1:]
    assert w.max() == 32767


def test_gates_nan_range_ratio(po, hg, ctx):
    rng = np.random.default_rng(7)
    pts = (rng.standard_normal((4000, 3)) * 6).astype(np.float32)
    pts[::37] = np.nan
    pts[5] = [0.01, 0.0, 0.0]     # below min_range
    pts[6] = [100.0, 0.0, 0.0]    # beyond max_range
    kw = dict(insertion_ratio=0.1, min_range=1.0, max_range=12.0)
    og = po.Grid(0.2)
    gg = hg.HybridGridTSDF(ctx, 0.2, max_blocks=1 << 14)
    a = og.insert([0.1, -0.2, 0.3], pts, po.InsertOpts(**kw))
    st = hg.TSDFRangeDataInserter3D(hg.InsertOpts(**kw)).Insert(hg.RangeData([0.1, -0.2, 0.3], pts), gg)
    assert (st.num_hits, st.num_updates) == a
    assert_grids_equal(og, gg)


def test_submap_pose_transform(po, hg, ctx):
    """Submap3D::InsertData frame change (submap_3d.cc:436-437) fused on the device."""
    pose = synth.pose_k(5)
    pts = synth.generate_scan(pose, 8, 128, stream=5)
    loc = synth.transform_points(pose, pts)
    inv = synth.pose_inverse(synth.pose_k(2)).astype(np.float32)
    og = po.Grid(0.1)
    gg = hg.HybridGridTSDF(ctx, 0.1, max_blocks=1 << 14)
    og.insert(pose[:3], loc, pose_tq=inv)
    hg.TSDFRangeDataInserter3D().Insert(hg.RangeData(pose[:3], loc), gg, pose_tq=inv)
    assert_grids_equal(og, gg)


def test_free_space_and_weight_dropoff(po, hg, ctx):
    """num_free_space_voxels > 0 walks from the origin (:303-307); epsilon < 1 enables the
    exponential weight (:333-340, double exp => codes compared with 1 LSB slack on weight)."""
    pose = synth.pose_k(1)
    pts = synth.generate_scan(pose, 4, 64, stream=9)
    kw = dict(num_free_space_voxels=1)
    og = po.Grid(0.2)
    gg = hg.HybridGridTSDF(ctx, 0.2, max_blocks=1 << 15)
    og.insert([0, 0, 0], pts, po.InsertOpts(**kw))
    hg.TSDFRangeDataInserter3D(hg.InsertOpts(**kw)).Insert(hg.RangeData([0, 0, 0], pts), gg)
    assert_grids_equal(og, gg)
    kw = dict(weight_function_epsilon=0.5)
    og = po.Grid(0.2)
    gg = hg.HybridGridTSDF(ctx, 0.2, max_blocks=1 << 15)
    og.insert([0, 0, 0], pts, po.InsertOpts(**kw))
    hg.TSDFRangeDataInserter3D(hg.InsertOpts(**kw)).Insert(hg.RangeData([0, 0, 0], pts), gg)
    o_ijk, o_t, o_w = og.export()
    g_ijk, g_t, g_w = gg.export()
    assert np.array_equal(o_ijk, g_ijk)
    assert np.abs(o_t.astype(int) - g_t.astype(int)).max() <= 1
    assert np.abs(o_w.astype(int) - g_w.astype(int)).max() <= 1


def test_batch_equals_sequential(po, hg, ctx):
    """Stream form (hg_grid_insert_batch) == scans inserted one at a time == oracle."""
    res = 0.1
    scans, origins, offs = [], [], [0]
    for k in range(6):
        pose = synth.pose_k(k)
        pts = synth.transform_points(pose, synth.generate_scan(pose, 8, 200, stream=k))
        scans.append(pts)
        origins.append(pose[:3])
        offs.append(offs[-1] + len(pts))
    og = po.Grid(res)
    for o, p in zip(origins, scans):
        og.insert(o, p)
    gg = hg.HybridGridTSDF(ctx, res, max_blocks=1 << 14)
    hg.TSDFRangeDataInserter3D().InsertBatch(np.array(origins), np.concatenate(scans), offs, gg)
    assert_grids_equal(og, gg)


def test_empty_and_capacity(hg, ctx):
    g = hg.HybridGridTSDF(ctx, 0.1, max_blocks=4)
    st = hg.TSDFRangeDataInserter3D().Insert(hg.RangeData([0, 0, 0], np.zeros((0, 3), np.float32)), g)
    assert st.num_updates == 0 and g.count() == 0
    pts = synth.generate_scan(synth.pose_k(0), 8, 128)
    with pytest.raises(hg.HgError):
        hg.TSDFRangeDataInserter3D().Insert(hg.RangeData([0, 0, 0], pts), g)


def test_device_resident_input(po, hg, ctx):
    torch = pytest.importorskip("torch")
    pose = synth.pose_k(0)
    pts = synth.generate_scan(pose, 16, 625)
    og = po.Grid(0.05)
    og.insert([0, 0, 0], pts)
    gg = hg.HybridGridTSDF(ctx, 0.05, max_blocks=1 << 16)
    d = torch.from_numpy(pts).to("cuda:0")
    torch.cuda.synchronize()
    hg.TSDFRangeDataInserter3D().Insert(hg.RangeData([0, 0, 0], d), gg)
    assert_grids_equal(og, gg)


def test_pyramid_insert_high_low_res_options(po, hg, ctx):
    """Submap3D::InsertData: same range data through the high- and low-resolution inserters
    (trajectory_builder_3d.lua:78-93 vs :100-115: different min/max range and insertion_ratio),
    fused into one device pass."""
    hi = dict(min_range=0.4, max_range=15.0, insertion_ratio=1.0)
    lo = dict(min_range=1.0, max_range=60.0, insertion_ratio=0.1)
    res = [0.10, 0.45]
    og = [po.Grid(r) for r in res]
    gg = [hg.HybridGridTSDF(ctx, r, max_blocks=1 << 15) for r in res]
    ins = [hg.TSDFRangeDataInserter3D(hg.InsertOpts(**hi)), hg.TSDFRangeDataInserter3D(hg.InsertOpts(**lo))]
    for k in range(4):
        pose = synth.pose_k(k)
        loc = synth.transform_points(pose, synth.generate_scan(pose, 16, 300, stream=k))
        a0 = og[0].insert(pose[:3], loc, po.InsertOpts(**hi))
        a1 = og[1].insert(pose[:3], loc, po.InsertOpts(**lo))
        st = hg.insert_pyramid(ins, hg.RangeData(pose[:3], loc), gg)
        assert (st[0].num_hits, st[0].num_updates) == a0
        assert (st[1].num_hits, st[1].num_updates) == a1
    for o, g in zip(og, gg):
        assert_grids_equal(o, g)


def test_pyramid_three_levels_async_and_status(po, hg, ctx):
    res = [0.05, 0.10, 0.20]
    og = [po.Grid(r) for r in res]
    gg = [hg.HybridGridTSDF(ctx, r, max_blocks=1 << 16) for r in res]
    ins = [hg.TSDFRangeDataInserter3D() for _ in res]
    for k in range(3):
        pose = synth.pose_k(k)
        pts = synth.generate_scan(pose, 16, 625, stream=k)
        for o in og:
            o.insert(pose[:3], synth.transform_points(pose, pts))
        # sensor-frame points + pose applied on the device, no stats => no synchronisation
        assert hg.insert_pyramid(ins, hg.RangeData([0, 0, 0], pts), gg,
                                 pose_tq=pose.astype(np.float32), want_stats=False) is None
    for o, g in zip(og, gg):
        g.status()
        assert_grids_equal(o, g)


def test_far_points_use_wide_keys(po, hg, ctx):
    """max_range large enough that the 32-bit key window does not fit: 64-bit key path."""
    rng = np.random.default_rng(3)
    d = rng.standard_normal((3000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = (d * rng.uniform(1.0, 55.0, (3000, 1))).astype(np.float32)
    kw = dict(max_range=60.0)
    og = po.Grid(0.05)
    gg = hg.HybridGridTSDF(ctx, 0.05, max_blocks=1 << 16)
    a = og.insert([0, 0, 0], pts, po.InsertOpts(**kw))
    st = hg.TSDFRangeDataInserter3D(hg.InsertOpts(**kw)).Insert(hg.RangeData([0, 0, 0], pts), gg)
    assert (st.num_hits, st.num_updates) == a
    assert_grids_equal(og, gg)


def test_block_arrays_roundtrip_through_torch(po, hg, ctx):
    """The multi-GPU gather ships (keys, voxels) of the block pool as torch tensors aliasing library
    memory; importing them into a fresh grid must reproduce the grid exactly."""
    torch = pytest.importorskip("torch")
    from hectorgrapher_amd import distributed as hgd
    pose = synth.pose_k(2)
    pts = synth.transform_points(pose, synth.generate_scan(pose, 16, 300, stream=2))
    src = hg.HybridGridTSDF(ctx, 0.1, max_blocks=1 << 14)
    hg.TSDFRangeDataInserter3D().Insert(hg.RangeData(pose[:3], pts), src)
    ctx.synchronize()
    keys, vox = hgd.grid_block_tensors(src, torch.device("cuda:0"))
    assert keys.shape[0] == src.num_blocks() and vox.shape == (keys.shape[0], 512)
    k2, v2 = keys.clone(), vox.clone()
    torch.cuda.synchronize()
    dst = hg.HybridGridTSDF(ctx, 0.1, max_blocks=1 << 14)
    dst.import_blocks(k2, v2)
    a, b = src.export(), dst.export()
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # host path too
    dst2 = hg.HybridGridTSDF(ctx, 0.1, max_blocks=1 << 14)
    dst2.import_blocks(k2.cpu().numpy().view(np.uint64), v2.cpu().numpy().view(np.uint32))
    assert all(np.array_equal(x, y) for x, y in zip(a, dst2.export()))


@pytest.mark.parametrize("res,hs,vs", [(0.10, 5, 1), (0.20, 20, 4), (0.05, 2, 2)])
def test_project_sdf_to_cloud_structure_normals(po, hg, ctx, res, hs, vs):
    """a7: project_sdf_distance_to_scan_normal with CLOUD_STRUCTURE normals (:502-607) and
    InsertHitWithNormal (:197-241) on the structured synthetic cloud (width = rings)."""
    rings, cols = 16, 400
    kw = dict(project_sdf_distance_to_scan_normal=1, normal_computation_method=1,
              normal_computation_horizontal_stride=hs, normal_computation_vertical_stride=vs)
    og = po.Grid(res)
    gg = hg.HybridGridTSDF(ctx, res, max_blocks=1 << 15)
    ins = hg.TSDFRangeDataInserter3D(hg.InsertOpts(**kw))
    assert ins.RequiresStructuredData()
    for k in range(3):
        pose = synth.pose_k(k)
        pts = synth.generate_scan(pose, rings, cols, stream=k)
        pts[::97] = np.nan
        loc = synth.transform_points(pose, pts)
        a = og.insert(pose[:3], loc, po.InsertOpts(**kw), width=rings)
        st = ins.Insert(hg.RangeData(pose[:3], loc, width=rings), gg)
        assert (st.num_hits, st.num_updates) == a and a[1] > 1000
    assert_grids_equal(og, gg)
    # PCL / Open3D / TRIANGLE_FILL_IN normals are refused, not approximated
    bad = hg.TSDFRangeDataInserter3D(hg.InsertOpts(project_sdf_distance_to_scan_normal=1,
                                                   normal_computation_method=0))
    with pytest.raises(hg.HgError):
        bad.Insert(hg.RangeData([0, 0, 0], pts, width=rings), gg)


def test_thousands_of_rays_through_one_voxel(po, hg, ctx):
    """Degenerate geometry: 5000 near-identical returns put > 2048 updates (one LDS pass of the
    binned path) on single voxels; the chain is then applied in seq-ordered rounds."""
    rng = np.random.default_rng(11)
    pts = (np.array([[2.01, 0.03, -0.02]], np.float32) + (rng.standard_normal((5000, 3)) * 2e-3).astype(np.float32))
    for res in (0.05, 0.2):
        og = po.Grid(res)
        gg = hg.HybridGridTSDF(ctx, res, max_blocks=256)
        a = og.insert([0, 0, 0], pts)
        st = hg.TSDFRangeDataInserter3D().Insert(hg.RangeData([0, 0, 0], pts), gg)
        assert (st.num_hits, st.num_updates) == a
        assert_grids_equal(og, gg)


def test_giant_voxel_rounds_with_gaps_and_dense_runs(po, hg, ctx):
    """Giant-voxel rounds of the binned path: 20000 returns through the same voxels with two far
    apart index clusters (empty seq buckets between them) embedded in an ordinary scan."""
    rng = np.random.default_rng(5)
    hot = np.array([[1.51, -0.52, 0.11]], np.float32)
    cluster = lambda m: hot + (rng.standard_normal((m, 3)) * 1.5e-3).astype(np.float32)
    filler = synth.generate_scan(synth.pose_k(0), 16, 500, stream=3)
    pts = np.concatenate([cluster(9000), filler[:4000], cluster(300), filler[4000:], cluster(11000)])
    for res in (0.05, 0.2):
        og = po.Grid(res)
        gg = hg.HybridGridTSDF(ctx, res, max_blocks=1 << 14)
        a = og.insert([0, 0, 0], pts)
        st = hg.TSDFRangeDataInserter3D().Insert(hg.RangeData([0, 0, 0], pts), gg)
        assert (st.num_hits, st.num_updates) == a
        assert_grids_equal(og, gg)


def test_small_block_pool_dense_scan(po, hg, ctx):
    """The apply work list is sized from the records of the call, not from the block pool: a coarse
    grid with a small pool takes a dense scan (about 10^5 returns -> 6 * 10^5 records, thousands of
    work items) without losing an update."""
    pose = synth.pose_k(4)
    pts = synth.generate_scan(pose, 50, 2000, stream=4)
    loc = synth.transform_points(pose, pts)
    for res, pool in ((0.2, 1024), (0.45, 256)):
        og = po.Grid(res)
        gg = hg.HybridGridTSDF(ctx, res, max_blocks=pool)
        a = og.insert(pose[:3], loc)
        st = hg.TSDFRangeDataInserter3D().Insert(hg.RangeData(pose[:3], loc), gg)
        assert (st.num_hits, st.num_updates) == a
        assert st.num_blocks <= pool
        assert_grids_equal(og, gg)


def test_more_than_2_pow_20_records_in_one_block(po, hg, ctx):
    """190 000 returns whose whole truncation band lies inside ONE 8^3 block of a 0.2 m grid: the
    block's bin holds more than 2^20 records (the record count of a work item is a full 32-bit
    word); tens of thousands of updates per voxel run through the giant-voxel rounds."""
    rng = np.random.default_rng(23)
    # block [0, 1.6)^3 m holds cells 0..7; rays along +x end at x = 0.8 +- 0.5 m
    hot = np.array([[0.8, 0.81, 0.79]], np.float32)
    pts = hot + (rng.standard_normal((190000, 3)) * 4e-3).astype(np.float32)
    origin = np.array([-3.0, 0.8, 0.8], np.float32)
    og = po.Grid(0.2)
    gg = hg.HybridGridTSDF(ctx, 0.2, max_blocks=64)
    a = og.insert(origin, pts)
    st = hg.TSDFRangeDataInserter3D().Insert(hg.RangeData(origin, pts), gg)
    assert a[1] > (1 << 20)
    assert (st.num_hits, st.num_updates) == a
    assert st.num_blocks == 1
    assert_grids_equal(og, gg)


def test_batch_with_empty_leading_scans_counts_only_this_call(po, hg, ctx):
    """hg_insert_stats of a batch whose first scans are empty: the counters restart with the first
    chunk that runs instead of adding onto the previous call's."""
    res = 0.1
    pose = synth.pose_k(1)
    pts = synth.transform_points(pose, synth.generate_scan(pose, 8, 200, stream=1))
    gg = hg.HybridGridTSDF(ctx, res, max_blocks=1 << 14)
    ins = hg.TSDFRangeDataInserter3D()
    first = ins.Insert(hg.RangeData(pose[:3], pts), gg)
    og = po.Grid(res)
    og.insert(pose[:3], pts)
    n_in, u = og.insert(pose[:3], pts)
    origins = np.array([pose[:3], pose[:3], pose[:3]], np.float32)
    st = ins.InsertBatch(origins, pts, [0, 0, 0, len(pts)], gg)
    assert (st.num_hits, st.num_updates) == (n_in, u)
    assert (first.num_hits, first.num_updates) == (n_in, u)
    assert_grids_equal(og, gg)


def test_async_insert_errors_reach_the_next_call(hg, ctx):
    """Inserts without a stats read-back (hg_pyramid_insert(stats = NULL), the insertion of
    hg_register_scan) leave their sticky error flags in a host-mapped mailbox: the next such call and
    hg_ctx_synchronize return the error instead of HG_OK."""
    g = hg.HybridGridTSDF(ctx, 0.1, max_blocks=4)
    ins = [hg.TSDFRangeDataInserter3D()]
    pts = synth.generate_scan(synth.pose_k(0), 8, 128)
    assert hg.insert_pyramid(ins, hg.RangeData([0, 0, 0], pts), [g], want_stats=False) is None
    with pytest.raises(hg.HgError, match="HG_ERR_CAPACITY"):
        ctx.synchronize()
    with pytest.raises(hg.HgError, match="HG_ERR_CAPACITY"):
        hg.insert_pyramid(ins, hg.RangeData([0, 0, 0], pts), [g], want_stats=False)
    with pytest.raises(hg.HgError, match="HG_ERR_CAPACITY"):
        g.status()
    g.clear()        # a cleared grid starts without sticky errors
    ctx.synchronize()
    g.close()


def _stream_call(hg, grids, scans, mode_exact=True):
    """hg_pyramid_insert_batch of `scans` [(pose, points in the sensor frame)] into `grids`."""
    import ctypes as C
    import torch
    from hectorgrapher_amd import _lib
    B = len(scans)
    xyz = torch.from_numpy(np.concatenate([p for _, p in scans])).to(torch.device("cuda", 0))
    poses = np.array([pose for pose, _ in scans], np.float32)
    origins = np.zeros((B, 3), np.float32)
    offs = np.concatenate([[0], np.cumsum([len(p) for _, p in scans])]).astype(np.uint64)
    L = _lib.load()
    n = len(grids)
    garr = (C.c_void_p * n)(*[g._h for g in grids])
    opts = (hg.InsertOpts * n)(*[hg.InsertOpts() for _ in grids])
    st = (hg.InsertStats * n)()
    hg.check(L.hg_pyramid_insert_batch(garr, opts, n, origins.ctypes.data_as(C.c_void_p), xyz.data_ptr(),
                                       offs.ctypes.data_as(C.c_void_p), B, 0, poses.ctypes.data_as(C.c_void_p),
                                       _lib.HG_INSERT_EXACT, 1, st), "hg_pyramid_insert_batch")
    return st


def test_grouped_stream_calls_of_different_layouts_on_one_context(po, hg):
    """The grouped scan stream keeps per-scan bin arrays in a context workspace that is all-zero between
    calls of ONE layout. Calls that lay it out differently (scan sizes, number of levels, pool size) on
    the same context must not see each other's leftovers: every call equals the oracle inserting its
    scans one after the other."""
    c = hg.Context(0)
    try:
        def scans_of(sizes, first):
            out = []
            for j, (rings, cols) in enumerate(sizes):
                pose = synth.pose_k(first + j)
                out.append((pose, synth.generate_scan(pose, rings, cols, stream=900 + first + j)))
            return out

        def oracle(ogrids, scans):
            for pose, pts in scans:
                loc = synth.transform_points(pose, pts)
                for g in ogrids:
                    g.insert(pose[:3].astype(np.float32), loc)

        res3 = [0.05, 0.10, 0.20]
        pyr = [hg.HybridGridTSDF(c, r, max_blocks=1 << 15) for r in res3]
        opyr = [po.Grid(r) for r in res3]
        single = [hg.HybridGridTSDF(c, 0.10, max_blocks=1 << 13)]
        osingle = [po.Grid(0.10)]
        # 1: ten 40k-point scans, three levels  2: nine 20k-point scans into ONE other grid (smaller pool)
        # 3: scans of 17k..48k points into the pyramid again  4: the single grid again with larger scans
        calls = [(pyr, opyr, scans_of([(16, 2500)] * 10, 0)),
                 (single, osingle, scans_of([(10, 2000)] * 9, 3)),
                 (pyr, opyr, scans_of([(16, 1100), (16, 3000), (20, 1500), (12, 1500), (16, 2000), (24, 2000)], 10)),
                 (single, osingle, scans_of([(16, 2500)] * 5, 12))]
        for gg, og, scans in calls:
            _stream_call(hg, gg, scans)
            oracle(og, scans)
            for o, g in zip(og, gg):
                assert_grids_equal(o, g)
    finally:
        c.close()


def test_stream_over_different_rooms_with_a_long_union(po, hg):
    """Sixteen 40k-point scans inserted 60 m apart from each other in ONE stream call: no block is shared, the union of
    the group's blocks holds ~30 000 entries on the finest level -- k_stream_units then gives half a wavefront FOUR
    blocks per batch (the short unions of the other stream tests take one) -- and most blocks live in the overflow area
    of the pool. Every voxel equals the oracle inserting the scans one after the other."""
    c = hg.Context(0)
    try:
        res3 = [0.05, 0.10, 0.20]
        pyr = [hg.HybridGridTSDF(c, r, max_blocks=1 << 16) for r in res3]
        opyr = [po.Grid(r) for r in res3]
        scans = []
        for j in range(16):
            pose = synth.pose_k(j % 5).copy()
            pts = synth.generate_scan(pose, 16, 2500, stream=1300 + j)
            pose[0] += 60.0 * (j % 4)
            pose[1] += 60.0 * (j // 4)
            scans.append((pose, pts))
        _stream_call(hg, pyr, scans)
        for pose, pts in scans:
            loc = synth.transform_points(pose, pts)
            for g in opyr:
                g.insert(pose[:3].astype(np.float32), loc)
        for o, g in zip(opyr, pyr):
            assert_grids_equal(o, g)
        assert pyr[0].num_blocks() >= 16 * 512  # (the union was long enough for the four-blocks path: 32 blocks per workgroup)
    finally:
        c.close()


def test_async_insert_errors_are_per_grid(hg):
    """A full grid reports its sticky error through calls that work on IT (and hg_ctx_synchronize); other
    grids of the context keep working, and clearing one of them does not drop the full grid's error."""
    c = hg.Context(0)
    try:
        full = hg.HybridGridTSDF(c, 0.1, max_blocks=4)
        fine = hg.HybridGridTSDF(c, 0.1, max_blocks=1 << 12)
        ins = [hg.TSDFRangeDataInserter3D()]
        pts = synth.generate_scan(synth.pose_k(0), 8, 128)
        assert hg.insert_pyramid(ins, hg.RangeData([0, 0, 0], pts), [full], want_stats=False) is None
        with pytest.raises(hg.HgError, match="HG_ERR_CAPACITY"):
            c.synchronize()
        # the other grid is not concerned ...
        assert hg.insert_pyramid(ins, hg.RangeData([0, 0, 0], pts), [fine], want_stats=False) is None
        fine.status()
        fine.clear()  # ... and clearing it leaves the full grid's error in place
        with pytest.raises(hg.HgError, match="HG_ERR_CAPACITY"):
            hg.insert_pyramid(ins, hg.RangeData([0, 0, 0], pts), [full], want_stats=False)
        with pytest.raises(hg.HgError, match="HG_ERR_CAPACITY"):
            full.status()
        full.close()  # a destroyed grid takes its error with it
        c.synchronize()
        fine.close()
    finally:
        c.close()


def test_children_may_outlive_their_context(hg):
    """A garbage-collected host destroys handles in any order: a context destroyed before its grids and
    problems orphans them (hg_ctx_destroy), and destroying them afterwards only frees their memory."""
    from hectorgrapher_amd import _lib
    import ctypes as C
    L = _lib.load()
    c = hg.Context(0)
    g = hg.HybridGridTSDF(c, 0.1, max_blocks=256)
    p = hg.Problem(c)
    gh, ph = g._h, p._h
    g._h = p._h = None           # keep the Python wrappers from closing them first
    c._children = set()
    c.close()
    assert L.hg_grid_destroy(gh) == 0
    assert L.hg_problem_destroy(ph) == 0


@pytest.mark.parametrize("env", [{"HG_STREAM_SLICE": "512"}, {"HG_STREAM_SLICE": "2048"}, {"HG_APPLY_TURNS": "1"},
                                 {"HG_STREAM_GROUP": "3"}, {"HG_STREAM_MERGE": "0"}])
def test_apply_schedules_are_all_bit_exact(env):
    """The slice size of a scan stream's large bins and the dispatch order of k_bin_apply are schedules, not
    semantics: the tests that compare grouped streams, giant voxels and dense pools with the oracle pass under the
    alternative settings too -- among them (round 6) the MERGED apply of a stream's groups (the default: units of
    (block, voxel slice) x scans) with groups of three and switched off (the switches are the defaults of a context's
    options, read from the environment when it is created: hence a child process)."""
    import subprocess
    import sys
    child_env = dict(os.environ)
    child_env.update(env)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k",
                          "grouped_stream or long_union or thousands_of_rays or giant_voxel or small_block_pool or batch_equals or 2_pow_20"],
                         env=child_env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert " passed" in out.stdout
