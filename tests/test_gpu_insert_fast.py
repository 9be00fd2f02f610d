"""HG_INSERT_FAST (tolerance mode): the updates a voxel receives in one call are summed and applied
once. Checked against the exact restatement of the reference (oracle).

Weight codes are identical to the reference's (its code sequence has a closed form). Tolerance on the
tsd, stated and tested here: |d tsd| <= 5e-3 * tau per call (1e-2 * tau over a sequence of calls) + half
a tsd code per update of the voxel.
The first term is inherent to any order-free scheme: the reference re-quantises the WEIGHT after
every update (weight 1 is stored as 1.0071), so its running mean weighs the k-th update by a factor
that depends on k (up to 0.18 % of the spread of the updates, i.e. of 2 tau); the second is the
tsd re-quantisation noise the reference accumulates. BASELINE.md's first guess for this mode
(1e-4 m per update) holds for 99.8 % of the voxels of a 0.05 m grid; the test reports the share."""
import numpy as np
import pytest

from hectorgrapher_amd import synth

pytestmark = pytest.mark.gpu

RES = (0.05, 0.10, 0.20)


def decode(ijk, t, w, tau, max_weight):
    """TSDValueConverter decode of raw codes (value_conversion_tables.cc:29-68)."""
    tv = (t & 0x7FFF).astype(np.float64)
    wv = (w & 0x7FFF).astype(np.float64)
    ks = 2 * tau / 32766.0
    kw = max_weight / 32766.0
    return tv * ks + (-tau - ks), wv * kw - kw


def sorted_cells(ijk, t, w):
    key = np.lexsort((ijk[:, 0], ijk[:, 1], ijk[:, 2]))
    return ijk[key], t[key], w[key]


def compare(po_grid, hg_grid, tau, max_weight, calls=1):
    a = sorted_cells(*po_grid.export())
    b = sorted_cells(*hg_grid.export())
    assert np.array_equal(a[0], b[0])                      # the same voxels are touched
    assert np.array_equal(a[2], b[2])                      # weight codes: identical (the code sequence is closed-form)
    ta, wa = decode(*a, tau, max_weight)
    tb, wb = decode(*b, tau, max_weight)
    m = np.maximum(1.0, np.round(wb))                      # updates the voxel has seen (fresh grid: its weight)
    dt = np.abs(ta - tb)
    tsd_code = 2 * tau / 32766.0
    assert np.all(dt <= 5e-3 * tau * min(calls, 2) + 0.5 * tsd_code * m), float((dt / tau).max())
    return float(dt.max()), float(np.mean(dt > 1e-4 * m))


@pytest.mark.parametrize("level", [0, 1, 2])
def test_fast_single_scan_within_tolerance(po, hg, ctx, level):
    res = RES[level]
    tau = float(np.float32(2.5 * res))
    pose = synth.pose_k(2)
    loc = synth.transform_points(pose, synth.generate_scan(pose, 50, 2000, stream=2))
    og = po.Grid(res)
    og.insert(pose[:3], loc)
    g = hg.HybridGridTSDF(ctx, res, max_blocks=1 << 16)
    fast = hg.TSDFRangeDataInserter3D(mode=hg._lib.HG_INSERT_FAST)
    st = fast.Insert(hg.RangeData(pose[:3], loc), g)
    nin, u = po.Grid(res).insert(pose[:3], loc)
    assert (st.num_hits, st.num_updates) == (nin, u)        # the same rays, the same update count
    worst, share_above_first_guess = compare(og, g, tau, 1000.0)
    assert share_above_first_guess < (0.01 if level < 2 else 0.05)   # BASELINE.md's first guess, 1e-4 m per update


def test_fast_is_order_free_and_deterministic(hg, ctx):
    """Integer sums commute: any arrival order and any order of the returns give the same codes
    (the exact mode, like the reference, depends on the order of the returns)."""
    pose = synth.pose_k(4)
    loc = synth.transform_points(pose, synth.generate_scan(pose, 32, 1000, stream=4))
    rng = np.random.default_rng(3)
    fast = hg.TSDFRangeDataInserter3D(mode=hg._lib.HG_INSERT_FAST)
    outs = []
    for perm in (np.arange(len(loc)), rng.permutation(len(loc)), np.arange(len(loc))[::-1]):
        g = hg.HybridGridTSDF(ctx, 0.2, max_blocks=1 << 14)
        fast.Insert(hg.RangeData(pose[:3], np.ascontiguousarray(loc[perm])), g)
        outs.append(sorted_cells(*g.export()))
    for o in outs[1:]:
        assert all(np.array_equal(x, y) for x, y in zip(outs[0], o))


def test_fast_pyramid_sequence_and_batch(po, hg, ctx):
    """Six scans into the 3-level pyramid: scan by scan (fused levels) and as one batched call."""
    scans = []
    for k in range(6):
        pose = synth.pose_k(k)
        scans.append((pose, synth.generate_scan(pose, 32, 1000, stream=k)))
    ogs = [po.Grid(r) for r in RES]
    for pose, pts in scans:
        loc = synth.transform_points(pose, pts)
        for og in ogs:
            og.insert(pose[:3], loc)
    fast = [hg.TSDFRangeDataInserter3D(mode=hg._lib.HG_INSERT_FAST) for _ in RES]
    seq = [hg.HybridGridTSDF(ctx, r, max_blocks=1 << 16) for r in RES]
    for pose, pts in scans:
        hg.insert_pyramid(fast, hg.RangeData([0, 0, 0], pts), seq, pose_tq=pose.astype(np.float32))
    for og, g, r in zip(ogs, seq, RES):
        compare(og, g, float(np.float32(2.5 * r)), 1000.0, calls=len(scans))
    # one call for all scans: every voxel's updates of the whole batch in one closed form
    bat = hg.HybridGridTSDF(ctx, 0.1, max_blocks=1 << 16)
    xyz = np.concatenate([p for _, p in scans])
    offs = np.arange(len(scans) + 1, dtype=np.uint64) * len(scans[0][1])
    poses = np.array([p for p, _ in scans], np.float32)
    st = fast[0].InsertBatch(np.zeros((len(scans), 3), np.float32), xyz, offs, bat, poses_tq=poses)
    assert st.num_updates > 0 and st.num_hits > 0
    compare(ogs[1], bat, 0.25, 1000.0, calls=len(scans))


def test_fast_saturating_weight(po, hg, ctx):
    """maximum_weight = 20: the clamp engages after a few scans and UpdateCell turns into a moving
    average; the closed form (all updates of a call at their mean) stays inside the tolerance."""
    opts_o = po.InsertOpts(maximum_weight=20.0)
    opts_h = hg.InsertOpts()
    opts_h.maximum_weight = 20.0
    og = po.Grid(0.2, max_weight=20.0)
    g = hg.HybridGridTSDF(ctx, 0.2, max_weight=20.0, max_blocks=1 << 14)
    fast = hg.TSDFRangeDataInserter3D(opts_h, mode=hg._lib.HG_INSERT_FAST)
    for k in range(8):
        pose = synth.pose_k(k)
        loc = synth.transform_points(pose, synth.generate_scan(pose, 16, 625, stream=k))
        og.insert(pose[:3], loc, opts=opts_o)
        fast.Insert(hg.RangeData(pose[:3], loc), g)
    a = sorted_cells(*og.export())
    b = sorted_cells(*g.export())
    assert np.array_equal(a[0], b[0])
    assert np.array_equal(a[2], b[2])                      # weight codes identical, clamp included
    ta, _ = decode(*a, 0.5, 20.0)
    tb, _ = decode(*b, 0.5, 20.0)
    # a saturated voxel forgets: every update of a call weighs 1/21 of the voxel, so the ORDER of a
    # call's updates matters to the reference; the order-free form keeps their mean (bounded by the
    # spread of the updates inside one voxel and call)
    assert np.abs(ta - tb).max() < 0.05 and np.mean(np.abs(ta - tb)) < 1e-3


def test_fast_refuses_options_it_cannot_honour(hg, ctx):
    g = hg.HybridGridTSDF(ctx, 0.1, max_blocks=1 << 10)
    o = hg.InsertOpts()
    o.weight_function_epsilon = 0.5                       # non-unit weights: exact mode only
    pts = np.array([[3.0, 0.1, 0.2]], np.float32)
    with pytest.raises(hg.HgError):
        hg.TSDFRangeDataInserter3D(o, mode=hg._lib.HG_INSERT_FAST).Insert(hg.RangeData([0, 0, 0], pts), g)


def test_register_scan_with_fast_insert(po, hg, ctx):
    """hg_register_scan_mode(HG_INSERT_FAST): the match is the one of the exact step (same map, same
    solver), the scan is inserted at the solved pose taken from device memory, in tolerance mode."""
    import torch
    dev = torch.device("cuda", 0)
    exact = [hg.HybridGridTSDF(ctx, r, max_blocks=1 << 16) for r in RES]
    fast = [hg.HybridGridTSDF(ctx, r, max_blocks=1 << 16) for r in RES]
    ins_e = [hg.TSDFRangeDataInserter3D() for _ in RES]
    ins_f = [hg.TSDFRangeDataInserter3D(mode=hg._lib.HG_INSERT_FAST) for _ in RES]
    for k in range(4):                                        # identical maps to start from
        pose = synth.pose_k(k)
        pts = synth.generate_scan(pose, 32, 900, stream=k)
        for grids in (exact, fast):
            hg.insert_pyramid(ins_e, hg.RangeData([0, 0, 0], pts), grids, pose_tq=pose.astype(np.float32))
    pose = synth.pose_k(4)
    pts = synth.generate_scan(pose, 32, 900, stream=4)
    d = torch.from_numpy(pts).to(dev)
    guess = synth.pose_mul(pose, synth.perturbation())
    out = []
    for grids, ins in ((exact, ins_e), (fast, ins_f)):
        p = hg.Problem(ctx)
        i = p.add_pose(guess)
        p.add_block(d, grids, 1.0 / np.sqrt(len(pts)), i, multi_res=True)
        est, summ = hg.register_scan(p, i, ins, hg.RangeData([0, 0, 0], d), grids)
        out.append((est, summ.num_iterations))
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]
    for ge, gf, r in zip(exact, fast, RES):
        tau = float(np.float32(2.5 * r))
        a = sorted_cells(*ge.export())
        b = sorted_cells(*gf.export())
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])   # cells and weight codes
        ta, _ = decode(*a, tau, 1000.0)
        tb, _ = decode(*b, tau, 1000.0)
        assert np.abs(ta - tb).max() <= 5e-3 * tau
        assert (ta != tb).any()                                            # it really took the other path
