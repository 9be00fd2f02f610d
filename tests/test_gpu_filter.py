"""GPU parity: device VoxelFilter / AdaptiveVoxelFilter vs the oracle (identical index lists)."""
import numpy as np
import pytest

from hectorgrapher_amd import synth

pytestmark = pytest.mark.gpu


def test_reference_kats(hg, ctx):
    """voxel_filter_test.cc:30-38 (first point of each voxel), :40-48 (coordinates of 1e5 m at 1 cm:
    cell indices of 1e7 need the reference's 3 x 32-bit key), :50-56 (time ignored)."""
    pc = np.array([[0, 0, 0], [0.1, -0.1, 0.1], [0.3, -0.1, 0], [0, 0, 0.1]], np.float32)
    assert hg.VoxelFilter(ctx, 0.3).Filter(pc).tolist() == [0, 2]
    timed = np.array([[-100.0, 0.3, 0.4, float(i)] for i in range(100)], np.float32)
    assert hg.VoxelFilter(ctx, 0.3).Filter(timed).tolist() == [0]
    far = np.array([[100000., 0, 0], [100000.001, -0.0001, 0.0001], [100000.003, -0.0001, 0],
                    [-200000., 0, 0]], np.float32)
    assert hg.VoxelFilter(ctx, 0.01).Filter(far).tolist() == [0, 3]
    assert hg.VoxelFilter(ctx, 0.3).Filter(np.zeros((0, 3), np.float32)).tolist() == []


def test_wide_keys_match_oracle(po, hg, ctx):
    """Clouds that leave the 3 x 21-bit key window (|cell| >= 2^20) go through the 3 x 32-bit pass:
    same index list as the oracle, including voxels that differ only in their high bits."""
    rng = np.random.default_rng(9)
    near = (rng.standard_normal((30000, 3)) * 2).astype(np.float32)
    far = near[:8000] + np.array([[40000.0, -70000.0, 25000.0]], np.float32)
    # aliases of `near` cells modulo 2^21 cells (what a truncated key would merge)
    alias = near[:5000] + np.float32(0.02 * (1 << 21)) * np.array([[1, 0, 0]], np.float32)
    pts = np.concatenate([near, far, alias, near[::-1]])[rng.permutation(30000 * 2 + 13000)]
    for res in (0.02, 0.05):
        a = po.voxel_filter(res, pts)
        b = hg.VoxelFilter(ctx, res).Filter(pts)
        assert np.array_equal(a, b)
        assert len(a) < len(pts)


@pytest.mark.parametrize("res", [0.05, 0.15, 0.5, 2.0])
def test_voxel_filter_matches_oracle(po, hg, ctx, res):
    pts = synth.generate_scan(synth.pose_k(1), 50, 2000, stream=1)
    a = po.voxel_filter(res, pts)
    b = hg.VoxelFilter(ctx, res).Filter(pts)
    assert np.array_equal(a, b)
    rng = np.random.default_rng(5)
    pts4 = np.concatenate([(rng.standard_normal((20000, 3)) * 3).astype(np.float32),
                           rng.random((20000, 1)).astype(np.float32)], axis=1)
    assert np.array_equal(po.voxel_filter(res, pts4), hg.VoxelFilter(ctx, res).Filter(pts4))


@pytest.mark.parametrize("opts", [(2.0, 150, 15.0), (4.0, 200, 60.0), (0.5, 5000, 10.0), (2.0, 1e9, 15.0)])
def test_adaptive_voxel_filter_matches_oracle(po, hg, ctx, opts):
    """trajectory_builder_3d.lua:23-33 defaults (high / low resolution filter) and edge cases."""
    for n_cfg in ((16, 625), (50, 2000)):
        pts = synth.generate_scan(synth.pose_k(2), *n_cfg, stream=2)
        a = po.adaptive_voxel_filter(*opts, pts)
        b = hg.AdaptiveVoxelFilter(ctx, *opts).Filter(pts)
        assert np.array_equal(a, b), (opts, n_cfg, len(a), len(b))


def test_device_input_and_gathered_output(po, hg, ctx):
    torch = pytest.importorskip("torch")
    import ctypes as C
    pts = synth.generate_scan(synth.pose_k(3), 16, 625, stream=3)
    d = torch.from_numpy(pts).to("cuda:0")
    torch.cuda.synchronize()
    keep = hg.AdaptiveVoxelFilter(ctx, 2.0, 150, 15.0).Filter(d)
    assert np.array_equal(keep, po.adaptive_voxel_filter(2.0, 150, 15.0, pts))
    L = hg._lib.load()
    idx, xyz, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
    hg.check(L.hg_filter_last_device(ctx._h, C.byref(idx), C.byref(xyz), C.byref(n)))
    assert n.value == len(keep)
    from hectorgrapher_amd.distributed import _DevArray
    got = torch.as_tensor(_DevArray(xyz.value, (n.value, 3), "<f4"), device="cuda:0").cpu().numpy()
    assert np.array_equal(got, pts[keep])


def test_adaptive_voxel_filter_random_sweep(po, hg, ctx):
    """The device replays the reference's search chain from the pass counts (no host round trips):
    sweep lengths / thresholds so every exit of the chain is taken (pass-through, max_length enough,
    bisection after each halving, no length enough)."""
    rng = np.random.default_rng(11)
    for case in range(40):
        n = int(rng.integers(1, 30000))
        spread = float(rng.choice([0.2, 1.0, 5.0, 20.0]))
        pts = (rng.standard_normal((n, 3)) * spread).astype(np.float32)
        opts = (float(rng.choice([0.3, 1.0, 2.0, 4.0])), float(rng.choice([1, 50, 150, 1000, 20000, 1e7])),
                float(rng.choice([0.5, 5.0, 15.0, 60.0])))
        a = po.adaptive_voxel_filter(*opts, pts)
        b = hg.AdaptiveVoxelFilter(ctx, *opts).Filter(pts)
        assert np.array_equal(a, b), (case, n, spread, opts, len(a), len(b))
