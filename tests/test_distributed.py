"""N > 1 path on CPU: world_size-2 gloo run of the shard + gather-of-TSDF-blocks logic."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from hectorgrapher_amd import distributed as hgd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank r owns submaps r, r+world, ...; fake finished blocks: nb = 3 + 2*rank
    owned = hgd.shard(5, rank, world)
    nb = 3 + 2 * rank
    keys = torch.arange(nb, dtype=torch.int64) + 1000 * rank
    vox = (torch.arange(nb * 512, dtype=torch.int32).reshape(nb, 512) + rank)
    res = hgd.gather_block_arrays(keys, vox, dist, rank, world, dst=0)
    if rank == 0:
        ok = len(res) == world
        for src in range(world):
            k, v = res[src]
            n = 3 + 2 * src
            ok &= k.tolist() == [1000 * src + i for i in range(n)]
            ok &= bool((v == torch.arange(n * 512, dtype=torch.int32).reshape(n, 512) + src).all())
        out.put(("ok" if ok else "bad", owned))
    else:
        assert res is None
        out.put(("peer", owned))
    dist.barrier()
    dist.destroy_process_group()


class _FakeGrid:
    """Host stand-in with the HybridGridTSDF surface verify_gather uses (block dict; export in key
    order): lets the collective protocol run on CPU, the device version is tests/test_gpu_distributed.py."""
    relative_truncation_distance = 2.5
    max_weight = 1000.0

    def __init__(self, ctx=None, resolution=0.1, rtd=2.5, max_weight=1000.0, max_blocks=0):
        self.blocks = {}
        self._res = resolution

    def resolution(self):
        return self._res

    def import_blocks(self, keys, voxels, num_blocks=None):
        v = np.asarray(voxels).reshape(-1, 512)
        for k, row in zip(np.asarray(keys).tolist(), v):
            self.blocks[int(k)] = row.copy()

    def export(self):
        ijk, t, w = [], [], []
        for k in sorted(self.blocks):
            row = self.blocks[k]
            nz = np.nonzero(row)[0]
            ijk += [(k, int(i), 0) for i in nz]
            t += [int(row[i]) & 0xFFFF for i in nz]
            w += [int(row[i]) >> 16 for i in nz]
        return (np.asarray(ijk, np.int32).reshape(-1, 3), np.asarray(t, np.uint16), np.asarray(w, np.uint16))

    def close(self):
        pass


class _FakeApi:
    HybridGridTSDF = _FakeGrid


def _verify_worker(rank, world, port, out, corrupt):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from hectorgrapher_amd import distributed as hgd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(7 + rank)
    grids, gathered = [], []
    for level in range(2):
        g = _FakeGrid(resolution=0.05 * (level + 1))
        nb = 4 + rank + level
        keys = np.arange(nb, dtype=np.uint64) + 100 * rank
        vox = rng.integers(0, 1 << 31, size=(nb, 512), dtype=np.uint32) * (rng.random((nb, 512)) < 0.3)
        g.import_blocks(keys, vox.astype(np.uint32))
        grids.append(g)
        k = torch.from_numpy(keys.view(np.int64).copy())
        v = torch.from_numpy(vox.astype(np.uint32).view(np.int32).copy())
        if corrupt and rank == 1 and level == 1:
            v = v.clone()
            v[0, 5] += 1  # what travels differs from what the peer holds
        gathered.append(hgd.gather_block_arrays(k, v, dist, rank, world, dst=0))
    chk = hgd.verify_gather(_FakeApi, None, grids, gathered, dist, rank, world)
    out.put((rank, chk))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("corrupt", [False, True])
def test_verify_gather_gloo_world2(corrupt):
    """gather -> import into a fresh grid -> export equals the peer's own export (digest), and a
    payload that differs from the peer's grid is caught."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_verify_worker, args=(r, 2, port, q, corrupt)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[1] is None
    assert got[0]["ok"] == (not corrupt)
    assert got[0]["ranks"] == 2 and got[0]["levels"] == 2 and got[0]["blocks"] == 4 + 5 + 5 + 6


def test_gather_blocks_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    tags = sorted(t for t, _ in got)
    assert tags == ["ok", "peer"]
    owned = sorted(sum((o for _, o in got), []))
    assert owned == [0, 1, 2, 3, 4]           # every submap has exactly one owner


class _FakeCtx:
    def synchronize(self):
        pass


class _FakeEngine:
    """Host stand-in of bench.py's offline-batch engine: every owned submap is a pyramid of two fake
    grids; a step adds one block per level whose content depends only on (submap, step, level)."""

    def __init__(self):
        self.ctx = _FakeCtx()

    def open(self, owned):
        self.owned = owned
        self.pyramids = []
        for _ in owned:
            pyr = [_FakeGrid(resolution=0.05 * (l + 1)) for l in range(2)]
            for g in pyr:
                g.ctx = self.ctx
                g.block_tensors = (lambda dev, g=g: _fake_tensors(g))
            self.pyramids.append(pyr)
        self.steps = []

    def step(self, i):
        self.steps.append(i)
        for j, pyr in zip(self.owned, self.pyramids):
            for l, g in enumerate(pyr):
                row = np.full((1, 512), 1 + 1000 * j + 10 * i + l, np.uint32)
                g.import_blocks(np.array([100 * j + i], np.uint64), row)

    def sync(self):
        pass

    def grids(self):
        return [g for pyr in self.pyramids for g in pyr]


def _fake_tensors(g):
    import torch
    keys = np.array(sorted(g.blocks), np.uint64)
    vox = np.stack([g.blocks[int(k)] for k in keys]) if len(keys) else np.zeros((0, 512), np.uint32)
    return torch.from_numpy(keys.view(np.int64).copy()), torch.from_numpy(vox.astype(np.uint32).view(np.int32).copy())


def _sharded_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from hectorgrapher_amd import distributed as hgd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = _FakeEngine()
    res = hgd.map_sharded(8, rank, world, eng, steps=5, warmup=2, barrier=dist.barrier, dist=dist)
    gathered = hgd.gather_grids(eng.grids(), dist, rank, world, None, host=True)
    chk = hgd.verify_gather(_FakeApi, None, eng.grids(), gathered, dist, rank, world)
    out.put((rank, res["owned"], eng.steps, res["scans"], res["elapsed"], chk,
             None if rank else [[int(k.shape[0]) for k, _ in row] for row in gathered]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_offline_batch_driver_gloo_world2():
    """BASELINE configs[3] driver logic on CPU: 8 submaps over 2 ranks (4 each, all of them stepped
    together per step), warm-up + timed steps, ONE gather of all owned grids (8 grids per rank) to rank 0
    and its import / export check."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {g[0]: g for g in (q.get(timeout=120) for _ in procs)}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0][1] == [0, 2, 4, 6] and got[1][1] == [1, 3, 5, 7]
    for r in (0, 1):
        assert got[r][2] == list(range(7))          # 2 warm-up + 5 timed steps, every owned submap each time
        assert got[r][3] == 40 and got[r][4] > 0    # whole-job scans of the timed region, max-over-ranks time
    assert got[1][5] is None
    chk = got[0][5]
    assert chk["ok"] and chk["ranks"] == 2 and chk["levels"] == 8
    assert chk["blocks"] == 2 * 8 * 7               # every (rank, grid) contributed its 7 blocks
    assert got[0][6] == [[7, 7]] * 8


class _UnevenEngine(_FakeEngine):
    """Submaps of different sizes, as finished submaps are: submap j adds (j + l) % 4 blocks to level l per step --
    so some (submap, level) grids stay EMPTY (zero blocks travel, the digest of an empty export still has to
    match) -- and the block counts differ from rank to rank and from grid to grid."""

    def step(self, i):
        self.steps.append(i)
        for j, pyr in zip(self.owned, self.pyramids):
            for l, g in enumerate(pyr):
                for b in range((j + l) % 4):
                    row = np.full((1, 512), 1 + 1000 * j + 10 * i + l + 100000 * b, np.uint32)
                    g.import_blocks(np.array([100 * j + 10 * i + b], np.uint64), row)


def _uneven_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from hectorgrapher_amd import distributed as hgd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = _UnevenEngine()
    res = hgd.map_sharded(8, rank, world, eng, steps=3, warmup=1, barrier=dist.barrier, dist=dist)
    gathered = hgd.gather_grids(eng.grids(), dist, rank, world, None, host=True)
    chk = hgd.verify_gather(_FakeApi, None, eng.grids(), gathered, dist, rank, world)
    out.put((rank, res["owned"], res["scans"], chk,
             None if rank else [[int(k.shape[0]) for k, _ in row] for row in gathered]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_sharded_offline_batch_driver_gloo_world_4_and_8(world):
    """The driver's 4- and 8-rank shapes of BASELINE configs[3] (8 submaps; at 8 ranks ONE submap per rank, the
    shape `bench.py --gpus 8 --total-submaps 8` runs): shard(8, r, world), an all_gather of `world` count rows,
    point-to-point transfers from every peer at once, grids without any block, uneven block counts -- and the
    import / export check of all of it on rank 0."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_uneven_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {g[0]: g for g in (q.get(timeout=300) for _ in procs)}
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    per_rank = 8 // world
    for r in range(world):
        assert got[r][1] == list(range(r, 8, world)) and len(got[r][1]) == per_rank
        assert got[r][2] == 8 * 3                      # whole-job scans of the timed region
        if r:
            assert got[r][3] is None and got[r][4] is None
    chk = got[0][3]
    steps = 4                                          # 1 warm-up + 3 timed
    # what rank 0 must hold: for grid (k-th owned submap, level l) and source rank src the blocks of submap src + k * world
    want = [[((src + k * world + l) % 4) * steps for src in range(world)] for k in range(per_rank) for l in range(2)]
    assert got[0][4] == want
    assert any(0 in row for row in want)               # (the empty grids are really there)
    assert chk["ok"] and chk["ranks"] == world and chk["levels"] == 2 * per_rank
    assert chk["blocks"] == sum(sum(row) for row in want)


def test_map_sharded_hands_step_ranges_to_an_engine_that_runs_them_itself():
    """An engine that maps its submaps in host-thread groups (bench.py's offline-batch Engine) exposes
    run_steps(first, last): map_sharded hands it the warm-up range and the timed range instead of stepping."""
    from hectorgrapher_amd import distributed as hgd

    class Ranged(_FakeEngine):
        def open(self, owned):
            super().open(owned)
            self.ranges = []

        def run_steps(self, first, last):
            self.ranges.append((first, last))
            for i in range(first, last):
                self.step(i)

    eng = Ranged()
    res = hgd.map_sharded(4, 0, 1, eng, steps=3, warmup=2, barrier=lambda: None)
    assert eng.ranges == [(0, 2), (2, 5)] and eng.steps == [0, 1, 2, 3, 4]
    assert res["owned"] == [0, 1, 2, 3] and res["scans"] == 12


def test_map_sharded_refuses_uneven_split():
    from hectorgrapher_amd import distributed as hgd
    with pytest.raises(ValueError):
        hgd.map_sharded(8, 0, 3, _FakeEngine(), 1, 0, lambda: None)


def test_shard_partitions():
    from hectorgrapher_amd import distributed as hgd
    for world in (1, 2, 4, 8):
        all_items = sorted(sum((hgd.shard(8, r, world) for r in range(world)), []))
        assert all_items == list(range(8))
        assert all(len(hgd.shard(8, r, world)) == 8 // world for r in range(world))


def test_bench_gpus_flag_fails_loudly_without_enough_gpus():
    """`python bench.py --gpus 2` without a launcher starts the ranks itself; on a box with fewer
    GPUs it must refuse instead of silently measuring one rank (this container has no GPU)."""
    import subprocess
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has >= 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2 requested but only" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_refuses_world_size_mismatch():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
