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


def test_shard_partitions():
    from hectorgrapher_amd import distributed as hgd
    for world in (1, 2, 4, 8):
        all_items = sorted(sum((hgd.shard(8, r, world) for r in range(world)), []))
        assert all_items == list(range(8))
        assert all(len(hgd.shard(8, r, world)) == 8 // world for r in range(world))
