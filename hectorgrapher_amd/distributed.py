"""Multi-GPU batch mapping: independent submaps per rank + one gather of finished TSDF blocks.

The path shards only across independent units: inside one submap scans are
sequentially dependent, so submap s runs on rank s mod world with no communication during
mapping. The single exchange step is a gather of the occupied 8^3 blocks (8-byte key + 2 KiB
voxels each) to rank 0: counts first (all_gather), then point-to-point send/recv per peer —
variable sizes, no reduction (submaps are disjoint). Works on any torch.distributed backend
(nccl = RCCL over xGMI on the GPU box, gloo in the CPU tests).
"""
import torch


def shard(num_items, rank, world):
    """Indices of the independent units (submaps) owned by `rank`."""
    return list(range(rank, num_items, world))


class _DevArray:
    """Zero-copy view of library-owned device memory for torch.as_tensor."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


def grid_block_tensors(grid, dev):
    """(keys int64[nb], voxels int32[nb, 512]) tensors aliasing the grid's block pool."""
    kptr, vptr, nb = grid.block_arrays()
    if nb == 0:
        return (torch.empty(0, dtype=torch.int64, device=dev),
                torch.empty((0, 512), dtype=torch.int32, device=dev))
    keys = torch.as_tensor(_DevArray(kptr, (nb,), "<i8"), device=dev)
    vox = torch.as_tensor(_DevArray(vptr, (nb, 512), "<i4"), device=dev)
    return keys, vox


def gather_block_arrays(keys, vox, dist, rank, world, dst=0):
    """Gathers variable-length (keys[nb], vox[nb,512]) from every rank to `dst`.

    Returns on dst a list over source ranks of (keys, vox); elsewhere None.
    """
    dev = keys.device
    n = torch.tensor([keys.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    if rank == dst:
        out = []
        reqs = []
        for src in range(world):
            if src == dst:
                out.append((keys.clone(), vox.clone()))
                continue
            k = torch.empty(counts[src], dtype=torch.int64, device=dev)
            v = torch.empty((counts[src], 512), dtype=torch.int32, device=dev)
            out.append((k, v))
            if counts[src] > 0:
                reqs.append(dist.irecv(k, src=src))
                reqs.append(dist.irecv(v, src=src))
        for r in reqs:
            r.wait()
        return out
    if keys.shape[0] > 0:
        r1 = dist.isend(keys.contiguous(), dst=dst)
        r2 = dist.isend(vox.contiguous(), dst=dst)
        r1.wait()
        r2.wait()
    return None


def gather_grids(grids, dist, rank, world, dev, dst=0, host=False):
    """Gathers the blocks of every grid of `grids` (the levels of one pyramid, or of all the submaps a
    rank owns: every rank must pass the same number of grids) to `dst`: ONE all_gather of the block
    counts of all grids, then point-to-point transfers of the non-empty ones, all in flight together.
    Returns [grid][src_rank] -> (keys, vox) on dst, a list of None elsewhere.
    host=True stages the block arrays through host memory (gloo, or ranks that share one GPU)."""
    ctxs = []
    for g in grids:
        if g.ctx not in ctxs:
            ctxs.append(g.ctx)
    for c in ctxs:
        c.synchronize()
    arrays = []
    for g in grids:
        # (a grid type may hand its block arrays over itself: host stand-ins of the CPU tests)
        keys, vox = g.block_tensors(dev) if hasattr(g, "block_tensors") else grid_block_tensors(g, dev)
        if host:
            keys, vox = keys.cpu(), vox.cpu()
        arrays.append((keys, vox))
    if not arrays:
        return []
    tdev = arrays[0][0].device
    mine = torch.tensor([a[0].shape[0] for a in arrays], dtype=torch.int64, device=tdev)
    allc = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allc, mine)
    counts = [c.tolist() for c in allc]  # [src][grid]
    reqs = []
    if rank == dst:
        out = []
        for gi, (keys, vox) in enumerate(arrays):
            row = []
            for src in range(world):
                if src == dst:
                    row.append((keys.clone(), vox.clone()))
                    continue
                k = torch.empty(counts[src][gi], dtype=torch.int64, device=tdev)
                v = torch.empty((counts[src][gi], 512), dtype=torch.int32, device=tdev)
                row.append((k, v))
                if counts[src][gi] > 0:
                    reqs.append(dist.irecv(k, src=src))
                    reqs.append(dist.irecv(v, src=src))
            out.append(row)
        for r in reqs:
            r.wait()
        return out
    for keys, vox in arrays:
        if keys.shape[0] > 0:
            reqs.append(dist.isend(keys.contiguous(), dst=dst))
            reqs.append(dist.isend(vox.contiguous(), dst=dst))
    for r in reqs:
        r.wait()
    return [None] * len(arrays)


def map_sharded(total_submaps, rank, world, engine, steps, warmup, barrier, dist=None):
    """BASELINE configs[3]: `total_submaps` independent submaps farmed to `world` ranks, rank r owning
    shard(total_submaps, r, world) -- no data-path collective while mapping. `engine` maps the owned
    submaps: engine.open(owned) builds them, engine.step(i) registers scan i of EVERY owned submap (the
    batched registration step; or engine.run_steps(first, last) for a range of steps), engine.sync() drains the
    device. Times `steps` steps behind `warmup`
    untimed ones, bracketed by `barrier()`; returns {"owned", "elapsed": max over ranks, "scans": all
    ranks' scans in the timed region}."""
    import time
    if total_submaps % world != 0:
        raise ValueError("map_sharded: %d submaps do not divide over %d ranks" % (total_submaps, world))
    owned = shard(total_submaps, rank, world)
    engine.open(owned)
    # an engine that maps its submaps in several host threads (one context each, so that one group's insertions
    # overlap another group's matches) runs a range of steps itself: engine.run_steps(first, last)
    run = getattr(engine, "run_steps", None)
    if run is None:
        def run(first, last):
            for i in range(first, last):
                engine.step(i)
    run(0, warmup)
    engine.sync()
    barrier()
    t0 = time.perf_counter()
    run(warmup, warmup + steps)
    engine.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.to(torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return {"owned": owned, "elapsed": elapsed, "scans": total_submaps * steps}


def export_digest(grid):
    """(voxel count, sha1 of the grid's export in iterator order): what two grids must share to be
    the same HybridGridTSDF (cells, order, tsd and weight codes)."""
    import hashlib
    import numpy as np
    ijk, t, w = grid.export()
    h = hashlib.sha1()
    for a in (ijk, t, w):
        h.update(np.ascontiguousarray(a).tobytes())
    return len(t), h.digest()


def verify_gather(api, ctx, grids, gathered, dist, rank, world, dst=0):
    """End-to-end check of the one exchange step: on `dst` every peer's gathered blocks are imported
    into a fresh grid (hg_grid_import_blocks) whose export must equal the export the peer computed
    from its own grid (digests travel by all_gather). Returns on dst
    {"ok", "levels", "ranks", "blocks", "voxels"}; elsewhere None. Collective: every rank calls it."""
    import numpy as np
    dev = gathered[0][0][0].device if (rank == dst and gathered and gathered[0]) else None
    mine = []
    for g in grids:
        n, d = export_digest(g)
        mine.append(np.frombuffer(np.uint64(n).tobytes() + d, np.uint8))
    mine = np.concatenate(mine) if mine else np.zeros(0, np.uint8)
    t_dev = dev if dev is not None else (torch.device("cuda", torch.cuda.current_device())
                                         if dist.get_backend() == "nccl" else torch.device("cpu"))
    local = torch.from_numpy(mine.copy()).to(t_dev)
    allv = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(allv, local)
    if rank != dst:
        return None
    ok, blocks, voxels = True, 0, 0
    for l, g in enumerate(grids):
        for src in range(world):
            keys, vox = gathered[l][src]
            nb = int(keys.shape[0])
            fresh = api.HybridGridTSDF(ctx, float(g.resolution()), g.relative_truncation_distance,
                                       float(g.max_weight), max_blocks=max(64, nb))
            if nb:
                if keys.is_cuda:
                    fresh.import_blocks(keys, vox, nb)
                else:
                    fresh.import_blocks(keys.numpy().view(np.uint64), vox.numpy().view(np.uint32).reshape(-1))
            n, d = export_digest(fresh)
            fresh.close()
            want = allv[src].cpu().numpy()[28 * l:28 * (l + 1)]
            want_n = int(np.frombuffer(want[:8].tobytes(), np.uint64)[0])
            ok = ok and n == want_n and d == want[8:].tobytes()
            blocks += nb
            voxels += n
    return {"ok": bool(ok), "levels": len(grids), "ranks": world, "blocks": blocks, "voxels": voxels}
