"""The one exchange step of batch mapping, end to end on the device: two rank processes map two
different submaps on the GPU, the finished TSDF blocks are gathered to rank 0 (gloo; both ranks
share the one GPU of the test box, which RCCL refuses), rank 0 imports every peer's blocks into a
fresh grid and its export must equal the export the peer made of its own grid
(BASELINE.json configs[3]: "RCCL gather of final TSDF")."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


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
    from hectorgrapher_amd import api, synth
    from hectorgrapher_amd import distributed as hgd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    ctx = api.Context(0)
    res = [0.05, 0.10, 0.20]
    grids = [api.HybridGridTSDF(ctx, r, max_blocks=1 << 14) for r in res]
    ins = [api.TSDFRangeDataInserter3D() for _ in grids]
    # rank r maps its own submap: different poses and noise streams
    for k in range(3):
        pose = synth.pose_k(10 * rank + k)
        pts = synth.generate_scan(pose, 16, 400, stream=100 * rank + k)
        api.insert_pyramid(ins, api.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), grids,
                           pose_tq=pose.astype(np.float32))
    nb = [g.num_blocks() for g in grids]
    gathered = hgd.gather_grids(grids, dist, rank, world, dev, host=True)
    chk = hgd.verify_gather(api, ctx, grids, gathered, dist, rank, world)
    if rank == 0:
        # device import of the gathered arrays as well (the N-GPU path hands device tensors over)
        keys, vox = gathered[0][1]
        fresh = api.HybridGridTSDF(ctx, res[0], max_blocks=max(64, int(keys.shape[0])))
        fresh.import_blocks(keys.to(dev), vox.to(dev), int(keys.shape[0]))
        n_dev, _ = hgd.export_digest(fresh)
        out.put(("rank0", chk, nb, n_dev))
    else:
        assert chk is None
        out.put(("peer", None, nb, hgd.export_digest(grids[0])[0]))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


def test_gather_import_export_two_ranks_one_gpu():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = {g[0]: g for g in got}
    chk = got["rank0"][1]
    assert chk["ok"], chk
    assert chk["ranks"] == 2 and chk["levels"] == 3
    assert chk["blocks"] == sum(got["rank0"][2]) + sum(got["peer"][2])
    assert chk["voxels"] > 10000
    assert got["rank0"][3] == got["peer"][3]  # device-memory import of rank 1's level 0


def _offline_batch(gpus, extra_env, submaps=4, size=("--rings", "16", "--cols", "625", "--map-scans", "3"), steps=3,
                   cpu=False):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--total-submaps", str(submaps),
                        "--steps", str(steps), "--warmup", "1", *size,
                        "--max-blocks", str(1 << 15)] + ([] if cpu else ["--no-cpu-baseline"]),
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_offline_batch_command_two_ranks_equal_one_rank():
    """BASELINE configs[3] as ONE command: `bench.py --gpus G --total-submaps 4`. Two rank processes
    (sharing the one GPU of the test box, so gloo instead of RCCL) map two submaps each through
    hg_register_scan_batch, the finished blocks of all 12 grids are gathered to rank 0 in one exchange and
    pass the import / export check; the same four submaps mapped by ONE rank (four per step) end with the
    same number of blocks and voxels: sharding changes who maps a submap, not the map."""
    two = _offline_batch(2, {"HG_RANKS_SHARE_GPU": "1"})
    one = _offline_batch(1, {"HG_FORCE_DIST": "1", "HG_DIST_BACKEND": "gloo", "MASTER_PORT": str(_free_port())})
    for out, g in ((two, 2), (one, 1)):
        assert out["n_gpus"] == g and out["scaling"] == "strong" and out["value"] > 0
        cfg = out["config"]
        assert cfg["total_submaps"] == 4 and cfg["submaps_per_gpu"] == 4 // g
        chk = cfg["gather_check"]
        assert chk and chk["ok"] and chk["ranks"] == g and chk["levels"] == 3 * (4 // g)
        assert cfg["gather_ms"] > 0
    assert two["config"]["gather_check"]["blocks"] == one["config"]["gather_check"]["blocks"]
    assert two["config"]["gather_check"]["voxels"] == one["config"]["gather_check"]["voxels"]
    assert abs(two["config"]["mean_pose_error_m"] - one["config"]["mean_pose_error_m"]) < 1e-9


def test_offline_batch_command_full_size_two_ranks():
    """The sharded command at BASELINE's scan size (100 000 returns, 50 x 2000): two rank processes on the
    one GPU, one submap each (the single registration chain per rank -- what an 8-GPU run of 8 submaps
    does), two timed steps, gather + import / export check; the same two submaps mapped by one rank
    (batched registration, with the oracle replaying its submap 0 inside the run) end with the same
    blocks and voxels."""
    size = ("--rings", "50", "--cols", "2000", "--map-scans", "3")
    two = _offline_batch(2, {"HG_RANKS_SHARE_GPU": "1"}, submaps=2, size=size, steps=2)
    one = _offline_batch(1, {"HG_FORCE_DIST": "1", "HG_DIST_BACKEND": "gloo", "MASTER_PORT": str(_free_port())},
                         submaps=2, size=size, steps=2)
    for out, g in ((two, 2), (one, 1)):
        assert out["n_gpus"] == g and out["value"] > 0
        chk = out["config"]["gather_check"]
        assert chk and chk["ok"] and chk["ranks"] == g and chk["levels"] == 3 * (2 // g)
    assert two["config"]["gather_check"]["blocks"] == one["config"]["gather_check"]["blocks"]
    assert two["config"]["gather_check"]["voxels"] == one["config"]["gather_check"]["voxels"]
    assert abs(two["config"]["mean_pose_error_m"] - one["config"]["mean_pose_error_m"]) < 1e-9
    # one rank, no process group: the in-run oracle replay of submap 0 at full size
    solo = _offline_batch(1, {}, submaps=2, size=size, steps=2, cpu=True)
    par = solo["parity"]
    assert par and par["max_dt_m"] <= 1e-4 and par["max_dr_rad"] <= 1e-4 and par["same_iterations_and_termination"]


def test_gpus_flag_line_carries_the_strong_scaling_leg():
    """`bench.py --gpus N` (N > 1) is the command the driver's scaling step runs: besides the weak-scaling headline
    (one independent submap per rank) the line must carry BASELINE configs[3] -- 8 submaps farmed to the N ranks in
    the SAME process group, the gather of their blocks and its check -- and say how many ranks the process group
    had. Two ranks on the test box's one GPU (gloo)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HG_RANKS_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--rings", "16", "--cols", "625", "--map-scans", "3", "--max-blocks", str(1 << 15)],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["process_group"]["nranks"] == 2
    assert out["config"]["gather_check"]["ok"] and out["config"]["gather_check"]["ranks"] == 2
    leg = out["secondary"]["offline8"]
    assert "error" not in leg, leg
    assert leg["n_gpus"] == 2 and leg["scaling"] == "strong" and leg["value"] > 0
    assert leg["total_submaps"] == 8 and leg["submaps_per_gpu"] == 4
    assert leg["process_group"]["nranks"] == 2
    assert leg["gather_ms"] > 0
    chk = leg["gather_check"]
    assert chk["ok"] and chk["ranks"] == 2 and chk["levels"] == 3 * 4
