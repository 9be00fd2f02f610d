#!/usr/bin/env python3
"""bench.py — scan registration throughput on MI355X (BASELINE.json metric).

One step = register one 100k-point synthetic scan into a 3-resolution TSDF
(0.05 / 0.10 / 0.20 m): multi-resolution TSDF scan matching (on-device LM) from a
perturbed initial guess, then exact TSDF insertion of the scan at the matched
pose into all three grids. Inputs (scans) are resident in HBM before the timed
region. N > 1: one process per GPU, one independent submap per rank (weak
scaling, no data-path collective); the finished TSDF blocks are gathered to rank
0 over RCCL after the timed region and reported as `gather_ms`.

    python bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RESOLUTIONS = [0.05, 0.10, 0.20]
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: 120 for the headline -- the whole bench trajectory, 28 ms of timed region: the "
                         "20 steps of rounds 1-5 were a 5 ms region, shorter than the device takes to settle its clocks, and "
                         "read 4 %% low -- and 20 for the other workloads)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rings", type=int, default=50)
    ap.add_argument("--cols", type=int, default=2000)
    ap.add_argument("--map-scans", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-scans", type=int, default=3)
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="host threads of the all-cores CPU baseline of the batch workloads (default: every core the box reports)")
    ap.add_argument("--max-blocks", type=int, default=1 << 18)
    ap.add_argument("--window", type=int, default=10, help="control points of --workload window")
    ap.add_argument("--no-window-unwarp", action="store_true",
                    help="--workload window: insert the leaving scan at control point 1's pose instead of unwarping it "
                         "per point between control points 0 and 1 (hg_register_scan_unwarped)")
    ap.add_argument("--workload", default="register", choices=["register", "insert_stream", "window", "window_batch", "register_filtered", "match_batch", "register_batch", "c1_10k"],
                    help="register = BASELINE configs[1] (default, the headline metric); insert_stream = "
                         "configs[2]: B scans with known poses inserted per step in one batched call")
    ap.add_argument("--stream-scans", type=int, default=64)
    ap.add_argument("--stream-tiles", type=int, default=1,
                    help="--workload insert_stream: scan i is shifted to copy (i mod T) of the room (copies 30 m apart "
                         "on a centred square lattice), so that the blocks the stream touches outgrow the 256 MB Infinity Cache "
                         "(SURVEY 8d: B = 500 with an HBM-resident working set); 1 = the single room")
    ap.add_argument("--batch", type=int, default=8, help="--workload match_batch: independent matches per call")
    ap.add_argument("--batch-maps", type=int, default=1,
                    help="--workload match_batch: identical copies of the map; consecutive batches of a handle set are matched "
                         "against different copies (a constraint search changes submaps from batch to batch: every problem's "
                         "pyramid description is uploaded again)")
    ap.add_argument("--batches-in-flight", type=int, default=2,
                    help="--workload match_batch: sets of problem handles the search keeps; 2 = build batch k + 1 while "
                         "batch k runs (hg_problem_solve_batch_async), 1 = one blocking call per batch")
    ap.add_argument("--insert-mode", default="exact", choices=["exact", "fast"],
                    help="exact = voxel codes bit-identical to the reference (default, headline); fast = "
                         "HG_INSERT_FAST tolerance mode (order-free sums, one quantisation per call) for "
                         "--workload insert_stream and, as an extra, for the registration step")
    ap.add_argument("--submaps", type=int, default=1,
                    help="--workload register: independent submaps mapped concurrently on ONE GPU, one process "
                         "each (BASELINE configs[3] at G = 1 puts all submaps on one GPU); 1 = the headline case")
    ap.add_argument("--batch-submaps", type=int, default=8,
                    help="--workload register_batch: independent submaps registered together in ONE process "
                         "(hg_register_scan_batch: shared launches), BASELINE configs[3] at G = 1")
    ap.add_argument("--host-threads", type=int, default=0,
                    help="offline batch (--total-submaps): host threads per rank, each mapping its share of the rank's "
                         "submaps on a context of its own (0 = 2 from eight owned submaps on, else 1)")
    ap.add_argument("--batch-threads", type=int, default=1,
                    help="--workload register_batch: host threads, each with its own context (stream) and an equal share of the submaps")
    ap.add_argument("--no-persistent-solve", action="store_true",
                    help="headline: a launch per evaluation instead of the persistent single-launch solve")
    ap.add_argument("--no-secondary", action="store_true",
                    help="default workload: skip the bounded runs of the secondary workloads (match_batch 64, "
                         "register_batch 8, insert_stream 32, window) that fill the `secondary` object")
    ap.add_argument("--host-steps", type=int, default=20,
                    help="default workload: extra registration steps with the scans handed over in HOST memory "
                         "(what a drop-in receives), reported as `host_inclusive`; 0 disables")
    ap.add_argument("--total-submaps", type=int, default=0,
                    help="BASELINE configs[3] (offline batch mapping): this many independent submaps farmed to the "
                         "--gpus ranks (rank r owns submaps r, r + G, ...; must divide), every rank registering one "
                         "scan of each of its submaps per step through hg_register_scan_batch, then ONE gather of "
                         "the finished TSDF blocks to rank 0 (RCCL), checked by import + export digests. Strong "
                         "scaling: the work is fixed as G grows")
    ap.add_argument("--scans-per-submap", type=int, default=0,
                    help="--total-submaps: timed scans per submap (overrides --steps; configs[3] names 500)")
    ap.add_argument("--submap-index", type=int, default=-1, help=argparse.SUPPRESS)  # child of --submaps
    ap.add_argument("--prof-every", type=int, default=4,
                    help="HIP-event kernel timing on every N-th timed step (each event pair costs "
                         "~8 us of stream serialisation; 0 disables)")
    return ap.parse_args()


def make_scans(rings, cols, first, count, stream_base):
    from hectorgrapher_amd import synth
    out = []
    for k in range(first, first + count):
        # SURVEY §8d poses P_k; long runs fold k back and forth over [0, 60] so the sensor stays
        # inside the room (P_k leaves it beyond k ~ 90)
        kk = k % 120
        kk = kk if kk <= 60 else 120 - kk
        pose = synth.pose_k(kk)
        pts = synth.generate_scan(pose, rings, cols, stream=stream_base + k)
        out.append((pose, pts))
    return out


def cpu_threads(args):
    """Host threads of the all-cores CPU legs: EVERY core the box reports (round 5 stopped at 64 of the
    driver box's 256), --cpu-threads to bound it; each thread builds a map of its own."""
    cores = os.cpu_count() or 1
    want = getattr(args, "cpu_threads", 0)
    return max(1, min(cores, want if want > 0 else cores))


def all_cores_baseline(args, prepare, work, units_per_thread, what):
    """the batch configurations' CPU figure with every core busy -- one INDEPENDENT unit (its own
    submap) per host thread, all threads at once (ctypes drops the GIL inside the oracle). prepare() builds a thread's
    state untimed, work(state) is what is timed, from the moment all threads are ready until the last one is done."""
    import threading
    T = cpu_threads(args)
    ready = threading.Barrier(T + 1)
    errors, ends = [], []

    def run():
        try:
            state = prepare()
            ready.wait()
            work(state)
            ends.append(time.perf_counter())
        except BaseException as e:  # surfaced below
            errors.append(e)
            try:
                ready.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=run) for _ in range(T)]
    for t in threads:
        t.start()
    try:
        ready.wait()
    except threading.BrokenBarrierError:
        pass
    t0 = time.perf_counter()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    elapsed = max(ends) - t0
    return {"value": T * units_per_thread / elapsed, "cores": T, "cores_available": os.cpu_count(),
            "sample": "%s; one per host thread, %d threads at once, oracle -O3" % (what, T)}


def oracle_map(po, synth, map_scans):
    og = [po.Grid(r) for r in RESOLUTIONS]
    for pose, pts in map_scans:
        loc = synth.transform_points(pose, pts)
        for g in og:
            g.insert(pose[:3], loc)
    return og


def cpu_baseline(args, map_scans, query_scans, gpu_steps):
    """Oracle (CPU restatement of the reference, 1 thread) on a bounded sample of the workload: the
    first `--cpu-scans` TIMED registration steps of this very run (same map: the map scans plus the
    warm-up scans inserted where the GPU inserted them; same guesses; scans inserted at the solved
    poses), so its poses are also the parity gate of the timed GPU steps:
    gpu_steps[i] = (pose, num_iterations, termination_type, termination_reason) of GPU step i
    (warm-up steps first)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    from hectorgrapher_amd import synth
    grids = [po.Grid(r) for r in RESOLUTIONS]
    for pose, pts in map_scans:
        loc = synth.transform_points(pose, pts)
        for g in grids:
            g.insert(pose[:3], loc)
    for k in range(args.warmup):  # untimed: bring the oracle's map to the state the timed steps start from
        at = gpu_steps[k][0]
        loc = synth.transform_points(at, query_scans[k][1])
        for g in grids:
            g.insert(at[:3].astype(np.float32), loc)
    t_match = t_insert = 0.0
    n = 0
    probed = lookups = 0
    u_total = nin_total = 0
    max_dt = max_dr = 0.0
    same_flow = True
    for k, (pose, pts) in enumerate(query_scans[args.warmup:args.warmup + args.cpu_scans], start=args.warmup):
        guess = synth.pose_mul(pose, synth.perturbation())
        t0 = time.perf_counter()
        pr = po.Problem()
        i = pr.add_pose(guess)
        pr.add_block(pts, grids, 1.0 / np.sqrt(len(pts)), i, multi_res=True)
        so = pr.solve()
        est = pr.get_pose(i)
        t1 = time.perf_counter()
        g_pose, g_it, g_tt, g_tr = gpu_steps[k]
        max_dt = max(max_dt, float(np.linalg.norm(est[:3] - g_pose[:3])))
        max_dr = max(max_dr, float(2.0 * np.arccos(min(1.0, abs(float(np.dot(est[3:], g_pose[3:])))))))
        same_flow = same_flow and (so.num_iterations, so.termination_type, so.termination_reason) == (g_it, g_tt, g_tr)
        # the map the next step is matched against must be the GPU's: insert where the GPU inserted if
        # the two float casts of the pose differ (they agree unless a component straddles a float tie)
        at = est if np.array_equal(est.astype(np.float32), g_pose.astype(np.float32)) else g_pose
        t1b = time.perf_counter()
        loc = synth.transform_points(at, pts)
        for g in grids:
            nin, u = g.insert(at[:3].astype(np.float32), loc)
            u_total += u
            nin_total += nin
        t2 = time.perf_counter()
        lk, pb = pr.lookup_stats()
        lookups += lk
        probed += pb
        t_match += t1 - t0
        t_insert += t2 - t1b
        n += 1
    total = t_match + t_insert
    return {
        "value": n / total, "unit": "scans/s", "cores": 1, "kind": "port",
        "sample": "the first %d timed registration steps of this run (match + 3-level insert), oracle -O3 1 thread; "
                  "match %.3f s/scan, insert %.3f s/scan" % (n, t_match / n, t_insert / n),
        "mean_levels_probed": probed / max(1, lookups),
        "updates_per_scan": u_total / n, "hits_per_scan": nin_total / n,
        "parity": {"max_dt_m": max_dt, "max_dr_rad": max_dr, "scans": n, "tolerance": 1e-4,
                   "same_iterations_and_termination": bool(same_flow),
                   "steps": "timed steps %d..%d" % (0, n - 1)},
    }


def main():
    args = parse_args()
    if args.steps is None:
        args.steps = 120 if (args.workload == "register" and args.total_submaps == 0) else 20
    # the timed loops are a few hundred microseconds per step: a generational collection of the
    # interpreter (tens of milliseconds with torch loaded) in the middle of one would be measured as
    # the library's time
    import gc
    gc.collect()
    gc.disable()
    # libraries (RCCL banner, ...) may write to fd 1: keep stdout for the single JSON line
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
            # no external launcher: this process starts the N ranks itself and never touches a GPU
            result = run_rank_processes(args)
        elif "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
            raise SystemExit("bench.py: --gpus %d but the launcher set WORLD_SIZE=%s" % (args.gpus, os.environ["WORLD_SIZE"]))
        elif args.submaps > 1 and args.workload == "register" and int(os.environ.get("WORLD_SIZE", "1")) == 1:
            result = run_submap_processes(args, saved_stdout)
        else:
            result = run(args, saved_stdout)
    finally:
        sys.stdout.flush()
        try:  # banners buffered in C stdio (RCCL prints through printf) must not land after the JSON line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if result is not None:
        print(json.dumps(result), flush=True)


def run_rank_processes(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes (one per GPU, the
    environment torch.distributed.run would give them) BEFORE anything initialises a GPU in this
    process, wait for them and hand rank 0's JSON line through. Fails loudly when the box has fewer
    than N GPUs -- a silent 1-rank run would be reported as an N-GPU number."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()  # counts devices without creating a HIP context
    share = os.environ.get("HG_RANKS_SHARE_GPU") == "1"  # tests on a one-GPU box: every rank on GPU 0, gloo
    if have < args.gpus and not (share and have >= 1):
        raise SystemExit("bench.py: --gpus %d requested but only %d GPU(s) are visible" % (args.gpus, have))
    if args.workload != "register" and args.total_submaps <= 0:
        raise SystemExit("bench.py: --gpus N > 1 runs the register workload (independent submap per rank) "
                         "or --total-submaps (offline batch mapping)")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(0 if share else r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if share:
            env["HG_DIST_BACKEND"] = "gloo"  # RCCL refuses two ranks on one device
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    if any(rcs):
        raise SystemExit("bench.py: rank processes failed, exit codes %r" % (rcs,))
    lines = [l for l in out0.splitlines() if l.startswith("{")]
    if not lines:
        raise SystemExit("bench.py: rank 0 printed no result line")
    result = json.loads(lines[-1])
    if result.get("n_gpus") != args.gpus:
        raise SystemExit("bench.py: rank 0 reported n_gpus=%r, expected %d" % (result.get("n_gpus"), args.gpus))
    return result


def run_submap_processes(args, out_fd):
    """S independent submaps on one GPU, one child process each (own HIP context and stream). The
    children warm up, report READY, start their timed steps together on GO and return their own JSON
    line with wall-clock start / end stamps; the whole-job rate is all their scans over the span from
    the first start to the last end. This process never touches the GPU."""
    import subprocess
    S = args.submaps
    base_cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps),
                "--warmup", str(args.warmup), "--rings", str(args.rings), "--cols", str(args.cols),
                "--map-scans", str(args.map_scans), "--max-blocks", str(args.max_blocks),
                "--cpu-scans", str(args.cpu_scans), "--prof-every", str(args.prof_every),
                "--insert-mode", args.insert_mode]
    children = []
    for j in range(S):
        cmd = base_cmd + ["--submap-index", str(j)]
        if j > 0 or args.no_cpu_baseline:
            cmd.append("--no-cpu-baseline")
        children.append(subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True))
    for c in children:
        line = c.stdout.readline()
        if line.strip() != "READY":
            for k in children:
                k.kill()
            raise RuntimeError("submap child failed before its timed region: %r" % line)
    for c in children:
        c.stdin.write("GO\n")
        c.stdin.flush()
    results = []
    for c in children:
        out, _ = c.communicate()
        lines = [l for l in out.splitlines() if l.startswith("{")]
        if c.returncode != 0 or not lines:
            raise RuntimeError("submap child failed (rc %s)" % c.returncode)
        results.append(json.loads(lines[-1]))
    t_start = min(r["t_start"] for r in results)
    t_end = max(r["t_end"] for r in results)
    elapsed = t_end - t_start
    out = dict(results[0])
    for k in ("t_start", "t_end"):
        out.pop(k, None)
    out["value"] = args.steps * S / elapsed
    out["ms_per_step"] = elapsed / args.steps * 1e3
    out["config"] = dict(out["config"])
    out["config"]["workload"] += "; a step registers one scan into each of %d concurrent submaps (one process each)" % S
    out["config"]["parallelism"] = "%d independent submaps on one GPU, one process each" % S
    out["config"]["submaps_per_gpu"] = S
    out["config"]["per_submap_scans_per_s"] = [r["value"] for r in results]
    out["config"]["mean_pose_error_m"] = float(np.mean([r["config"]["mean_pose_error_m"] for r in results]))
    if "cpu_baseline" in out:
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def run_insert_stream(args):
    """BASELINE configs[2]: TSDFRangeDataInserter3D on a 100k-pt scan stream, 3 hashed-block grids.
    One step = args.stream_scans scans (known poses) inserted by ONE hg_pyramid_insert_batch call."""
    import torch
    from hectorgrapher_amd import api, synth
    dev = torch.device("cuda", 0)
    ctx = api.Context(0)
    n_pts = args.rings * args.cols
    B = args.stream_scans
    scans = make_scans(args.rings, args.cols, 0, B, 0)
    if args.stream_tiles > 1:
        shifted = []
        lattice = int(np.ceil(np.sqrt(args.stream_tiles)))  # centred square lattice: +-8192 cells of 0.05 m reach +-409 m
        for i, (pose, pts) in enumerate(scans):
            t = i % args.stream_tiles
            pose = pose.copy()
            pose[0] += 30.0 * (t % lattice - lattice // 2)
            pose[1] += 30.0 * (t // lattice - lattice // 2)
            shifted.append((pose, pts))
        scans = shifted
    xyz = torch.from_numpy(np.concatenate([p for _, p in scans])).to(dev)
    poses = np.array([pose for pose, _ in scans], np.float32)
    origins = np.zeros((B, 3), np.float32)
    offs = np.arange(B + 1, dtype=np.uint64) * n_pts
    torch.cuda.synchronize()
    grids = [api.HybridGridTSDF(ctx, r, max_blocks=args.max_blocks) for r in RESOLUTIONS]
    ins = api.TSDFRangeDataInserter3D()
    L = api._lib.load()
    import ctypes as C
    garr = (C.c_void_p * 3)(*[g._h for g in grids])
    opts = (api.InsertOpts * 3)(*[api.InsertOpts() for _ in grids])
    st = (api.InsertStats * 3)()
    mode = api._lib.HG_INSERT_FAST if args.insert_mode == "fast" else api._lib.HG_INSERT_EXACT

    def step(stats):
        api.check(L.hg_pyramid_insert_batch(garr, opts, 3, origins.ctypes.data_as(C.c_void_p), xyz.data_ptr(),
                                            offs.ctypes.data_as(C.c_void_p), B, 0,
                                            poses.ctypes.data_as(C.c_void_p), mode, 1, st if stats else None),
                  "hg_pyramid_insert_batch")

    for _ in range(args.warmup):
        step(False)
    ctx.prof_reset()
    ctx.synchronize()
    t0 = time.perf_counter()
    sampled = 0
    for i in range(args.steps):
        # kernel durations are sampled with HIP events on every prof_every-th step (an event pair
        # serialises the stream for ~8 us, 128 of them per step)
        on = args.prof_every > 0 and i % args.prof_every == 0
        ctx.prof_enable(on)
        sampled += 1 if on else 0
        step(False)
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    prof = ctx.prof_read()
    ctx.prof_enable(False)
    step(True)
    U = sum(s_.num_updates for s_ in st)
    N_in = sum(s_.num_hits for s_ in st)
    base = None
    if not args.no_cpu_baseline:
        # (i) 1 thread = the reference's single-threaded inserter; (ii) all cores, one
        # independent submap per thread (ctypes drops the GIL inside the oracle calls)
        import threading
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle as po
        sample = scans[:args.cpu_scans]

        def cpu_insert():
            og = [po.Grid(r) for r in RESOLUTIONS]
            for pose, pts in sample:
                loc = synth.transform_points(pose, pts)
                for g in og:
                    g.insert(pose[:3], loc)

        t1 = time.perf_counter()
        cpu_insert()
        one = len(sample) / (time.perf_counter() - t1)
        cores = os.cpu_count() or 1
        threads = [threading.Thread(target=cpu_insert) for _ in range(cores)]
        t1 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        allc = cores * len(sample) / (time.perf_counter() - t1)
        base = {"value": one, "unit": "scans/s", "cores": 1, "kind": "port",
                "sample": "%d scans x 3 levels into fresh grids, oracle -O3 1 thread" % len(sample),
                "all_cores": {"value": allc, "cores": cores,
                              "sample": "one independent submap per thread, %d scans each" % len(sample)}}
    parity = None
    if not args.no_cpu_baseline:
        # gate: the first --cpu-scans scans of the stream, ONE batched call into fresh grids, against the
        # oracle inserting them one after the other: every cell, code and the order of the export
        n_chk = min(B, args.cpu_scans)
        fresh = [api.HybridGridTSDF(ctx, r, max_blocks=args.max_blocks) for r in RESOLUTIONS]
        farr = (C.c_void_p * 3)(*[g._h for g in fresh])
        offs_chk = np.arange(n_chk + 1, dtype=np.uint64) * n_pts
        api.check(L.hg_pyramid_insert_batch(farr, opts, 3, origins.ctypes.data_as(C.c_void_p), xyz.data_ptr(),
                                            offs_chk.ctypes.data_as(C.c_void_p), n_chk, 0,
                                            poses.ctypes.data_as(C.c_void_p), mode, 1, st), "hg_pyramid_insert_batch")
        og = [po.Grid(r) for r in RESOLUTIONS]
        for pose, pts in scans[:n_chk]:
            loc = synth.transform_points(pose, pts)
            for g in og:
                g.insert(pose[:3], loc)
        same = all(np.array_equal(x, y) for o, g in zip(og, fresh) for x, y in zip(o.export(), g.export()))
        voxels = int(sum(len(g.export()[1]) for g in fresh))
        parity = {"bit_exact": bool(same), "scans": n_chk, "voxels": voxels,
                  "check": "cells, order, tsd and weight codes of the three grids after one batched call"}
        if args.insert_mode == "fast":
            # tolerance mode (tests/test_gpu_insert_fast.py states the bound): the same cells in the same order,
            # identical weight codes, |d tsd| <= 1e-2 tau + half a tsd code per update of the voxel
            cells = codes_w = True
            worst = 0.0
            within = True
            for o, g, r in zip(og, fresh, RESOLUTIONS):
                (ia, ta, wa), (ib, tb, wb) = o.export(), g.export()
                cells = cells and np.array_equal(ia, ib)
                if not np.array_equal(ia, ib):
                    continue
                codes_w = codes_w and np.array_equal(wa, wb)
                tau = float(np.float32(2.5 * r))
                ks = 2 * tau / 32766.0
                m = np.maximum(1.0, np.round((wb & 0x7FFF).astype(np.float64) * (1000.0 / 32766.0)))
                dt = np.abs((ta & 0x7FFF).astype(np.float64) - (tb & 0x7FFF).astype(np.float64)) * ks
                worst = max(worst, float((dt / tau).max()))
                within = within and bool(np.all(dt <= 1e-2 * tau + 0.5 * ks * m))
            ok = bool(cells and codes_w and within)
            parity = {"tolerance_ok": ok, "same_cells_and_order": bool(cells), "weight_codes_identical": bool(codes_w),
                      "max_dtsd_over_tau": worst, "scans": n_chk, "voxels": voxels,
                      "check": "tolerance mode: cells + order + weight codes identical, |d tsd| <= 1e-2 tau + half a code per update"}
            if not ok:
                raise SystemExit("bench.py: tolerance gate of the fast insertion failed: %r" % (parity,))
        for g in fresh:
            g.close()
        if args.insert_mode == "exact" and not same:
            raise SystemExit("bench.py: parity gate failed, the streamed insertion differs from the oracle: %r" % (parity,))
    blocks = [g.num_blocks() for g in grids]
    t_fam = sum(prof[k][1] for k in ("ray_count", "scan", "ray_expand", "sort", "alloc", "apply"))
    avg_ms = t_fam / max(1, prof["apply"][0])
    bytes_per = 12.0 * N_in + 8.0 * U
    launches = max(1, prof["apply"][0])
    achieved = bytes_per * (max(1, sampled) / launches) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    return {
        "metric": "scans/s (100k-pt scan stream, %s TSDF insert into 3 hashed-block grids)" % args.insert_mode,
        "value": args.steps * B / elapsed, "unit": "scans/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32/u16", "data": "synthetic",
        "config": {"workload": "insert_stream: %d scans x %d pts per batched call, 3-res TSDF, %s mode"
                               % (B, n_pts, args.insert_mode), "updates_per_step": U, "hits_per_step": N_in,
                   "insert_mode": args.insert_mode, "room_copies": args.stream_tiles, "blocks_per_level": blocks,
                   "voxel_working_set_mib": sum(blocks) * 2048 / 2.0 ** 20},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "insert family (expand+sort+alloc+apply) per chunk launch",
                     "avg_launch_ms": avg_ms, "algorithmic_bytes_per_step": bytes_per,
                     "hip_event_sampling": "every %d-th of the %d timed steps" % (max(1, args.prof_every), args.steps),
                     "per_kernel_ms_total": {k: round(v[1], 3) for k, v in prof.items()},
                     "per_kernel_launches": {k: v[0] for k, v in prof.items()}},
        "parity": parity, "cpu_baseline": base,
        "gpu_over_cpu": (args.steps * B / elapsed) / base["value"] if base else None,
    }


def run_c1_10k(args):
    """BASELINE configs[0]: a 10k-point scan (16 rings x 625 columns) registered into ONE 0.10 m TSDF (+ the 0.45 m
    low-resolution grid every Submap3D carries, submap_3d.h:89-90). Two sub-rows:
      all_points   every return matched (single-resolution LM on the 0.10 m grid) + exact insertion into both grids, one
                   hg_register_scan call per scan from this process; oracle beside it, poses gated at 1e-4 m / 1e-4 rad
      lua_default  the shipped C++ OptimizingLocalTrajectoryBuilder (cpp/hg_adapter.h) with the options of
                   trajectory_builder_3d.lua -- adaptive voxel filters (>= 150 / 200 points), CONSTANT control points,
                   odometry + IMU blocks -- over cpp/example_oltb's sensor stream at 16 x 625 returns per scan; the
                   value is scans / time inside AddRangeData; gate and CPU figure: the same messages replayed through the
                   Python statement of optimizing_local_trajectory_builder.cc over the oracle (tests/oltb_replay.py)."""
    import subprocess
    import tempfile
    import torch
    from hectorgrapher_amd import api, synth
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    dev = torch.device("cuda", 0)
    rings, cols = 16, 625
    n_pts = rings * cols
    res = (0.10, 0.45)
    out = {}
    # ---- all_points ----
    ctx = api.Context(0)
    grids = [api.HybridGridTSDF(ctx, r, max_blocks=1 << 15) for r in res]
    ins = [api.TSDFRangeDataInserter3D() for _ in res]
    map_scans = make_scans(rings, cols, 0, args.map_scans, 0)
    for pose, pts in map_scans:
        api.insert_pyramid(ins, api.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), grids, pose_tq=pose.astype(np.float32))
    query = make_scans(rings, cols, args.map_scans, args.warmup + args.steps, 0)
    d_scans = [torch.from_numpy(pts).to(dev) for _, pts in query]
    guesses = [synth.pose_mul(pose, synth.perturbation()) for pose, _ in query]
    scale = 1.0 / np.sqrt(float(n_pts))
    problem = api.Problem(ctx)
    gpu_steps = []

    def step(i):
        problem.reset()
        pi = problem.add_pose(guesses[i])
        problem.add_block(d_scans[i], [grids[0]], scale, pi, multi_res=False, width=rings)
        est, summ = api.register_scan(problem, pi, ins, api.RangeData([0, 0, 0], d_scans[i], width=rings), grids)
        gpu_steps.append((est, summ.num_iterations, summ.termination_type, summ.termination_reason))

    for i in range(args.warmup):
        step(i)
    ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    og = [po.Grid(r) for r in res]
    for pose, pts in map_scans:
        loc = synth.transform_points(pose, pts)
        for g in og:
            g.insert(pose[:3], loc)
    n_cpu = min(args.cpu_scans + args.warmup, len(query))
    max_dt = max_dr = 0.0
    same = True
    t1 = time.perf_counter()
    for i in range(n_cpu):
        pr = po.Problem()
        pi = pr.add_pose(guesses[i])
        pr.add_block(query[i][1], [og[0]], scale, pi, multi_res=False)
        so = pr.solve()
        o = pr.get_pose(pi)
        est, it, tt, tr = gpu_steps[i]
        # (as cpu_baseline: the next step's map must be the GPU's -- insert where the GPU inserted if the float casts differ)
        at = o if np.array_equal(o.astype(np.float32), est.astype(np.float32)) else est
        loc = synth.transform_points(at, query[i][1])
        for g in og:
            g.insert(at[:3].astype(np.float32), loc)
        max_dt = max(max_dt, float(np.linalg.norm(o[:3] - est[:3])))
        max_dr = max(max_dr, float(2.0 * np.arccos(min(1.0, abs(float(np.dot(o[3:], est[3:])))))))
        same = same and (so.num_iterations, so.termination_type, so.termination_reason) == (it, tt, tr)
    cpu_rate = n_cpu / (time.perf_counter() - t1)
    problem.close()
    for g in grids:
        g.close()
    ctx.close()
    out["all_points"] = {"value": args.steps / elapsed, "unit": "scans/s", "ms_per_step": elapsed / args.steps * 1e3, "steps": args.steps,
                         "parity": {"max_dt_m": max_dt, "max_dr_rad": max_dr, "scans": n_cpu, "tolerance": 1e-4,
                                    "same_iterations_and_termination": bool(same)},
                         "cpu_baseline": {"value": cpu_rate, "unit": "scans/s", "cores": 1, "kind": "port",
                                          "sample": "%d scans of the same workload, oracle -O3 1 thread" % n_cpu},
                         "gpu_over_cpu": (args.steps / elapsed) / cpu_rate}
    if not (max_dt <= 1e-4 and max_dr <= 1e-4):
        raise SystemExit("bench.py: parity gate of c1_10k failed: %r" % (out["all_points"]["parity"],))
    # ---- lua_default: the C++ builder ----
    exe = os.path.join(ROOT, "hectorgrapher_amd", "cpp", "example_oltb")
    scans = 40
    try:
        if not os.path.exists(exe):
            raise RuntimeError("hectorgrapher_amd/cpp/example_oltb is not built (python -c 'import __graft_entry__ as g; g.build()')")
        with tempfile.TemporaryDirectory() as td:
            dump = os.path.join(td, "oltb.bin")
            run_ = subprocess.run([exe, dump, "0", str(scans), str(rings), str(cols), "1"], capture_output=True, text=True, timeout=600)
            if run_.returncode != 0:
                raise RuntimeError(run_.stderr[-500:])
            m = [l for l in run_.stdout.splitlines() if l.startswith("timing:")][-1].split()
            gpu_s = float(m[6])
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oltb_replay as rp
            r = rp.replay_and_compare(po, open(dump, "rb").read(), run_.stdout, 0, scans, width=rings, check_window=False)
        ok = max(r["b"].cloud_errors) < 1e-5 if r["b"].cloud_errors else True
        out["lua_default"] = {"value": scans / gpu_s, "unit": "scans/s", "ms_per_step": gpu_s / scans * 1e3, "steps": scans,
                              "solves": r["solves"],
                              "parity": {"same_solves_iterations_terminations_and_blocks": True, "inserted_range_data_max_err_m":
                                         float(max(r["b"].cloud_errors)) if r["b"].cloud_errors else 0.0, "ok": bool(ok),
                                         "check": "every step of the C++ builder against tests/oltb_replay.py over the oracle: solve / no solve, "
                                                  "iterations, termination, block wiring, result and insertion decisions, inserted range data"},
                              "cpu_baseline": {"value": scans / r["cpu_seconds"], "unit": "scans/s", "cores": 1, "kind": "port",
                                               "sample": "the same %d scans: Python statement of optimizing_local_trajectory_builder.cc "
                                                         "driving the -O3 oracle (inserts, filters and solves in C; orchestration in Python)" % scans},
                              "gpu_over_cpu": (scans / gpu_s) / (scans / r["cpu_seconds"])}
        if not ok:
            raise SystemExit("bench.py: parity gate of c1_10k lua_default failed")
    except AssertionError as e:
        raise SystemExit("bench.py: c1_10k lua_default differs from the replay: %r" % (e,))
    v = out["all_points"]
    return {"metric": "scans/s (10k-pt scan into one 0.10 m TSDF, BASELINE configs[0])", "value": v["value"], "unit": "scans/s",
            "ms_per_step": v["ms_per_step"], "steps": args.steps, "n_gpus": 1, "warmup": args.warmup, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "c1_10k: 16 x 625 returns, one 0.10 m TSDF (+ 0.45 m low-resolution grid): all points matched "
                                   "(value) and the Lua-default builder (rows.lua_default)"},
            "roofline": None, "parity": v["parity"], "cpu_baseline": v["cpu_baseline"], "rows": out}


def window_spec(synth, first, n_cp):
    """Numbers of one OptimizingLocalTrajectoryBuilder-shaped window (control points first .. first +
    n_cp - 1): initial poses (the first one at ground truth, the others perturbed), odometry deltas and
    IMU delta rotations between neighbours. Pure host arithmetic, prepared outside the timed loop."""
    poses = [synth.pose_k(first) if i == 0 else synth.pose_mul(synth.pose_k(first + i), synth.perturbation())
             for i in range(n_cp)]
    deltas, dqs = [], []
    for i in range(1, n_cp):
        k = first + i
        deltas.append(synth.pose_mul(synth.pose_inverse(synth.pose_k(k)), synth.pose_k(k - 1)))
        dqs.append(synth.pose_mul(synth.pose_inverse(synth.pose_k(k - 1)), synth.pose_k(k))[3:])
    return {"poses": poses, "deltas": deltas, "dqs": dqs, "velocity": np.array([0.5, 0.2, 0.0])}


def window_build(pr, spec, clouds, pyramid, n_pts):
    """The problem of a window: n_cp control points (first constant, oltb.cc:1268-1275) with velocities,
    IMU pre-integration + odometry blocks between neighbours (:928-1074), one multi-resolution scan block
    per free control point (:343-364)."""
    n_cp = len(spec["poses"])
    for i in range(n_cp):
        pr.add_pose(spec["poses"][i], i == 0)
        pr.set_velocity(i, spec["velocity"], i == 0)
    scale = 1.0 / np.sqrt(float(n_pts))
    for i in range(1, n_cp):
        pr.add_odometry_block(i - 1, i, 12.0, 30.0, spec["deltas"][i - 1])
        pr.add_imu_block(i - 1, i, 3.0, 2.0, 70.0, 0.1, spec["dqs"][i - 1])
        pr.add_block(clouds[i - 1], pyramid, scale, i, multi_res=True)


def window_prepare(pr, spec, clouds, pyramid, n_pts, width=0):
    """window_build with the arguments marshalled up front: returns run() that makes the same C-ABI calls
    (hg_problem_reset, add_pose, set_velocity, add_odometry_block, add_imu_block, add_block) with nothing but
    the foreign calls inside -- what a C++ host spends on building a window's problem. The timed loop of
    run_window uses this; the oracle leg and the tests use window_build."""
    import ctypes as C
    from hectorgrapher_amd import _lib
    L = _lib.load()
    h = pr._h
    n_cp = len(spec["poses"])
    keep = []

    def ptr(a, dtype):
        a = np.ascontiguousarray(a, dtype)
        keep.append(a)
        return a.ctypes.data_as(C.c_void_p)

    vel = ptr(spec["velocity"], np.float64)
    garr = (C.c_void_p * len(pyramid))(*[g._h for g in pyramid])
    keep.append(garr)
    scale = 1.0 / np.sqrt(float(n_pts))
    calls = [(L.hg_problem_reset, (h,))]
    for i in range(n_cp):
        calls.append((L.hg_problem_add_pose, (h, ptr(spec["poses"][i], np.float64), int(i == 0))))
        calls.append((L.hg_problem_set_velocity, (h, i, vel, int(i == 0))))
    for i in range(1, n_cp):
        calls.append((L.hg_problem_add_odometry_block, (h, i - 1, i, 12.0, 30.0, ptr(spec["deltas"][i - 1], np.float64))))
        calls.append((L.hg_problem_add_imu_block, (h, i - 1, i, 3.0, 2.0, 70.0, 0.1, ptr(spec["dqs"][i - 1], np.float64))))
        cloud = clouds[i - 1]
        keep.append(cloud)
        calls.append((L.hg_problem_add_block, (h, cloud.data_ptr(), int(cloud.shape[0]), _lib.HG_DEVICE, garr, len(pyramid), 1,
                                               scale, i, -1, 0.0)))
        if width:
            calls.append((L.hg_problem_set_block_width, (h, i - 1, int(width))))

    def run():
        for f, a in calls:
            if f(*a) < 0:
                raise RuntimeError("window_prepare: a problem call failed: %s" % L.hg_last_error().decode())
        return keep

    return run


def window_problem(pr, synth, first, n_cp, clouds, pyramid, n_pts):
    window_build(pr, window_spec(synth, first, n_cp), clouds, pyramid, n_pts)


def run_window(args):
    """Sliding-window registration (the OptimizingLocalTrajectoryBuilder shape): every step solves
    a window of --window control points over (window - 1) 100k-point scans against the 3-resolution
    TSDF, then inserts the scan that leaves the window at its solved pose. Extra workload, not the
    headline metric."""
    import torch
    from hectorgrapher_amd import api, synth
    dev = torch.device("cuda", 0)
    ctx = api.Context(0)
    n_pts = args.rings * args.cols
    n_cp = args.window
    map_scans = make_scans(args.rings, args.cols, 0, args.map_scans, 0)
    total = args.warmup + args.steps
    # poses beyond k = 60 leave the folded range of make_scans: keep windows inside [map_scans, 60]
    scans = [synth.generate_scan(synth.pose_k(args.map_scans + j), args.rings, args.cols, stream=args.map_scans + j)
             for j in range(total + n_cp)]
    grids = [api.HybridGridTSDF(ctx, r, max_blocks=args.max_blocks) for r in RESOLUTIONS]
    inserters = [api.TSDFRangeDataInserter3D() for _ in grids]
    for pose, pts in map_scans:
        api.insert_pyramid(inserters, api.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), grids,
                           pose_tq=pose.astype(np.float32))
    d_scans = [torch.from_numpy(p).to(dev) for p in scans]
    torch.cuda.synchronize()
    problem = api.Problem(ctx)
    its, evals, inserted_at, solved = [], [], [], []

    # control point 0 of window s sits on the last inserted scan
    specs = [window_spec(synth, args.map_scans - 1 + s, n_cp) for s in range(total)]
    builds = [window_prepare(problem, specs[s], d_scans[s:s + n_cp - 1], grids, n_pts, args.rings) for s in range(total)]

    leaving = [api.RangeData([0, 0, 0], d_scans[s], width=args.rings) for s in range(total)]
    # Per-point unwarping of the scan that leaves the window (use_per_point_unwarping, oltb.cc:1331-1379): control
    # point i of window s sits at time (first + i) * 0.1 s; the leaving scan belongs to control point 1. Its
    # insertion copy is taken by a MOVING sensor (round 5): the columns are stamped over a 0.1 s sweep from 60 ms
    # before to 40 ms behind control point 1 and measured from the pose the trajectory has at that time
    # (synth.generate_swept_scan), so the returns fall between control points 0..1 AND 1..2 -- two control-point
    # pairs, every return its own interpolation factor -- and only the unwarping puts them where the map expects
    # them. (The matching blocks keep the standing-sensor scans.) --no-window-unwarp inserts at control point 1's pose.
    unwarp = not args.no_window_unwarp
    cp_dt = 1_000_000  # 0.1 s in ticks, the delta_time of the IMU blocks
    d_timed = timed_host = None
    if unwarp:
        col_t = np.linspace(-0.06, 0.04, args.cols).astype(np.float32)
        pt_t = np.repeat(col_t, args.rings)[:, None]
        timed_host = []
        for s_ in range(total):
            k1 = args.map_scans + s_
            swept = synth.generate_swept_scan(lambda c: synth.pose_k(k1 + float(col_t[c]) / 0.1), args.rings, args.cols,
                                              stream=args.map_scans + s_)
            timed_host.append(np.ascontiguousarray(np.concatenate([swept, pt_t], 1), np.float32))
        d_timed = [torch.from_numpy(t).to(dev) for t in timed_host]
    torch.cuda.synchronize()

    def control_times(s):
        first = args.map_scans - 1 + s
        return np.array([(first + i) * cp_dt for i in range(n_cp)], np.int64)

    def step(s, sample=False):
        builds[s]()
        if unwarp:
            # solve the window, then unwarp + insert the leaving scan with the SOLVED control poses read from
            # device memory (hg_register_scan_unwarped): no host round trip between solve and insertion
            ct = control_times(s)
            poses, summ = api.register_scan_unwarped(problem, inserters, [(int(ct[1]), [0, 0, 0], d_timed[s])], args.rings,
                                                     list(range(n_cp)), ct, grids)
            at = poses[1]
        else:
            # solve the window, then insert the scan that leaves it at control point 1's solved pose, handed
            # over in device memory (hg_register_scan_mode works on any problem shape): the insertion is
            # enqueued behind the solve before the host has seen its result
            at, summ = api.register_scan(problem, 1, inserters, leaving[s], grids)
            poses = np.array([problem.get_pose(i) for i in range(n_cp)])
        its.append(summ.num_iterations)
        if sample:
            evals.append(summ.num_cost_evaluations)
        solved.append((poses.copy(), summ.num_iterations, summ.termination_type, summ.termination_reason))
        inserted_at.append(at.astype(np.float32))

    for s_ in range(args.warmup):
        step(s_)
    its.clear()
    ctx.prof_reset()
    ctx.synchronize()
    t0 = time.perf_counter()
    for s_ in range(args.warmup, total):
        on = args.prof_every > 0 and (s_ - args.warmup) % args.prof_every == 0
        ctx.prof_enable(on)
        step(s_, on)
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    prof = ctx.prof_read()
    ctx.prof_enable(False)
    base = None
    parity = None
    lbar = 4.0 / 3.0
    if not args.no_cpu_baseline:
        # the oracle replays the FIRST TIMED step: same map (the warm-up scans inserted where the GPU
        # inserted them), same window, same guesses -> its poses are the parity gate of that step
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle as po
        og = [po.Grid(r) for r in RESOLUTIONS]
        for pose, pts in map_scans:
            loc = synth.transform_points(pose, pts)
            for g in og:
                g.insert(pose[:3], loc)
        def oracle_insert(w, poses):
            """Insertion of the scan leaving window w at the window's `poses`, as the GPU step does it."""
            if unwarp:
                ct = control_times(w)
                xyz, origin, ok = po.unwarp_range_data(ct, poses, [(int(ct[1]), [0, 0, 0], timed_host[w])])
                assert ok
                opt = np.asarray(poses[0], np.float64).astype(np.float32)  # optimized_pose.cast<float>() (:1437-1440)
                xyz = po.transform_points(opt, xyz)
                origin = po.transform_points(opt, origin[None])[0]
                for g in og:
                    g.insert(origin, xyz, width=args.rings)
            else:
                at = np.asarray(poses[1])
                loc = synth.transform_points(at, scans[w])
                for g in og:
                    g.insert(at[:3].astype(np.float32), loc)

        for w in range(args.warmup):
            oracle_insert(w, solved[w][0])
        s0 = args.warmup
        t1 = time.perf_counter()
        pr = po.Problem()
        window_problem(pr, synth, args.map_scans - 1 + s0, n_cp, scans[s0:s0 + n_cp - 1], og, n_pts)
        so = pr.solve()
        oracle_insert(s0, np.array([pr.get_pose(i) for i in range(n_cp)]))
        cpu_s = time.perf_counter() - t1
        lk, pb = pr.lookup_stats()
        lbar = pb / max(1, lk)
        g_poses, g_it, g_tt, g_tr = solved[s0]
        max_dt = max_dr = 0.0
        for i in range(n_cp):
            o = pr.get_pose(i)
            max_dt = max(max_dt, float(np.linalg.norm(o[:3] - g_poses[i][:3])))
            max_dr = max(max_dr, float(2.0 * np.arccos(min(1.0, abs(float(np.dot(o[3:], g_poses[i][3:])))))))
        parity = {"max_dt_m": max_dt, "max_dr_rad": max_dr, "windows": 1, "control_points": n_cp, "tolerance": 1e-4,
                  "same_iterations_and_termination": bool((so.num_iterations, so.termination_type, so.termination_reason)
                                                          == (g_it, g_tt, g_tr)),
                  "step": "first timed step, %d x %d-pt blocks" % (n_cp - 1, n_pts)}
        if not (max_dt <= 1e-4 and max_dr <= 1e-4):
            raise SystemExit("bench.py: parity gate failed, GPU and oracle window poses differ: %r" % (parity,))
        base = {"value": 1.0 / cpu_s, "unit": "scans/s", "cores": 1, "kind": "port",
                "sample": "1 window of the same workload (solve %d iterations + insert), oracle -O3 1 thread" % so.num_iterations}
    # roofline of the window pass (k_window_residuals): every launch evaluates all blocks of the window;
    # launches behind the solve's termination exit at once and move no bytes
    n_launch = max(1, prof["residuals"][0])
    avg_ms = prof["residuals"][1] / n_launch
    share = min(1.0, sum(evals) / n_launch) if evals else 1.0
    bytes_per_launch = (n_cp - 1) * n_pts * (12.0 + 32.0 * lbar) * share
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    return {
        "metric": "scans/s (sliding window of %d control points over %d x 100k-pt scans, 3-res TSDF)" % (n_cp, n_cp - 1),
        "value": args.steps / elapsed, "unit": "scans/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "window: %d control points (81 free columns at 10), %d multi-res scan blocks + IMU/odometry blocks per solve, exact insert of the scan leaving the window%s"
                               % (n_cp, n_cp - 1, " after per-point unwarping on the device (hg_register_scan_unwarped): the leaving scan is taken by a moving sensor, its "
                                  "columns stamped over a 0.1 s sweep across control point 1, so its returns interpolate between two control-point pairs" if unwarp else ""),
                   "mean_lm_iterations": float(np.mean(its)),
                   "problem_build": "pre-marshalled C-ABI calls per window (window_prepare); initial guesses from the synthetic ground truth + a fixed perturbation, not from the previous solve",
                   "unwarp_ms_per_call": (prof["unwarp"][1] / max(1, prof["unwarp"][0])) if unwarp else None},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "k_window_residuals<false>",
                     "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                     "launches_evaluating": share, "mean_levels_probed": lbar,
                     "lm_avg_launch_ms": prof["lm"][1] / max(1, prof["lm"][0]),
                     "per_kernel_ms_total": {k: round(v[1], 4) for k, v in prof.items()},
                     "per_kernel_launches": {k: v[0] for k, v in prof.items()},
                     "hip_event_sampling": "every %d-th of the %d timed steps" % (max(1, args.prof_every), args.steps)},
        "parity": parity, "cpu_baseline": base,
        "gpu_over_cpu": (args.steps / elapsed) / base["value"] if base else None,
    }


def run_window_batch(args):
    """BASELINE configs[3] with the reference's real builder: S independent submaps, each advanced by the
    OptimizingLocalTrajectoryBuilder step -- a sliding window of --window control points over (window - 1)
    100k-point scans, IMU + odometry blocks -- with shared launches: hg_register_scan_batch on S window problems
    (k_window_residuals / k_lm over a table of problems, grid row = problem), then the scans leaving the
    windows inserted at control point 1's solved pose through the insert kernels' job table. Submap 0's first
    timed step is replayed by the oracle inside the run."""
    import torch
    from hectorgrapher_amd import api, synth
    dev = torch.device("cuda", 0)
    ctx = api.Context(0)
    n_pts = args.rings * args.cols
    n_cp, S = args.window, args.batch_submaps
    total = args.warmup + args.steps
    inserters = [api.TSDFRangeDataInserter3D() for _ in RESOLUTIONS]
    subs = []
    for j in range(S):
        sb = 100000 * (j + 1) if j else 0  # PRNG streams of submap j (submap 0 = the single-window workload's)
        map_scans = make_scans(args.rings, args.cols, 0, args.map_scans, sb)
        scans = [synth.generate_scan(synth.pose_k(args.map_scans + k), args.rings, args.cols, stream=sb + args.map_scans + k)
                 for k in range(total + n_cp)]
        grids = [api.HybridGridTSDF(ctx, r, max_blocks=args.max_blocks) for r in RESOLUTIONS]
        for pose, pts in map_scans:
            api.insert_pyramid(inserters, api.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), grids,
                               pose_tq=pose.astype(np.float32))
        d_scans = [torch.from_numpy(p_).to(dev) for p_ in scans]
        problem = api.Problem(ctx)
        specs = [window_spec(synth, args.map_scans - 1 + s_, n_cp) for s_ in range(total)]
        builds = [window_prepare(problem, specs[s_], d_scans[s_:s_ + n_cp - 1], grids, n_pts, args.rings) for s_ in range(total)]
        subs.append({"map": map_scans, "scans": scans, "d": d_scans, "grids": grids, "problem": problem, "builds": builds})
    torch.cuda.synchronize()
    its, evals, solved0, inserted0 = [], [], [], []
    problems = [sub["problem"] for sub in subs]
    pyramids = [sub["grids"] for sub in subs]

    def step(s_, sample=False):
        for sub in subs:
            sub["builds"][s_]()
        poses, summ = api.register_scan_batch(problems, [1] * S, inserters,
                                              [api.RangeData([0, 0, 0], sub["d"][s_], width=args.rings) for sub in subs], pyramids)
        its.extend(x.num_iterations for x in summ)
        if sample:
            evals.append(sum(x.num_cost_evaluations for x in summ) / S)
        solved0.append((np.array([problems[0].get_pose(i) for i in range(n_cp)]), summ[0].num_iterations,
                        summ[0].termination_type, summ[0].termination_reason))
        inserted0.append(poses[0].astype(np.float32))

    for s_ in range(args.warmup):
        step(s_)
    its.clear()
    ctx.prof_reset()
    ctx.synchronize()
    t0 = time.perf_counter()
    for s_ in range(args.warmup, total):
        on = args.prof_every > 0 and (s_ - args.warmup) % args.prof_every == 0
        ctx.prof_enable(on)
        step(s_, on)
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    prof = ctx.prof_read()
    ctx.prof_enable(False)
    for sub in subs:
        for g in sub["grids"]:
            g.status()
    base = parity = None
    lbar = 4.0 / 3.0
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle as po
        sub = subs[0]
        og = [po.Grid(r) for r in RESOLUTIONS]
        for pose, pts in sub["map"]:
            loc = synth.transform_points(pose, pts)
            for g in og:
                g.insert(pose[:3], loc)
        for w in range(args.warmup):
            loc = synth.transform_points(inserted0[w], sub["scans"][w])
            for g in og:
                g.insert(inserted0[w][:3], loc)
        s0 = args.warmup
        t1 = time.perf_counter()
        pr = po.Problem()
        window_problem(pr, synth, args.map_scans - 1 + s0, n_cp, sub["scans"][s0:s0 + n_cp - 1], og, n_pts)
        so = pr.solve()
        est = pr.get_pose(1)
        loc = synth.transform_points(est, sub["scans"][s0])
        for g in og:
            g.insert(est[:3].astype(np.float32), loc)
        cpu_s = time.perf_counter() - t1
        lk, pb = pr.lookup_stats()
        lbar = pb / max(1, lk)
        g_poses, g_it, g_tt, g_tr = solved0[s0]
        max_dt = max_dr = 0.0
        for i in range(n_cp):
            o = pr.get_pose(i)
            max_dt = max(max_dt, float(np.linalg.norm(o[:3] - g_poses[i][:3])))
            max_dr = max(max_dr, float(2.0 * np.arccos(min(1.0, abs(float(np.dot(o[3:], g_poses[i][3:])))))))
        parity = {"max_dt_m": max_dt, "max_dr_rad": max_dr, "windows": 1, "control_points": n_cp, "tolerance": 1e-4,
                  "same_iterations_and_termination": bool((so.num_iterations, so.termination_type, so.termination_reason)
                                                          == (g_it, g_tt, g_tr)),
                  "step": "submap 0 of the batch, its first timed window (%d x %d-pt blocks)" % (n_cp - 1, n_pts)}
        if not (max_dt <= 1e-4 and max_dr <= 1e-4):
            raise SystemExit("bench.py: parity gate failed, GPU and oracle window poses differ: %r" % (parity,))
        base = {"value": 1.0 / cpu_s, "unit": "scans/s", "cores": 1, "kind": "port",
                "sample": "1 window of the same workload (solve %d iterations + insert), oracle -O3 1 thread" % so.num_iterations}

        def one_window(state):  # the first window of a submap of its own: solve + insert
            pr_ = po.Problem()
            window_problem(pr_, synth, args.map_scans - 1, n_cp, sub["scans"][0:n_cp - 1], state, n_pts)
            pr_.solve()
            e_ = pr_.get_pose(1)
            l_ = synth.transform_points(e_, sub["scans"][0])
            for g_ in state:
                g_.insert(e_[:3].astype(np.float32), l_)
        base["all_cores"] = all_cores_baseline(args, lambda: oracle_map(po, synth, sub["map"]), one_window, 1,
                                               "1 window step (solve + insert) of an independent submap")
    n_launch = max(1, prof["residuals"][0])
    avg_ms = prof["residuals"][1] / n_launch
    share = min(1.0, sum(evals) / n_launch) if evals else 1.0
    bytes_per_launch = S * (n_cp - 1) * n_pts * (12.0 + 32.0 * lbar) * share
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    value = args.steps * S / elapsed
    return {
        "metric": "scans/s (%d submaps x sliding window of %d control points over %d x 100k-pt scans, 3-res TSDF)" % (S, n_cp, n_cp - 1),
        "value": value, "unit": "scans/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "window_batch: %d independent submaps, each a window of %d control points (%d multi-res scan blocks + IMU/odometry "
                               "blocks) solved with shared launches (hg_register_scan_batch), exact insert of every leaving scan"
                               % (S, n_cp, n_cp - 1),
                   "submaps": S, "mean_lm_iterations": float(np.mean(its)),
                   "problem_build": "pre-marshalled C-ABI calls per window (window_prepare); initial guesses from the synthetic ground truth + a fixed perturbation",
                   "resident_voxel_gib": S * len(RESOLUTIONS) * (2 * args.max_blocks) * 2048 / 2.0 ** 30},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "k_window_residuals_jobs<false>",
                     "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                     "launches_evaluating": share, "mean_levels_probed": lbar,
                     "lm_avg_launch_ms": prof["lm"][1] / max(1, prof["lm"][0]),
                     "per_kernel_ms_total": {k: round(v[1], 4) for k, v in prof.items()},
                     "per_kernel_launches": {k: v[0] for k, v in prof.items()},
                     "hip_event_sampling": "every %d-th of the %d timed steps" % (max(1, args.prof_every), args.steps)},
        "parity": parity, "cpu_baseline": base,
        "gpu_over_cpu": value / base["value"] if base else None,
    }


def run_register_filtered(args):
    """Context run: the registration step with the reference's Lua-default matching
    set -- the adaptive voxel filter (max_length 2 m, >= 150 points, max_range 15 m,
    trajectory_builder_3d.lua:23-27) picks the points the matcher sees; ALL points are inserted at the
    solved pose. Extra workload, not the headline metric."""
    import torch
    from hectorgrapher_amd import api, synth
    dev = torch.device("cuda", 0)
    ctx = api.Context(0)
    n_pts = args.rings * args.cols
    map_scans = make_scans(args.rings, args.cols, 0, args.map_scans, 0)
    query = make_scans(args.rings, args.cols, args.map_scans, args.warmup + args.steps, 0)
    grids = [api.HybridGridTSDF(ctx, r, max_blocks=args.max_blocks) for r in RESOLUTIONS]
    inserters = [api.TSDFRangeDataInserter3D() for _ in grids]
    for pose, pts in map_scans:
        api.insert_pyramid(inserters, api.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), grids,
                           pose_tq=pose.astype(np.float32))
    d_scans = [torch.from_numpy(pts).to(dev) for _, pts in query]
    guesses = [synth.pose_mul(pose, synth.perturbation()) for pose, _ in query]
    avf = api.AdaptiveVoxelFilter(ctx, 2.0, 150, 15.0)
    problem = api.Problem(ctx)
    errs, kept = [], []
    torch.cuda.synchronize()

    def step(i):
        idx = avf.Filter(d_scans[i])
        sel = d_scans[i][torch.from_numpy(idx.astype(np.int64)).to(dev)].contiguous()
        problem.reset()
        pi = problem.add_pose(guesses[i])
        problem.add_block(sel, grids, 1.0 / np.sqrt(float(len(idx))), pi, multi_res=True)
        est, _ = api.register_scan(problem, pi, inserters, api.RangeData([0, 0, 0], d_scans[i], width=args.rings), grids)
        errs.append(float(np.linalg.norm(est[:3] - query[i][0][:3])))
        kept.append(len(idx))

    for i in range(args.warmup):
        step(i)
    errs.clear()
    kept.clear()
    ctx.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    ctx.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    base = None
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle as po
        og = [po.Grid(r) for r in RESOLUTIONS]
        for pose, pts in map_scans:
            loc = synth.transform_points(pose, pts)
            for g in og:
                g.insert(pose[:3], loc)
        t_cpu, n_cpu = 0.0, 0
        for pose, pts in query[args.warmup:args.warmup + args.cpu_scans]:
            guess = synth.pose_mul(pose, synth.perturbation())
            t1 = time.perf_counter()
            sel = pts[po.adaptive_voxel_filter(2.0, 150, 15.0, pts)]
            pr = po.Problem()
            i = pr.add_pose(guess)
            pr.add_block(sel, og, 1.0 / np.sqrt(len(sel)), i, multi_res=True)
            pr.solve()
            est = pr.get_pose(i)
            loc = synth.transform_points(est, pts)
            for g in og:
                g.insert(est[:3].astype(np.float32), loc)
            t_cpu += time.perf_counter() - t1
            n_cpu += 1
        base = {"value": n_cpu / t_cpu, "unit": "scans/s", "cores": 1, "kind": "port",
                "sample": "%d scans of the same workload (filter + match + 3-level insert), oracle -O3 1 thread" % n_cpu}
    return {
        "metric": "scans/s (100k-pt scan, adaptive-voxel-filtered match + full 3-res TSDF insert)",
        "value": args.steps / elapsed, "unit": "scans/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "register_filtered: AdaptiveVoxelFilter(2 m, >=150 pts, 15 m) matching set, "
                               "multi-res LM match, exact insert of all %d points x3" % n_pts,
                   "mean_matched_points": float(np.mean(kept)), "mean_pose_error_m": float(np.mean(errs))},
        "roofline": None, "cpu_baseline": base,
        "gpu_over_cpu": (args.steps / elapsed) / base["value"] if base else None,
    }


def run_match_batch(args):
    """Independent scan-to-submap matches solved together (hg_problem_solve_batch): the shape of a
    constraint search, which hands one finished submap and many scans to CeresScanMatcher3D::Match. One
    step = --batch matches of different 100k-point scans against the same 3-resolution TSDF, no
    insertion. Extra workload, not the headline metric."""
    import torch
    from hectorgrapher_amd import api, synth
    dev = torch.device("cuda", 0)
    ctx = api.Context(0)
    n_pts = args.rings * args.cols
    B = args.batch
    map_scans = make_scans(args.rings, args.cols, 0, args.map_scans, 0)
    pyramids = []
    for _ in range(max(1, args.batch_maps)):
        grids = [api.HybridGridTSDF(ctx, r, max_blocks=args.max_blocks) for r in RESOLUTIONS]
        inserters = [api.TSDFRangeDataInserter3D() for _ in grids]
        for pose, pts in map_scans:
            api.insert_pyramid(inserters, api.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), grids,
                               pose_tq=pose.astype(np.float32))
        pyramids.append(grids)
    # queries around the mapped stretch of the trajectory
    queries = []
    for j in range(B):
        k = j % args.map_scans
        pose = synth.pose_k(k)
        pts = synth.generate_scan(pose, args.rings, args.cols, stream=5000 + j)
        queries.append((pose, pts, torch.from_numpy(pts).to(dev), synth.pose_mul(pose, synth.perturbation())))
    torch.cuda.synchronize()
    scale = 1.0 / np.sqrt(float(n_pts))
    # --batches-in-flight 2 (default): the search keeps two sets of problem handles; while the device solves batch k the
    # host builds batch k + 1 on the other set and enqueues it (hg_problem_solve_batch_async), then fetches batch k --
    # the device does not wait for the host between batches. 1 = hg_problem_solve_batch, one batch at a time.
    depth = max(1, args.batches_in_flight)
    sets = [[api.Problem(ctx) for _ in range(B)] for _ in range(depth)]
    problems = sets[0]
    stats = {"its": [], "evals": 0}
    flight = []  # (set index, sampled)

    def collect():
        k, sample = flight.pop(0)
        summ = api.fetch_batch(sets[k])
        stats["its"].append(np.mean([s_.num_iterations for s_ in summ]))
        stats["last"] = summ
        stats["last_set"] = k
        if sample:
            stats["evals"] += sum(s_.num_cost_evaluations for s_ in summ)

    def step(sample, number):
        k = number % depth
        target = pyramids[(number + number // depth) % len(pyramids)]  # (a handle set meets the copies in turn)
        for p, (_, _, d, guess) in zip(sets[k], queries):
            p.reset()
            i = p.add_pose(guess)
            p.add_block(d, target, scale, i, multi_res=True, width=args.rings)
        if depth == 1:
            summ = api.solve_batch(sets[k])
            stats["its"].append(np.mean([s_.num_iterations for s_ in summ]))
            stats["last"] = summ
            stats["last_set"] = k
            if sample:
                stats["evals"] += sum(s_.num_cost_evaluations for s_ in summ)
            return
        api.solve_batch_async(sets[k])
        flight.append((k, sample))
        if len(flight) >= depth:
            collect()

    for w in range(max(args.warmup, depth)):  # (every set of handles has allocated its buffers)
        step(False, w)
    while flight:
        collect()
    stats = {"its": [], "evals": 0}
    ctx.prof_reset()
    ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        sampling = args.prof_every > 0 and i % args.prof_every == 0
        ctx.prof_enable(sampling)
        step(sampling, i)
    while flight:
        collect()
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    problems = sets[stats["last_set"]]
    prof = ctx.prof_read()
    ctx.prof_enable(False)
    errs = [float(np.linalg.norm(p.get_pose(0)[:3] - q[0][:3])) for p, q in zip(problems, queries)]
    last = stats["last"]  # every step solves the same matches: the last timed step's results are the gate's
    base = None
    parity = None
    lbar = 4.0 / 3.0
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle as po
        og = [po.Grid(r) for r in RESOLUTIONS]
        for pose, pts in map_scans:
            loc = synth.transform_points(pose, pts)
            for g in og:
                g.insert(pose[:3], loc)
        t1 = time.perf_counter()
        n_cpu = min(B, args.cpu_scans)
        lk = pb = 0
        max_dt = max_dr = 0.0
        same_flow = True
        for j, (pose, pts, _, guess) in enumerate(queries[:n_cpu]):
            pr = po.Problem()
            i = pr.add_pose(guess)
            pr.add_block(pts, og, scale, i, multi_res=True)
            so = pr.solve()
            a, b = pr.lookup_stats()
            lk += a
            pb += b
            o, g = pr.get_pose(i), problems[j].get_pose(0)
            max_dt = max(max_dt, float(np.linalg.norm(o[:3] - g[:3])))
            max_dr = max(max_dr, float(2.0 * np.arccos(min(1.0, abs(float(np.dot(o[3:], g[3:])))))))
            same_flow = same_flow and (so.num_iterations, so.termination_type, so.termination_reason) == (
                last[j].num_iterations, last[j].termination_type, last[j].termination_reason)
        lbar = pb / max(1, lk)
        base = {"value": n_cpu / (time.perf_counter() - t1), "unit": "matches/s", "cores": 1, "kind": "port",
                "sample": "%d matches of the same workload, oracle -O3 1 thread" % n_cpu}

        def some_matches(state):  # (matches only read the map: the threads share the one built above)
            for (pose_, pts_, _, guess_) in queries[:n_cpu]:
                pr_ = po.Problem()
                i_ = pr_.add_pose(guess_)
                pr_.add_block(pts_, og, scale, i_, multi_res=True)
                pr_.solve()
        base["all_cores"] = all_cores_baseline(args, lambda: None, some_matches, n_cpu, "%d independent matches against the map" % n_cpu)
        parity = {"max_dt_m": max_dt, "max_dr_rad": max_dr, "matches": n_cpu, "tolerance": 1e-4,
                  "same_iterations_and_termination": bool(same_flow),
                  "step": "matches 0..%d of the last timed step (batch of %d, %d-pt scans)" % (n_cpu - 1, B, n_pts)}
        if not (max_dt <= 1e-4 and max_dr <= 1e-4):
            raise SystemExit("bench.py: parity gate failed, GPU and oracle poses of the batch differ: %r" % (parity,))
    n_launch = max(1, prof["residuals"][0])
    avg_ms = prof["residuals"][1] / n_launch
    # launches after a problem's termination move no data for it: scale by the share that evaluated
    sampled_steps = max(1, len([i for i in range(args.steps) if args.prof_every > 0 and i % args.prof_every == 0]))
    evals_per_launch = stats["evals"] / max(1, n_launch)
    bytes_per_launch = n_pts * (12.0 + 32.0 * lbar) * evals_per_launch
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    return {
        "metric": "matches/s (batch of %d independent 100k-pt scan-to-submap matches, 3-res TSDF)" % B,
        "value": args.steps * B / elapsed, "unit": "matches/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "match_batch: %d independent single-pose multi-res LM matches per call "
                               "(%s), %d-pt scans, no insertion" % (
                                   B, "hg_problem_solve_batch" if depth == 1 else
                                   "hg_problem_solve_batch_async + hg_problem_fetch, %d batches in flight: the host builds "
                                   "the next batch while the device solves this one" % depth, n_pts),
                   "batches_in_flight": depth, "map_copies": len(pyramids),
                   "mean_lm_iterations": float(np.mean(stats["its"])), "mean_pose_error_m": float(np.mean(errs))},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "k_tsdf_residuals_single_batch",
                     "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                     "problems_evaluating_per_launch": evals_per_launch, "sampled_steps": sampled_steps},
        "parity": parity, "cpu_baseline": base,
        "gpu_over_cpu": (args.steps * B / elapsed) / base["value"] if base else None,
    }


def oracle_replay_submap(args, stream_base, steps0, scale, total):
    """Parity gate + CPU rate of a batched mapping run: the oracle replays ONE submap (PRNG streams
    `stream_base`): map scans, the warm-up scans where the GPU inserted them, then the first --cpu-scans
    TIMED steps (match + insert). steps0[i] = (pose, iterations, termination type, reason) of the GPU's
    step i of that submap. Returns (cpu_baseline, parity); aborts above the 1e-4 tolerance."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    from hectorgrapher_amd import synth
    og = [po.Grid(r) for r in RESOLUTIONS]
    for pose, pts in make_scans(args.rings, args.cols, 0, args.map_scans, stream_base):
        loc = synth.transform_points(pose, pts)
        for g in og:
            g.insert(pose[:3], loc)
    n_cpu = min(args.steps, args.cpu_scans)
    q0 = make_scans(args.rings, args.cols, args.map_scans, args.warmup + n_cpu, stream_base)
    for i in range(args.warmup):
        at = steps0[i][0]
        loc = synth.transform_points(at, q0[i][1])
        for g in og:
            g.insert(at[:3].astype(np.float32), loc)
    max_dt = max_dr = 0.0
    same_flow = True
    t1 = time.perf_counter()
    for i in range(args.warmup, args.warmup + n_cpu):
        pose, pts = q0[i]
        pr = po.Problem()
        pi = pr.add_pose(synth.pose_mul(pose, synth.perturbation()))
        pr.add_block(pts, og, scale, pi, multi_res=True)
        so = pr.solve()
        o = pr.get_pose(pi)
        g_pose, g_it, g_tt, g_tr = steps0[i]
        max_dt = max(max_dt, float(np.linalg.norm(o[:3] - g_pose[:3])))
        max_dr = max(max_dr, float(2.0 * np.arccos(min(1.0, abs(float(np.dot(o[3:], g_pose[3:])))))))
        same_flow = same_flow and (so.num_iterations, so.termination_type, so.termination_reason) == (g_it, g_tt, g_tr)
        at = o if np.array_equal(o.astype(np.float32), g_pose.astype(np.float32)) else g_pose
        loc = synth.transform_points(at, pts)
        for g in og:
            g.insert(at[:3].astype(np.float32), loc)
    cpu_s = time.perf_counter() - t1
    parity = {"max_dt_m": max_dt, "max_dr_rad": max_dr, "scans": n_cpu, "tolerance": 1e-4,
              "same_iterations_and_termination": bool(same_flow),
              "steps": "one submap of the batch, its first %d timed steps" % n_cpu}
    if not (max_dt <= 1e-4 and max_dr <= 1e-4):
        raise SystemExit("bench.py: parity gate failed, GPU and oracle poses of the replayed submap differ: %r" % (parity,))
    base = {"value": n_cpu / cpu_s, "unit": "scans/s", "cores": 1, "kind": "port",
            "sample": "%d registration steps of one submap (match + 3-level insert), oracle -O3 1 thread" % n_cpu}
    map_scans = make_scans(args.rings, args.cols, 0, args.map_scans, stream_base)

    def some_steps(state):  # the same steps on a submap of the thread's own
        for i in range(args.warmup, args.warmup + n_cpu):
            pose_, pts_ = q0[i]
            pr_ = po.Problem()
            pi_ = pr_.add_pose(synth.pose_mul(pose_, synth.perturbation()))
            pr_.add_block(pts_, state, scale, pi_, multi_res=True)
            pr_.solve()
            at_ = pr_.get_pose(pi_)
            loc_ = synth.transform_points(at_, pts_)
            for g_ in state:
                g_.insert(at_[:3].astype(np.float32), loc_)
    base["all_cores"] = all_cores_baseline(args, lambda: oracle_map(po, synth, map_scans), some_steps, n_cpu,
                                           "%d registration steps of an independent submap" % n_cpu)
    return base, parity


def init_dist(args):
    """(dist or None, rank, local_rank, world): one process per GPU, RCCL (backend "nccl") unless
    HG_DIST_BACKEND=gloo. A single rank started by torch.distributed.run still gets a process group."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if world > 1 or launched or os.environ.get("HG_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        # RCCL pins the calling thread to the GPU's NUMA-local cores at communicator creation, next to its
        # own proxy threads; the thread that feeds the registration chain loses 2 % to that
        os.environ.setdefault("NCCL_IGNORE_CPU_AFFINITY", "1")
        torch.cuda.set_device(local_rank)
        try:
            if os.environ.get("HG_DIST_BACKEND") == "gloo":
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world,
                                        device_id=torch.device("cuda", local_rank))
        except Exception as e:
            if world > 1:
                raise
            sys.stderr.write("process group of one rank not initialised (%r): running without it\n" % (e,))
            dist = None
    return dist, rank, local_rank, world


def process_group_info(dist):
    """What the process group really is -- so that the line itself shows how many ranks the collectives saw and
    which library carried them (backend "nccl" IS RCCL on ROCm)."""
    if dist is None:
        return None
    info = {"backend": dist.get_backend(), "nranks": dist.get_world_size()}
    if info["backend"] == "nccl":
        try:
            import torch
            v = torch.cuda.nccl.version()
            info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
        except Exception as e:  # a version query must never cost the measurement
            info["rccl_version"] = "unknown (%r)" % (e,)
    return info


def run_offline_batch(args, out_fd=None, group=None):
    """BASELINE configs[3], one command: --total-submaps S independent submaps x K scans each farmed to
    the --gpus ranks (hectorgrapher_amd.distributed.map_sharded), every rank registering its S / G
    submaps together per step (hg_register_scan_batch; a single owned submap takes the single chain), then
    the one exchange: gather of all finished TSDF blocks to rank 0 (RCCL over xGMI; one count round, then
    point to point) and its end-to-end check (import into fresh grids, export digests)."""
    import torch
    from hectorgrapher_amd import api, synth
    from hectorgrapher_amd import distributed as hgd
    # group: the (dist, rank, local_rank, world) of a process group the caller already runs in (the configs[3] leg of
    # `bench.py --gpus N`); it is used as it is and left open
    own_group = group is None
    dist, rank, local_rank, world = init_dist(args) if own_group else group
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    S = args.total_submaps
    steps = args.scans_per_submap if args.scans_per_submap > 0 else args.steps
    n_pts = args.rings * args.cols
    scale = 1.0 / np.sqrt(float(n_pts))
    total = args.warmup + steps
    distinct = min(total, 120)  # the bench trajectory folds back after 120 poses: longer runs revisit them

    class Group:
        """The submaps one host thread maps: a context (stream) of its own, one hg_register_scan_batch per step."""

        def open(self, owned):
            self.ctx = api.Context(local_rank)
            self.ins = [api.TSDFRangeDataInserter3D() for _ in RESOLUTIONS]
            self.owned = owned
            self.pyramids, self.queries, self.guesses, self.problems = [], [], [], []
            self.steps0, self.errs, self.its, self.evals = {}, [], [], 0
            for j in owned:
                sb = 100000 * (j + 1)  # PRNG streams of submap j (the same submaps whatever G is)
                grids = [api.HybridGridTSDF(self.ctx, r, max_blocks=args.max_blocks) for r in RESOLUTIONS]
                for pose, pts in make_scans(args.rings, args.cols, 0, args.map_scans, sb):
                    api.insert_pyramid(self.ins, api.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), grids,
                                       pose_tq=pose.astype(np.float32))
                q = make_scans(args.rings, args.cols, args.map_scans, distinct, sb)
                self.pyramids.append(grids)
                self.queries.append([(pose, torch.from_numpy(pts).to(dev)) for pose, pts in q])
                self.guesses.append([synth.pose_mul(pose, synth.perturbation()) for pose, _ in q])
                self.problems.append(api.Problem(self.ctx))
            torch.cuda.synchronize()

        def step(self, i):
            n, k = len(self.problems), i % distinct
            # kernel durations by HIP events on every prof_every-th timed step (an event pair serialises the stream)
            sampling = args.prof_every > 0 and i >= args.warmup and (i - args.warmup) % args.prof_every == 0
            if i == args.warmup:
                self.ctx.prof_reset()
            self.ctx.prof_enable(sampling)
            for j in range(n):
                p = self.problems[j]
                p.reset()
                p.add_pose(self.guesses[j][k])
                p.add_block(self.queries[j][k][1], self.pyramids[j], scale, 0, multi_res=True)
            poses, summ = api.register_scan_batch(self.problems, [0] * n, self.ins,
                                                  [api.RangeData([0, 0, 0], self.queries[j][k][1], width=args.rings) for j in range(n)],
                                                  self.pyramids)
            self.steps0[i] = (poses[0].copy(), summ[0].num_iterations, summ[0].termination_type, summ[0].termination_reason)
            if i >= args.warmup:
                for j in range(n):
                    self.errs.append(float(np.linalg.norm(poses[j][:3] - self.queries[j][k][0][:3])))
                    self.its.append(summ[j].num_iterations)
                    if sampling:
                        self.evals += summ[j].num_cost_evaluations

        def grids(self):
            return [g for pyr in self.pyramids for g in pyr]

    class Engine:
        """A rank's submaps in T groups, one host thread and one context each (--host-threads; default 2 from eight
        owned submaps on): the groups run free of each other, so one group's insertion kernels -- a few long
        per-voxel chains on an otherwise idle chip -- overlap another group's match launches."""

        def open(self, owned):
            T = args.host_threads if args.host_threads > 0 else (2 if len(owned) >= 8 else 1)
            T = max(1, min(T, len(owned)))
            self.groups = []
            for t in range(T):
                g = Group()
                g.open(owned[t::T])
                self.groups.append(g)
            self.ctx = self.groups[0].ctx
            self.host_threads = T

        def run_steps(self, first, last):
            if len(self.groups) == 1:
                for i in range(first, last):
                    self.groups[0].step(i)
                return
            import threading
            errors = []

            def work(g):
                try:
                    for i in range(first, last):
                        g.step(i)
                except BaseException as e:  # surfaced in the caller's thread
                    errors.append(e)

            threads = [threading.Thread(target=work, args=(g,)) for g in self.groups]
            for th in threads:
                th.start()
            for th in threads:
                th.join()
            if errors:
                raise errors[0]

        def sync(self):
            for g in self.groups:
                g.ctx.synchronize()
            torch.cuda.synchronize()

        def grids(self):  # in submap order
            by = {}
            for g in self.groups:
                for j, pyr in zip(g.owned, g.pyramids):
                    by[j] = pyr
            return [gr for j in sorted(by) for gr in by[j]]

        @property
        def errs(self):
            return [e for g in self.groups for e in g.errs]

        @property
        def its(self):
            return [e for g in self.groups for e in g.its]

        @property
        def steps0(self):  # of the rank's first submap
            return self.groups[0].steps0

    def barrier():
        if dist is not None:
            dist.barrier()

    eng = Engine()
    res = hgd.map_sharded(S, rank, world, eng, steps, args.warmup, barrier, dist)
    for g in eng.grids():
        g.status()  # raises on sticky capacity / range flags
    gather_ms = gather_check = None
    if dist is not None:
        eng.sync()
        barrier()
        tg = time.perf_counter()
        gathered = hgd.gather_grids(eng.grids(), dist, rank, world, dev, host=dist.get_backend() != "nccl")
        eng.sync()
        barrier()
        gather_ms = (time.perf_counter() - tg) * 1e3
        gather_check = hgd.verify_gather(api, eng.ctx, eng.grids(), gathered, dist, rank, world)
        del gathered
    # statistics over ALL submaps (every rank holds those of its own)
    acc = torch.tensor([float(np.sum(eng.errs)), float(np.sum(eng.its)), float(len(eng.errs))], dtype=torch.float64)
    if dist is not None:
        if dist.get_backend() == "nccl":
            acc = acc.to(dev)
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    mean_err, mean_its = float(acc[0] / acc[2]), float(acc[1] / acc[2])
    pg_info = process_group_info(dist)
    # the residual family of this rank (dominant kernel of the batched step): launches, time, evaluating problems
    res_launches = res_ms = 0.0
    ins_ms_total, ins_calls = 0.0, 0
    evals = 0
    for g in eng.groups:
        pr_ = g.ctx.prof_read()
        g.ctx.prof_enable(False)
        res_launches += pr_["residuals"][0]
        res_ms += pr_["residuals"][1]
        ins_ms_total += sum(pr_[k][1] for k in ("ray_count", "scan", "ray_expand", "apply"))
        ins_calls += pr_["apply"][0]
        evals += g.evals
    for g in eng.groups:  # the maps of this leg are done with (a caller's group goes on)
        for gr in g.grids():
            gr.close()
        g.ctx.close()
    if rank != 0:
        if dist is not None and own_group:
            dist.destroy_process_group()
        return None
    value = res["scans"] / res["elapsed"]
    base = parity = None
    if not args.no_cpu_baseline and world == 1:
        base, parity = oracle_replay_submap(args, 100000 * (res["owned"][0] + 1), eng.steps0, scale, total)
    per_rank = len(res["owned"])
    roofline = None
    if res_launches > 0:
        lbar = base["mean_levels_probed"] if base and base.get("mean_levels_probed") else 4.0 / 3.0
        avg_ms = res_ms / res_launches
        # a launch of a group evaluates the group's problems that are still running: algorithmic bytes per launch =
        # (evaluating problems per launch) x N x (12 + 32 Lbar)
        per_launch = evals / res_launches
        bytes_per_launch = per_launch * n_pts * (12.0 + 32.0 * lbar)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": None,
                    "kernel": "k_tsdf_residuals_single_batch" if per_rank > 1 else "k_tsdf_residuals_single",
                    "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                    "problems_evaluating_per_launch": per_launch, "mean_levels_probed": lbar,
                    "insert_ms_per_call": ins_ms_total / max(1, ins_calls),
                    "hip_event_sampling": "every %d-th timed step of rank 0's groups" % max(1, args.prof_every)}
    out = {
        "metric": "scans/s (offline batch mapping: %d independent submaps x %d scans, 100k-pt scans, 3-res TSDF registration)" % (S, steps),
        "value": value, "unit": "scans/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": res["elapsed"] / steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "offline_batch: %d submaps x %d scans of %d points farmed to %d rank(s); a step registers one scan "
                               "(multi-res LM match + exact insert x3) of each of a rank's %d submaps together (hg_register_scan_batch, "
                               "%d host thread(s) / context(s) per rank)"
                               % (S, steps, n_pts, world, per_rank, eng.host_threads),
                   "total_submaps": S, "submaps_per_gpu": per_rank, "scans_per_submap": steps,
                   "host_threads_per_rank": eng.host_threads,
                   "parallelism": "submap s on rank s mod %d, no data-path collective; one gather at the end" % world,
                   "mean_lm_iterations": mean_its, "mean_pose_error_m": mean_err,
                   "resident_voxel_gib_per_gpu": per_rank * len(RESOLUTIONS) * (2 * args.max_blocks) * 2048 / 2.0 ** 30,
                   "gather_ms": gather_ms, "gather_check": gather_check, "process_group": pg_info},
        "roofline": roofline, "parity": parity, "cpu_baseline": base,
        "gpu_over_cpu": value / base["value"] if base else None,
    }
    if dist is not None and own_group:
        dist.destroy_process_group()
    return out


def run_register_batch(args):
    """BASELINE configs[3] on one GPU: S independent submaps mapped together in one process. One step
    = one registration (multi-res LM match + exact 3-level insert of a 100k-point scan) for EVERY
    submap through hg_register_scan_batch: the matches share their launches (grid row = submap), the
    insertions run through kernels that take a table of pyramids. --batch-threads T splits the submaps
    over T host threads with a context (HIP stream) each, so that one group's insertion kernels overlap
    another group's match launches. Extra workload, not the headline."""
    import threading
    import torch
    from hectorgrapher_amd import api, synth
    dev = torch.device("cuda", 0)
    n_pts = args.rings * args.cols
    S, T = args.batch_submaps, max(1, min(args.batch_threads, args.batch_submaps))
    total = args.warmup + args.steps
    scale = 1.0 / np.sqrt(float(n_pts))

    class Group:
        def __init__(self, first, count):
            self.ctx = api.Context(0)
            self.ins = [api.TSDFRangeDataInserter3D() for _ in RESOLUTIONS]
            self.pyramids, self.queries, self.guesses, self.problems = [], [], [], []
            self.errs, self.its, self.evals, self.prof = [], [], 0, None
            for j in range(first, first + count):
                sb = 100000 * (j + 1)  # PRNG streams of submap j (as the children of --submaps)
                grids = [api.HybridGridTSDF(self.ctx, r, max_blocks=args.max_blocks) for r in RESOLUTIONS]
                for pose, pts in make_scans(args.rings, args.cols, 0, args.map_scans, sb):
                    api.insert_pyramid(self.ins, api.RangeData([0, 0, 0], torch.from_numpy(pts).to(dev)), grids,
                                       pose_tq=pose.astype(np.float32))
                q = make_scans(args.rings, args.cols, args.map_scans, total, sb)
                self.pyramids.append(grids)
                self.queries.append([(pose, torch.from_numpy(pts).to(dev)) for pose, pts in q])
                self.guesses.append([synth.pose_mul(pose, synth.perturbation()) for pose, _ in q])
                self.problems.append(api.Problem(self.ctx))

        def step(self, i, sample):
            n = len(self.problems)
            if not hasattr(self, "steps0"):
                self.steps0 = {}  # step -> (pose, iterations, termination) of submap 0 of this group
            for j in range(n):
                p = self.problems[j]
                p.reset()
                p.add_pose(self.guesses[j][i])
                p.add_block(self.queries[j][i][1], self.pyramids[j], scale, 0, multi_res=True)
            poses, summ = api.register_scan_batch(self.problems, [0] * n, self.ins,
                                                  [api.RangeData([0, 0, 0], self.queries[j][i][1], width=args.rings) for j in range(n)],
                                                  self.pyramids)
            self.steps0[i] = (poses[0].copy(), summ[0].num_iterations, summ[0].termination_type, summ[0].termination_reason)
            for j in range(n):
                self.errs.append(float(np.linalg.norm(poses[j][:3] - self.queries[j][i][0][:3])))
                self.its.append(summ[j].num_iterations)
                if sample:
                    self.evals += summ[j].num_cost_evaluations

        def run(self, start, done):
            for i in range(args.warmup):
                self.step(i, False)
            self.errs.clear()
            self.its.clear()
            self.ctx.prof_reset()
            self.ctx.synchronize()
            start.wait()
            for i in range(args.warmup, total):
                sampling = args.prof_every > 0 and (i - args.warmup) % args.prof_every == 0
                self.ctx.prof_enable(sampling)
                self.step(i, sampling)
            self.ctx.synchronize()
            done.wait()
            self.prof = self.ctx.prof_read()
            self.ctx.prof_enable(False)

    per = [S // T + (1 if t < S % T else 0) for t in range(T)]
    groups, first = [], 0
    for t in range(T):
        groups.append(Group(first, per[t]))
        first += per[t]
    torch.cuda.synchronize()
    start, done = threading.Barrier(T + 1), threading.Barrier(T + 1)
    threads = [threading.Thread(target=g.run, args=(start, done)) for g in groups]
    for th in threads:
        th.start()
    start.wait()
    t0 = time.perf_counter()
    done.wait()
    elapsed = time.perf_counter() - t0
    for th in threads:
        th.join()
    for g in groups:
        for grids in g.pyramids:
            for gr in grids:
                gr.status()  # raises on sticky capacity / range flags
    prof = {k: tuple(sum(g.prof[k][c] for g in groups) for c in range(3)) for k in groups[0].prof}
    errs = sum((g.errs for g in groups), [])
    its = sum((g.its for g in groups), [])
    evals = sum(g.evals for g in groups)
    n_launch = max(1, prof["residuals"][0])
    avg_ms = prof["residuals"][1] / n_launch
    lbar = 4.0 / 3.0
    bytes_per_launch = n_pts * (12.0 + 32.0 * lbar) * (evals / n_launch)
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    ins_ms = sum(prof[k][1] for k in ("ray_count", "scan", "ray_expand", "apply")) / max(1, prof["apply"][0])
    base = None
    parity = None
    if not args.no_cpu_baseline:
        base, parity = oracle_replay_submap(args, 100000, groups[0].steps0, scale, total)
    resident_gib = S * len(RESOLUTIONS) * (2 * args.max_blocks) * 2048 / 2.0 ** 30
    return {
        "metric": "scans/s (%d independent submaps mapped together on one GPU, 100k-pt scans, 3-res TSDF registration)" % S,
        "value": args.steps * S / elapsed, "unit": "scans/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "register_batch: one registration step (multi-res LM match + exact insert x3 of a %d-pt scan) "
                               "for each of %d independent submaps (hg_register_scan_batch, %d host thread(s) / stream(s))"
                               % (n_pts, S, T),
                   "submaps": S, "host_threads": T, "mean_lm_iterations": float(np.mean(its)),
                   "mean_pose_error_m": float(np.mean(errs)), "insert_ms_per_call": ins_ms,
                   "resident_voxel_gib": resident_gib},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "k_tsdf_residuals_single_batch",
                     "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                     "problems_evaluating_per_launch": evals / n_launch,
                     "per_kernel_ms_total": {k: round(v[1], 4) for k, v in prof.items()},
                     "per_kernel_launches": {k: v[0] for k, v in prof.items()}},
        "parity": parity, "cpu_baseline": base,
        "gpu_over_cpu": (args.steps * S / elapsed) / base["value"] if base else None,
    }


# BASELINE configs[3] bounded for the default line and for the --gpus N line: 8 submaps x 30 scans
OFFLINE_SUBMAPS = 8
OFFLINE_LEG = {"total_submaps": OFFLINE_SUBMAPS, "scans_per_submap": 30, "steps": 30, "warmup": 2, "cpu_scans": 2}


def secondary_workloads(args):
    """Bounded runs of the other workloads inside the default command, so that the driver's record holds
    them too (their wall time is map building and the oracle legs; the timed steps are milliseconds, so they run as many
    steps as the stand-alone workloads' profiles and sample the kernels by HIP events on every fourth or fifth -- an
    event pair serialises the stream, which with half of four steps sampled read 5 % low in round 6's first record): each {value, unit, ms_per_step, frac (algorithmic bytes / HBM peak of its dominant kernel),
    parity_ok (its in-run oracle gate)}. Every run builds its own context and maps and frees them."""
    import copy
    out = {}
    plan = [
        ("match_batch_64", run_match_batch, {"workload": "match_batch", "batch": 64, "steps": 12, "warmup": 2, "prof_every": 4, "cpu_scans": 2}),
        ("register_batch_8", run_register_batch, {"workload": "register_batch", "batch_submaps": 8, "batch_threads": 1, "steps": 12, "warmup": 2, "prof_every": 4, "cpu_scans": 2}),
        ("insert_stream_32", run_insert_stream, {"workload": "insert_stream", "stream_scans": 32, "steps": 10, "warmup": 2, "prof_every": 5, "cpu_scans": 2}),
        # the tolerance mode of the same stream (order-free sums on the bins, one closed-form update per voxel and chunk)
        ("insert_stream_fast", run_insert_stream, {"workload": "insert_stream", "stream_scans": 32, "insert_mode": "fast", "steps": 12,
                                                   "warmup": 2, "prof_every": 4, "cpu_scans": 2}),
        # BASELINE configs[0]: the 10k-point scan into one 0.10 m grid, all points matched and through the C++ builder
        ("c1_10k", run_c1_10k, {"workload": "c1_10k", "steps": 30, "warmup": 3, "cpu_scans": 3}),
        ("window_10", run_window, {"workload": "window", "window": 10, "steps": 20, "warmup": 3, "prof_every": 5}),
        # BASELINE configs[2] with the voxels in HBM: 64 scans over 64 copies of the room (~0.4 GB of voxel
        # blocks touched per call, beyond the 256 MB Infinity Cache); insert_stream_32 above stays in cache
        ("insert_stream_64_hbm", run_insert_stream, {"workload": "insert_stream", "stream_scans": 64, "stream_tiles": 64,
                                                     "steps": 4, "warmup": 1, "prof_every": 2, "cpu_scans": 2}),
        # configs[3] with the reference's real builder: 8 submaps, each a 10-control-point window per step, shared launches
        # (its all-cores CPU leg inside this bounded run: 64 threads -- a window of nine 100k-point blocks per thread on all 256
        # cores of the driver box took 215 s of wall time; `--workload window_batch` on its own uses every core)
        ("window_batch_8", run_window_batch, {"workload": "window_batch", "window": 10, "batch_submaps": 8, "steps": 8, "warmup": 2,
                                              "prof_every": 4, "cpu_threads": 64}),
        # BASELINE configs[3] bounded: 8 submaps x 30 scans on this GPU, then the gather of all finished blocks
        # through a one-rank process group and its import / export-digest check (the full 8 x 500 run is
        # `bench.py --total-submaps 8 --scans-per-submap 500`, profiles/r05_bench_offline8x500.json)
        ("offline8", run_offline_batch, dict(OFFLINE_LEG, _force_dist=True)),
    ]
    for name, fn, kw in plan:
        a = copy.copy(args)
        for k, v in kw.items():
            setattr(a, k, v)
        t0 = time.perf_counter()
        forced = bool(kw.get("_force_dist"))
        if forced and "RANK" in os.environ:
            # this process already belongs to a process group (torch.distributed.run with one rank)
            out[name] = {"skipped": "needs a process group of its own: run bench.py bare"}
            continue
        if forced:
            os.environ["HG_FORCE_DIST"] = "1"
        try:
            r = fn(a)
            par = r.get("parity") or {}
            ok = par.get("bit_exact") if "bit_exact" in par else par.get("tolerance_ok") if "tolerance_ok" in par else (
                par.get("max_dt_m", 1.0) <= 1e-4 and par.get("max_dr_rad", 1.0) <= 1e-4) if par else None
            roof = r.get("roofline") or {}
            out[name] = {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"],
                         "frac": roof.get("frac"), "kernel": roof.get("kernel"), "avg_launch_ms": roof.get("avg_launch_ms"),
                         "parity_ok": None if ok is None else bool(ok), "parity": par or None,
                         "wall_s": round(time.perf_counter() - t0, 2)}
            if r.get("cpu_baseline"):
                out[name]["cpu_baseline"] = r["cpu_baseline"]
                out[name]["gpu_over_cpu"] = r["value"] / r["cpu_baseline"]["value"]
            if r.get("rows"):
                out[name]["rows"] = r["rows"]
            cfg = r.get("config") or {}
            for k in ("gather_ms", "gather_check", "process_group", "voxel_working_set_mib", "room_copies"):
                if cfg.get(k) is not None:
                    out[name][k] = cfg[k]
        except SystemExit as e:  # a failed parity gate of a secondary run is reported, the headline stands
            out[name] = {"error": str(e), "parity_ok": False}
        except Exception as e:
            out[name] = {"error": repr(e)}
        finally:
            if forced:
                os.environ.pop("HG_FORCE_DIST", None)
    return out


def run(args, out_fd=None):
    if args.total_submaps > 0:
        return run_offline_batch(args, out_fd)
    if args.workload == "register_batch":
        return run_register_batch(args)
    if args.workload == "match_batch":
        return run_match_batch(args)
    if args.workload == "insert_stream":
        return run_insert_stream(args)
    if args.workload == "register_filtered":
        return run_register_filtered(args)
    if args.workload == "c1_10k":
        return run_c1_10k(args)
    if args.workload == "window":
        return run_window(args)
    if args.workload == "window_batch":
        return run_window_batch(args)
    import torch
    dist, rank, local_rank, world = init_dist(args)
    from hectorgrapher_amd import api, synth
    from hectorgrapher_amd import distributed as hgd

    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    ctx = api.Context(local_rank)
    # The single-pose solve as ONE persistent launch (hg_ctx_set_option "persistent_solve", off by default in the
    # library: the launch needs the GPU to itself, include/hg_mi355x.h). Here the process owns its GPU: one rank per
    # device, no --submaps siblings; ranks that share a device in the one-GPU tests keep a launch per evaluation.
    persistent = (not args.no_persistent_solve and args.submap_index < 0 and os.environ.get("HG_RANKS_SHARE_GPU") != "1")
    if persistent:
        ctx.set_option("persistent_solve", 1)
    persistent = bool(ctx.get_option("persistent_solve"))
    n_pts = args.rings * args.cols

    # independent submap per rank: rank r uses PRNG streams offset by 1000*r
    sb = 1000 * rank + (100000 * (args.submap_index + 1) if args.submap_index >= 0 else 0)
    map_scans = make_scans(args.rings, args.cols, 0, args.map_scans, sb)
    lead = rank == 0 and world == 1 and args.submap_index < 0  # the one process that reports extras
    host_steps = args.host_steps if lead else 0
    host_warm = 2 if host_steps > 0 else 0  # the first host-memory call creates the copy stream and the staging slots
    query = make_scans(args.rings, args.cols, args.map_scans, args.warmup + args.steps + host_warm + host_steps, sb)

    grids = [api.HybridGridTSDF(ctx, r, max_blocks=args.max_blocks) for r in RESOLUTIONS]
    ins_mode = api._lib.HG_INSERT_FAST if args.insert_mode == "fast" else api._lib.HG_INSERT_EXACT
    inserters = [api.TSDFRangeDataInserter3D(mode=ins_mode) for _ in grids]
    for pose, pts in map_scans:
        d = torch.from_numpy(pts).to(dev)
        torch.cuda.synchronize()
        api.insert_pyramid(inserters, api.RangeData([0, 0, 0], d), grids,
                           pose_tq=pose.astype(np.float32))
    d_scans = [torch.from_numpy(pts).to(dev) for _, pts in query[:args.warmup + args.steps]]
    guesses = [synth.pose_mul(pose, synth.perturbation()) for pose, _ in query]
    torch.cuda.synchronize()
    problem = api.Problem(ctx)
    scale = 1.0 / np.sqrt(float(n_pts))
    stats = {"U": 0, "N_in": 0, "evals": 0}
    errs = []
    sampling = [False]
    gpu_steps = []  # (pose, iterations, termination) of every step from the first warmup step on

    def step(i):
        problem.reset()
        pi = problem.add_pose(guesses[i])
        problem.add_block(d_scans[i], grids, scale, pi, multi_res=True)
        est, summ = api.register_scan(problem, pi, inserters, api.RangeData([0, 0, 0], d_scans[i], width=args.rings), grids)
        errs.append(float(np.linalg.norm(est[:3] - query[i][0][:3])))
        gpu_steps.append((est, summ.num_iterations, summ.termination_type, summ.termination_reason))
        if sampling[0]:
            stats["evals"] += summ.num_cost_evaluations  # launches that evaluated (the rest exit early)

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    def steps(lo, hi, prof_every):
        """Steps [lo, hi) in ONE library call (hg_register_scan_sequence: the loop a C++ host runs per scan;
        the interpreter's per-step work and its GIL hand-overs stay out of the timed region). Kernel
        durations are sampled with HIP events on every prof_every-th step: the residual family each time,
        the insert kernels on every fifth of those (an event pair costs ~8 us of stream serialisation)."""
        scans = [api.RangeData([0, 0, 0], d_scans[i], width=args.rings) for i in range(lo, hi)]
        call = api.register_scan_sequence(problem, inserters, scans, guesses[lo:hi], grids, scale,
                                          multi_res=True, prof_every=prof_every, prepare_only=True)

        def collect(result):
            poses, summ = result[0], result[1]
            for k, i in enumerate(range(lo, hi)):
                errs.append(float(np.linalg.norm(poses[k][:3] - query[i][0][:3])))
                gpu_steps.append((poses[k].copy(), summ[k].num_iterations, summ[k].termination_type,
                                  summ[k].termination_reason))
                if prof_every > 0 and k % prof_every == 0:
                    stats["evals"] += summ[k].num_cost_evaluations  # launches that evaluated (the rest exit early)
        return call, collect

    if os.environ.get("HG_BENCH_PYTHON_LOOP") == "1":
        for i in range(args.warmup):
            step(i)
    else:
        call, collect = steps(0, args.warmup, 0)
        collect(call())
    timed_call, timed_collect = steps(args.warmup, args.warmup + args.steps, args.prof_every)  # marshalled here
    stats = {"U": 0, "N_in": 0, "evals": 0}
    errs.clear()
    ctx.prof_reset()
    barrier()
    if args.submap_index >= 0:  # child of --submaps: all submaps start their timed steps together
        os.write(out_fd, b"READY\n")
        sys.stdin.readline()
    t_start = time.time()
    t0 = time.perf_counter()
    if os.environ.get("HG_BENCH_PYTHON_LOOP") == "1":
        for i in range(args.warmup, args.warmup + args.steps):
            sampling[0] = args.prof_every > 0 and (i - args.warmup) % args.prof_every == 0
            ctx.prof_enable(sampling[0])
            step(i)
        timed_result = None
    else:
        timed_result = timed_call()
    barrier()
    elapsed = time.perf_counter() - t0
    t_end = time.time()
    if timed_result is not None:
        timed_collect(timed_result)
    prof = ctx.prof_read()
    ctx.prof_enable(False)
    # host-inclusive leg (untimed for `value`): the same step with the scans handed over in HOST memory,
    # as INTEGRATION.md binds Insert / Match (sensor::RangeData lives in host vectors). Scan k + 1 travels
    # to the device while step k runs (hg_register_scan_sequence); continues the same trajectory.
    host_inclusive = None
    if host_steps > 0:
        lo = args.warmup + args.steps
        w_scans = [api.RangeData([0, 0, 0], np.ascontiguousarray(query[i][1], np.float32), width=args.rings) for i in range(lo, lo + host_warm)]
        api.register_scan_sequence(problem, inserters, w_scans, guesses[lo:lo + host_warm], grids, scale, multi_res=True)
        lo += host_warm
        h_scans = [api.RangeData([0, 0, 0], np.ascontiguousarray(query[i][1], np.float32), width=args.rings) for i in range(lo, lo + host_steps)]
        h_call = api.register_scan_sequence(problem, inserters, h_scans, guesses[lo:lo + host_steps], grids, scale,
                                            multi_res=True, prof_every=0, prepare_only=True)
        barrier()
        th = time.perf_counter()
        h_res = h_call()
        barrier()
        th = time.perf_counter() - th
        h_err = [float(np.linalg.norm(h_res[0][k][:3] - query[lo + k][0][:3])) for k in range(host_steps)]
        host_inclusive = {"value": host_steps / th, "unit": "scans/s", "steps": host_steps, "ms_per_step": th / host_steps * 1e3,
                          "mean_pose_error_m": float(np.mean(h_err)),
                          "mean_lm_iterations": float(np.mean([s_.num_iterations for s_ in h_res[1][:host_steps]])),
                          "positions": "steps %d..%d of the trajectory, BEHIND the timed ones: other positions, other iteration counts "
                                       "(compare mean_lm_iterations with config.mean_lm_iterations) -- not a like-for-like ratio to `value`"
                                       % (lo, lo + host_steps - 1),
                          "handover": "scans in pageable host memory (HG_HOST), 1.2 MB each; copy of scan k + 1 overlapped with step k"}
    # accounting pass (untimed): N_in and U of one more scan of the same workload
    last = args.warmup + args.steps - 1
    acc = api.insert_pyramid(inserters, api.RangeData([0, 0, 0], d_scans[last], width=args.rings), grids,
                             pose_tq=query[last][0].astype(np.float32))
    stats["U"] = sum(a.num_updates for a in acc)
    stats["N_in"] = sum(a.num_hits for a in acc)

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # one-shot exchange at the end of mapping: gather finished TSDF blocks to rank 0
    gather_ms = None
    gather_check = None
    if dist is not None:
        try:
            barrier()
            tg = time.perf_counter()
            gathered = hgd.gather_grids(grids, dist, rank, world, dev, host=dist.get_backend() != "nccl")
            barrier()
            gather_ms = (time.perf_counter() - tg) * 1e3
            # rank 0 imports every peer's blocks into fresh grids; their export must equal the peer's own
            gather_check = hgd.verify_gather(api, ctx, grids, gathered, dist, rank, world)
            del gathered
        except Exception as e:  # the exchange is reported separately; never lose the timed result
            sys.stderr.write("gather of TSDF blocks failed: %r\n" % (e,))
            gather_ms = None

    # BASELINE configs[3] in the same process group: `bench.py --gpus N` (N > 1) is the command the driver's scaling
    # step runs, so the line carries the strong-scaling job too -- 8 independent submaps farmed to the N ranks, the
    # gather of their finished blocks over RCCL and its import / export check (N = 1: `secondary.offline8` of the
    # default line is the same job in a one-rank group)
    offline = None
    if dist is not None and world > 1 and not args.no_secondary and args.insert_mode == "exact" and OFFLINE_SUBMAPS % world == 0:
        import copy
        a = copy.copy(args)
        for k, v in OFFLINE_LEG.items():
            setattr(a, k, v)
        a.no_cpu_baseline = True
        del problem
        for g in grids:
            g.close()
        ctx.close()
        try:
            offline = run_offline_batch(a, group=(dist, rank, local_rank, world))
        except Exception as e:  # never lose the timed headline over the extra leg
            sys.stderr.write("offline batch leg failed on rank %d: %r\n" % (rank, e))
            offline = {"error": repr(e)} if rank == 0 else None

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return None

    total_scans = args.steps * world
    value = total_scans / elapsed

    base = None
    parity = None
    if not args.no_cpu_baseline and world == 1:  # timed on rank 0 at N = 1 only
        base = cpu_baseline(args, map_scans, query, gpu_steps)
        parity = base["parity"]
        if args.insert_mode == "exact" and not (parity["max_dt_m"] <= 1e-4 and parity["max_dr_rad"] <= 1e-4):
            # no timing counts unless the GPU poses are the CPU poses
            raise SystemExit("bench.py: parity gate failed, GPU and oracle poses differ: %r" % (parity,))

    # ---- roofline of the dominant kernel family (HIP-event time on the ctx stream) ----
    insert_kernels = ["ray_count", "scan", "ray_expand", "sort", "alloc", "apply"]
    t_insert = sum(prof[k][1] for k in insert_kernels)
    t_resid = prof["residuals"][1]
    n_insert_calls = prof["apply"][0]  # one fused call covers all 3 levels
    n_resid = prof["residuals"][0]
    lbar = base["mean_levels_probed"] if base else 1.0
    # insert 12*N_in + 8*U bytes per (scan, level); match N_m*(12 + 32*Lbar) per evaluation
    ins_bytes_per_launch = 12.0 * stats["N_in"] + 8.0 * stats["U"]  # summed over the 3 levels
    # a solve enqueues max_num_iterations + 1 launches; those after termination exit at once and move
    # no data, so the per-launch average is scaled by the share of launches that evaluated
    # (persistent solve: ONE launch per solve that runs all of its evaluations, so the share is the number of
    # evaluations per launch; rocprofv3 then shows k_tsdf_residuals_single_persist<512> with the solve's duration)
    active_share = (stats["evals"] / n_resid if persistent else min(1.0, stats["evals"] / n_resid)) if n_resid else 1.0
    res_bytes_per_launch = n_pts * (12.0 + 32.0 * lbar) * active_share
    fam = {
        "insert(expand+sort+alloc+apply, 3 levels fused)": (t_insert / max(1, n_insert_calls), ins_bytes_per_launch, t_insert),
        "k_tsdf_residuals": (t_resid / max(1, n_resid), res_bytes_per_launch, t_resid),  # rocprof: k_tsdf_residuals_single<512> / _single_persist<512>
    }
    dom = max(fam, key=lambda k: fam[k][2])
    avg_ms, bytes_per, _ = fam[dom]
    # HBM traffic of the dominant kernel: PMC counters cannot be collected by the timed run itself (separate
    # rocprofv3 --pmc passes, MI355X guide), so the figure comes from the committed passes of this same
    # command (scripts/r06_profile.sh -> profiles/r06_bench_pmc_traffic.json) and is labelled as such
    traffic = None
    traffic_source = None
    build_digest = api._lib.source_digest()
    try:
        with open(os.path.join(ROOT, "profiles", "r06_bench_pmc_traffic.json")) as f:
            doc = json.load(f)
        pmc = doc["kernels"]
        if doc.get("csrc_sha16") != build_digest:
            # the committed passes were taken on other sources: their bytes are not this build's
            traffic_source = ("none: profiles/r06_bench_pmc_traffic.json was collected on sources %s, this build is %s "
                              "(re-run scripts/r06_profile.sh)" % (doc.get("csrc_sha16"), build_digest))
        else:
            if dom == "k_tsdf_residuals":
                # the single-pose registration step runs the k_tsdf_residuals_single<512> instantiation
                key = [k for k in pmc if "k_tsdf_residuals" in k]
                traffic = max(pmc[k]["traffic_bytes"] for k in key)
            else:
                traffic = sum(pmc[k]["traffic_bytes"] for k in pmc if k.startswith("hg::k_bin_"))
            traffic_source = ("from_profile: profiles/r06_bench_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                              "this command on the same sources, scripts/r06_profile.sh), not collected by this run")
    except Exception as e:
        traffic = None
        traffic_source = "none: %r" % (e,)
    achieved = bytes_per / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    families = {}
    for name, (ms, nbytes, _) in fam.items():
        gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        families[name] = {"avg_launch_ms": ms, "algorithmic_bytes_per_launch": nbytes, "achieved": gbs,
                          "frac": gbs / HBM_PEAK_GBS}
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source, "kernel": dom,
                "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": bytes_per,
                "per_kernel_ms_total": {k: round(v[1], 4) for k, v in prof.items()},
                "per_kernel_launches": {k: v[0] for k, v in prof.items()},
                "families": families,
                "hip_event_sampling": "residual family on every %d-th of the %d timed steps, insert kernels on every %d-th; "
                                      "an event pair serialises the stream (~8 us), so sampled steps run longer than the others "
                                      "and per_kernel_ms_total does not add up to ms_per_step (rocprofv3 trace: profiles/r06_trace_gaps.txt)"
                                      % (max(1, args.prof_every), args.steps, 5 * max(1, args.prof_every)),
                "build": {"version": api._lib.load().hg_version().decode(), "csrc_sha16": build_digest},
                "residual_launches_evaluating": active_share,
                "solve_form": ("persistent: one k_tsdf_residuals_single_persist<512> launch per solve, %.1f evaluations each"
                               % active_share) if persistent else "one k_tsdf_residuals_single<512> launch per evaluation"}

    out = {
        "metric": "scans/s (100k-pt scan, 3-res TSDF registration)" + (
            "" if args.insert_mode == "exact" else " [tolerance insert mode: NOT the headline configuration]"),
        "value": value, "unit": "scans/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "register %d-pt synthetic scan (%d rings x %d cols) into 3-res TSDF "
                               "0.05/0.10/0.20 m: multi-res LM match (<=12 it) + exact insert x3"
                               % (n_pts, args.rings, args.cols),
                   "points_per_scan": n_pts, "map_scans": args.map_scans,
                   "parallelism": "independent submap per GPU x%d" % world,
                   "insert_mode": args.insert_mode, "mean_pose_error_m": float(np.mean(errs)),
                   "mean_lm_iterations": float(np.mean([g_[1] for g_ in gpu_steps[args.warmup:args.warmup + args.steps]])),
                   "resident_voxel_gib": len(RESOLUTIONS) * (2 * args.max_blocks) * 2048 / 2.0 ** 30,
                   "gather_ms": gather_ms, "gather_check": gather_check},
        "roofline": roofline,
    }
    if host_inclusive:
        out["host_inclusive"] = host_inclusive
    out["config"]["process_group"] = process_group_info(dist)
    if offline is not None:
        cfg = offline.get("config") or {}
        out["secondary"] = {"offline8": {k: offline.get(k) for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "error")
                                         if offline.get(k) is not None}}
        for k in ("gather_ms", "gather_check", "process_group", "submaps_per_gpu", "total_submaps", "mean_pose_error_m"):
            if cfg.get(k) is not None:
                out["secondary"]["offline8"][k] = cfg[k]
    if lead and not args.no_secondary and args.insert_mode == "exact":
        # free this run's maps first: the secondary runs build their own
        del problem
        for g in grids:
            g.close()
        ctx.close()
        out["secondary"] = secondary_workloads(args)
    if base:
        out["parity"] = parity
        out["cpu_baseline"] = {k: base[k] for k in ("value", "unit", "cores", "kind", "sample")}
        out["cpu_baseline"]["cores_available"] = os.cpu_count()
        out["gpu_over_cpu"] = value / world / base["value"]
    if args.submap_index >= 0:
        out["t_start"], out["t_end"] = t_start, t_end
    if dist is not None:
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
