"""The single-pose solve as ONE persistent launch (hg_ctx_set_option "persistent_solve", csrc/hg_match.hip
k_tsdf_residuals_single_persist) and the gating around it. Reference semantics unchanged: the Ceres-1.13 solve of
optimizing_local_trajectory_builder.cc:1238-1291; the form only changes how the evaluations are launched."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from hectorgrapher_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES = (0.05, 0.10, 0.20)


def test_persistent_solve_equals_a_launch_per_evaluation_bit_for_bit():
    """Same arithmetic in the same order: poses, iteration counts, termination and the voxels inserted at the solved
    poses are IDENTICAL in both forms (a process of its own: the library takes the persistent form only while the
    process holds one context)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "persist_check.py"), "12", "32", "1000"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["persistent_option"] == 1
    assert len(d["persistent"]) == 12
    for a, b in zip(d["per_evaluation"], d["persistent"]):
        assert a == b, (a, b)                       # floats compared as Python floats: every bit
        assert a[2] in (0, 1) and a[3] != 7          # converged or out of iterations, never the time-out failure
    assert d["voxels_equal"]


def test_two_contexts_in_two_threads_solve_concurrently(hg, ctx):
    """Two contexts of one process, a host thread each, both ASKING for the persistent form: the library sees more
    than one live context and keeps a launch per evaluation (two persistent launches could each hold half of the CUs
    and wait for the other half for ever). 200 registration steps per thread: no failed step, no NaN, and every pose
    equal to the same trajectory run alone."""
    steps, rings, cols = 200, 16, 400

    def trajectory(c, seed, out):
        grids = [hg.HybridGridTSDF(c, r, max_blocks=1 << 15) for r in RES]
        ins = [hg.TSDFRangeDataInserter3D() for _ in RES]
        for k in range(3):
            pose = synth.pose_k(k)
            hg.insert_pyramid(ins, hg.RangeData([0, 0, 0], synth.generate_scan(pose, rings, cols, stream=seed + k)), grids,
                              pose_tq=pose.astype(np.float32))
        p = hg.Problem(c)
        for k in range(3, 3 + steps):
            pose = synth.pose_k(3 + (k % 40))       # a loop over 40 poses of the room
            pts = scans[(seed, k)]
            p.reset()
            i = p.add_pose(synth.pose_mul(pose, synth.perturbation()))
            p.add_block(pts, grids, 1.0 / np.sqrt(len(pts)), i, multi_res=True)
            est, summ = hg.register_scan(p, i, ins, hg.RangeData([0, 0, 0], pts, width=rings), grids)
            out.append((est.copy(), summ.num_iterations, summ.termination_type, summ.termination_reason))
        p.close()
        for g in grids:
            g.close()

    scans = {}
    for seed in (0, 5000):
        for k in range(3, 3 + steps):
            scans[(seed, k)] = synth.generate_scan(synth.pose_k(3 + (k % 40)), rings, cols, stream=seed + k)
    alone = {}
    for seed in (0, 5000):
        alone[seed] = []
        trajectory(ctx, seed, alone[seed])
    c1, c2 = hg.Context(0), hg.Context(0)
    try:
        for c in (c1, c2):
            c.set_option("persistent_solve", 1)
        both = {0: [], 5000: []}
        errors = []

        def run(c, seed):
            try:
                trajectory(c, seed, both[seed])
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        threads = [threading.Thread(target=run, args=(c1, 0)), threading.Thread(target=run, args=(c2, 5000))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        for seed in (0, 5000):
            assert len(both[seed]) == steps
            for (pa, ia, ta, ra), (pb, ib, tb, rb) in zip(alone[seed], both[seed]):
                assert np.all(np.isfinite(pb)) and rb != 7 and tb != 2
                assert np.array_equal(pa, pb) and (ia, ta, ra) == (ib, tb, rb)
    finally:
        c1.close()
        c2.close()


def test_options_are_per_context_and_unknown_keys_are_refused(hg, ctx):
    assert ctx.get_option("persistent_solve") in (0, 1)
    with ctx.option("stream_group", 4):
        assert ctx.get_option("stream_group") == 4
    assert ctx.get_option("stream_group") == 32
    with pytest.raises(hg.HgError):
        ctx.set_option("no_such_switch", 1)
