"""Host-side mirrors of the reference's orchestration logic (no GPU needed)."""
import numpy as np

from hectorgrapher_amd import api


def test_time_conversions_match_common_time():
    # common::FromSeconds truncates toward zero to 100 ns ticks (common/time.cc:30-33)
    assert api.from_seconds(0.1) == 1000000
    assert api.from_seconds(1e-7 * 2.9) == 2
    assert api.from_seconds(-1e-7 * 2.9) == -2
    assert api.to_seconds(12345678) == 1.2345678


def test_per_point_subdivisions_bracketing():
    """AddPerPointMatchingResiduals (oltb.cc:521-565): subdivisions of num_points_per_subdivision
    returns timed at the mean of their first and last return; outside (front, back) omitted;
    ratio = (t - prev) / (next - prev) clamped to [0, 1]."""
    t0 = 636_000_000_000_000_000
    control = [t0, t0 + api.from_seconds(0.1), t0 + api.from_seconds(0.25)]
    times = np.linspace(-0.05, 0.30, 36).astype(np.float32)   # relative to the cloud time
    cloud_time = t0 + api.from_seconds(0.01)
    subs = api.per_point_subdivisions(times, cloud_time, control, 4)
    # brute force
    expect = []
    for s in range(0, 36, 4):
        e = min(s + 3, 35)
        t = cloud_time + api.from_seconds(0.5 * float(np.float32(times[s]) + np.float32(times[e])))  # float add
        if not (control[0] < t < control[-1]):
            continue
        nxt = next(i for i in range(1, 3) if control[i] > t)
        ratio = api.to_seconds(t - control[nxt - 1]) / api.to_seconds(control[nxt] - control[nxt - 1])
        expect.append((s, e + 1, nxt - 1, nxt, min(max(ratio, 0.0), 1.0)))
    assert subs == expect
    assert subs[0][0] > 0 and subs[-1][1] < 36          # both ends fall outside the window
    assert {s[2] for s in subs} == {0, 1}               # both control-point pairs are used
    for s in subs:
        assert 0.0 <= s[4] <= 1.0
    # a ragged tail: the last subdivision is shorter
    subs7 = api.per_point_subdivisions(times[:10], cloud_time + api.from_seconds(0.08), control, 7)
    assert [(a, b) for a, b, *_ in subs7] == [(0, 7), (7, 10)]


def test_subdivision_time_adds_the_two_point_times_in_float():
    """oltb.cc:537-542: `0.5 * (a.time + b.time)` with float times is a FLOAT sum promoted by the
    double 0.5. For these two times the float sum rounds, the double sum does not, and FromSeconds
    lands one 100 ns tick apart: the float form is the reference's."""
    a, b = np.float32(0.13493788), np.float32(0.11411292)
    as_float = api.from_seconds(0.5 * float(np.float32(a + b)))
    as_double = api.from_seconds(0.5 * (float(a) + float(b)))
    assert as_float != as_double          # the case distinguishes the two forms
    t0 = 1_000_000_000
    control = [t0 - api.from_seconds(1.0), t0 + api.from_seconds(1.0)]
    subs = api.per_point_subdivisions(np.array([a, b], np.float32), t0, control, 2)
    want = api.to_seconds(t0 + as_float - control[0]) / api.to_seconds(control[1] - control[0])
    assert subs == [(0, 2, 0, 1, want)]


def test_cpp_adapter_host_functions_against_oracle_and_python_statement():
    """The host-only pieces of cpp/hg_adapter.h that the reference-shaped OptimizingLocalTrajectoryBuilder stands on
    -- InterpolateTransform (timestamped_transform.h:41-65, Eigen slerp), TransformInterpolationBuffer::Lookup /
    LookupUntilDelta (transform_interpolation_buffer.cc:55-125: the ADAPTIVE control-point sampling) and the
    pre-integrated IMU rotation (imu_integration.h:99-131) -- against the oracle's InterpolateTransform (pinned by the
    reference's own 21-factor test) and the Python statement of the same functions (tests/oltb_replay.py). No GPU."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle"))
    sys.path.insert(0, os.path.join(root, "tests"))
    import pyoracle as po
    import oltb_replay as rp
    cpp = os.path.join(root, "hectorgrapher_amd", "cpp")
    exe = os.path.join(cpp, "example_host_logic")
    src = os.path.join(cpp, "example_host_logic.cc")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(cpp, "hg_adapter.h"))):
        subprocess.check_call(["g++", "-std=c++11", "-O2", src, "-L" + os.path.join(root, "hectorgrapher_amd"), "-lhg_mi355x",
                               "-Wl,-rpath," + os.path.join(root, "hectorgrapher_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    odom, imu, lookups, untils, deltas = [], [], [], [], []
    for line in out.stdout.splitlines():
        f = line.split()
        if f[0] == "odom":
            odom.append((int(f[1]), np.array([float(x) for x in f[2:]])))
        elif f[0] == "imu":
            imu.append((int(f[1]), np.array([float(x) for x in f[2:]])))
        elif f[0] == "lookup":
            lookups.append((int(f[1]), np.array([float(x) for x in f[2:]])))
        elif f[0] == "until":
            untils.append((int(f[1]), [float(x) for x in f[2:5]], int(f[6]), [float(x) for x in f[7:10]]))
        elif f[0] == "imu_delta":
            deltas.append((int(f[1]), int(f[2]), np.array([float(x) for x in f[3:]])))
    assert len(odom) == 40 and len(imu) == 80 and len(lookups) == 5 and len(untils) == 12 and len(deltas) == 4
    buf = rp.InterpolationBuffer(po, odom)
    for t, pose in lookups:
        np.testing.assert_allclose(pose, buf.lookup(t), rtol=0, atol=1e-15)
    assert any(t not in buf.times for t, _ in lookups) and any(t in buf.times for t, _ in lookups)
    capped = 0
    for t, lim, cand, ratios in untils:
        c, tr, rr, dr = buf.lookup_until_delta(t, lim[0], lim[1], lim[2])
        assert c == cand, (t, lim, c, cand)              # integer ticks: exactly
        np.testing.assert_allclose(ratios, [tr, rr, dr], rtol=1e-12, atol=1e-15)
        capped += max(ratios) > 1.0 - 1e-12
    assert capped >= 9                                   # translation, rotation and time each decide some of them
    for a, b, q in deltas:
        np.testing.assert_allclose(q, rp.imu_delta_rotation(imu, a, b), rtol=0, atol=1e-15)


def _host_logic_lines():
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cpp = os.path.join(root, "hectorgrapher_amd", "cpp")
    exe = os.path.join(cpp, "example_host_logic")
    src = os.path.join(cpp, "example_host_logic.cc")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(cpp, "hg_adapter.h"))):
        subprocess.check_call(["g++", "-std=c++11", "-O2", src, "-L" + os.path.join(root, "hectorgrapher_amd"), "-lhg_mi355x",
                               "-Wl,-rpath," + os.path.join(root, "hectorgrapher_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    return [line.split() for line in out.stdout.splitlines()]


def _angular_distance_to_identity(q):
    return 2.0 * np.arccos(min(1.0, abs(q[0]) / np.linalg.norm(q)))


def test_cpp_transform_interpolation_buffer_reference_kat():
    """The adapter's transform::TransformInterpolationBuffer through the reference's own test vectors,
    transform/transform_interpolation_buffer_test.cc:29-74 (testHas, testLookup, testLookupSingleTransform; IsNearly 1e-6)."""
    lines = {f[0]: f[1:] for f in _host_logic_lines() if f[0].startswith("kat_")}
    # testHas: empty buffer, one entry at 50, then a second at 100; earliest / latest time
    assert [int(x) for x in lines["kat_has"]] == [0, 0, 1, 0, 0, 1, 1, 1, 0, 50, 100]
    # testLookup: halfway between identity and T(10, 10, 10) * Rz(2): T(5, 5, 5) * Rz(1)
    p = np.array([float(x) for x in lines["kat_lookup"]])
    np.testing.assert_allclose(p[:3], [5.0, 5.0, 5.0], rtol=0, atol=1e-6)
    want = np.array([np.cos(0.5), 0.0, 0.0, np.sin(0.5)])
    assert np.abs(p[3:] - want).max() < 1e-6 or np.abs(p[3:] + want).max() < 1e-6
    # testLookupSingleTransform
    np.testing.assert_allclose([float(x) for x in lines["kat_single"]], [0, 0, 0, 1, 0, 0, 0], rtol=0, atol=1e-6)


def test_cpp_imu_euler_integration_reference_kat():
    """The adapter's mapping::IntegrateImuWithTranslationEuler (which its IMU pre-integration blocks stand on) through the
    reference's own known answers, mapping/internal/3d/imu_integration_test.cc:30-120: zero integration over 100 s,
    and constant acceleration (0, 0, 9.80665) at 100 Hz from 0 to every sample time: |dv - t g| < 1e-8,
    |dp - 0.5 t^2 g| < 10 t, rotation = identity to 1e-8 (kPrecision)."""
    rows = _host_logic_lines()
    zero = [np.array([float(x) for x in f[1:]]) for f in rows if f[0] == "kat_imu_zero"]
    assert len(zero) == 1
    assert np.linalg.norm(zero[0][0:3]) < 1e-8 and np.linalg.norm(zero[0][3:6]) < 1e-8
    assert _angular_distance_to_identity(zero[0][6:10]) < 1e-8
    const = [(int(f[1]), np.array([float(x) for x in f[2:]])) for f in rows if f[0] == "kat_imu_const"]
    assert len(const) == 1000
    g = np.array([0.0, 0.0, 9.80665])
    for ticks, r in const:
        t = api.to_seconds(ticks)
        assert np.linalg.norm(r[0:3] - t * g) < 1e-8
        assert np.linalg.norm(r[3:6] - 0.5 * t * t * g) < t * 10
        assert _angular_distance_to_identity(r[6:10]) < 1e-8
    assert const[-1][0] == 1000 * api.from_seconds(0.01)
