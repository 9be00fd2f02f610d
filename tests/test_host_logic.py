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
