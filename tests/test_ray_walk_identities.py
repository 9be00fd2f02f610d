"""Arithmetic identities behind the integer front end of the binned insert (hg_insert.hip: RayWalk,
ray_block_runs), checked exhaustively on the CPU. The reference walks a ray with
`begin + round(delta.cast<float>() * float(position) / float(num_samples))`
(tsdf_range_data_inserter_3d.cc:318-320, Eigen array round = half away from zero); the binned path only
sees rays of at most 7 steps (8 record slots per return), for which the kernels use exact integers."""
import numpy as np


def _round_half_away(x):
    return np.sign(x) * np.floor(np.abs(x) + 0.5)


def _reference_offset(d, pos, n):
    q = np.float32(np.float32(d) * np.float32(pos)) / np.float32(n)  # fp32 product, IEEE fp32 division
    return int(_round_half_away(np.float64(q)))


def test_sample_offset_is_an_integer_quotient():
    for n in range(1, 8):
        for d in range(-n, n + 1):
            for pos in range(0, n + 1):
                want = _reference_offset(d, pos, n)
                got = (1 if d >= 0 else -1) * ((2 * abs(d) * pos + n) // (2 * n))
                assert got == want, (n, d, pos, got, want)


def test_digital_differential_follows_the_quotient():
    for n in range(1, 8):
        for d in range(-n, n + 1):
            c, rem, a, s, n2 = 0, n, 2 * abs(d), (-1 if d < 0 else 1), 2 * n
            for pos in range(0, n + 1):
                assert c == _reference_offset(d, pos, n), (n, d, pos)
                rem += a
                if rem >= n2:  # at most one carry: rem < 2 n before, a <= 2 n
                    rem -= n2
                    c += s
                assert rem < n2


def _magic(ad, ulp_shift=0):
    """floor(32768 / ad) + 1 from a reciprocal that may be an ulp off (v_rcp_f32 is not correctly rounded)."""
    r = np.float32(1.0) / np.float32(ad)
    if ulp_shift:
        r = np.nextafter(r, np.float32(2.0 if ulp_shift > 0 else 0.0), dtype=np.float32)
    return int(np.float32(32768.0) * r) + 1


def test_block_crossing_position():
    """A coordinate that starts u cells into its 8-cell block and moves d cells over n steps leaves the block
    at the first position whose offset reaches t; the kernel computes ceil(n (2 t - 1) / (2 |d|)) with a
    16-bit reciprocal."""
    for n in range(1, 8):
        for d in range(-n, n + 1):
            if d == 0:
                continue
            ad = abs(d)
            for u in range(8):
                t = 8 - u if d > 0 else u + 1
                # by walking
                first = None
                for pos in range(0, n + 1):
                    cell = u + _reference_offset(d, pos, n)
                    if cell // 8 != 0:  # floor division: negative cells lie in the block before
                        first = pos
                        break
                crosses = ad >= t
                assert (first is not None) == crosses, (n, d, u)
                if not crosses:
                    continue
                y = 2 * ad
                x = n * (2 * t - 1) + y - 1
                assert x // y == first
                for shift in (-1, 0, 1):
                    m = _magic(ad, shift)
                    assert (x * m) >> 16 == first, (n, d, u, shift)


def test_a_short_ray_crosses_each_axis_at_most_once():
    for n in range(1, 8):
        for d in range(-n, n + 1):
            for u in range(8):
                blocks = [(u + _reference_offset(d, pos, n)) // 8 for pos in range(0, n + 1)]
                changes = sum(1 for a, b in zip(blocks, blocks[1:]) if a != b)
                assert changes <= 1
