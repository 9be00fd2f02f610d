"""Deterministic synthetic scans for tests and bench.

Scene and sensor follow the reference's evaluation harness
(ref: cartographer/evaluation/trajectory_builder_evaluation.cc:143-148,
evaluation/simulation/scene.cc:14-76, evaluation/simulation/range_sensor.cc:27-41):
an axis-aligned room with three spheres, a multi-ring spinning lidar with
elevation in +-15 deg, first hit wins. Generalised to R rings x C columns,
stored azimuth-major with width = R. Range noise N(0, sigma) along the ray from
a counter-based PRNG (Philox) + Box-Muller, seed 42 as the reference tests use
(ref: mapping/internal/3d/local_trajectory_builder_3d_test.cc:118).
"""
import numpy as np

ROOM_MIN = np.array([-5.0, -5.0, -1.0])
ROOM_SIZE = np.array([10.0, 20.0, 10.0])
SPHERES = [(np.array([5.0, 5.0, 0.0]), 3.0), (np.array([-3.0, 2.0, -1.0]), 1.0),
           (np.array([-2.0, -2.0, 2.0]), 2.0)]
RAY_LENGTH = 100.0


def quat_from_axis_angle(axis, angle):
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    s = np.sin(0.5 * angle)
    return np.array([np.cos(0.5 * angle), axis[0] * s, axis[1] * s, axis[2] * s])


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw])


def quat_rotate(q, v):
    """Rotate (..., 3) vectors by unit quaternion q = (w, x, y, z)."""
    v = np.asarray(v)
    u = np.asarray(q[1:], v.dtype)
    w = v.dtype.type(q[0])
    uv = np.cross(u, v)
    uv = uv + uv
    return v + w * uv + np.cross(u, uv)


def pose_mul(a, b):
    """a * b for poses (t xyz, q wxyz)."""
    t = quat_rotate(a[3:], np.asarray(b[:3], np.float64)) + a[:3]
    q = quat_mul(a[3:], b[3:])
    return np.concatenate([t, q / np.linalg.norm(q)])


def pose_inverse(a):
    qc = np.array([a[3], -a[4], -a[5], -a[6]])
    t = -quat_rotate(qc, np.asarray(a[:3], np.float64))
    return np.concatenate([t, qc])


def pose_k(k):
    """Map-building pose P_k = translate(0.05k, 0.02k, 0) * Rz(0.01k)."""
    return np.concatenate([[0.05 * k, 0.02 * k, 0.0], quat_from_axis_angle([0, 0, 1], 0.01 * k)])


def perturbation():
    """Initial-guess offset: 5/-3/2 cm, 0.01 rad about (1,-1,2)/sqrt(6)."""
    return np.concatenate([[0.05, -0.03, 0.02], quat_from_axis_angle([1, -1, 2], 0.01)])


def transform_points(tq, pts, dtype=np.float32):
    """tq * pts evaluated in `dtype` with the Eigen quaternion formula."""
    pts = np.asarray(pts, dtype)
    q = np.asarray(tq[3:], dtype)
    t = np.asarray(tq[:3], dtype)
    return (quat_rotate(q, pts) + t).astype(dtype)


def directions(rings, cols):
    """Unit ray directions in the sensor frame, azimuth-major (width = rings)."""
    az = -np.pi + 2.0 * np.pi * np.arange(cols) / cols
    if rings > 1:
        el = np.deg2rad(-15.0 + 30.0 * np.arange(rings) / (rings - 1))
    else:
        el = np.zeros(1)
    azg, elg = np.meshgrid(az, el, indexing="ij")  # (cols, rings)
    d = np.stack([np.cos(elg) * np.cos(azg), np.cos(elg) * np.sin(azg), np.sin(elg)], -1)
    return d.reshape(-1, 3)


def raycast(origin, dirs):
    """First-hit range along unit `dirs` from `origin` (scene.cc:14-76)."""
    to = dirs * RAY_LENGTH
    ratio = np.ones(len(dirs))
    lo = ROOM_MIN - origin
    hi = ROOM_MIN + ROOM_SIZE - origin
    with np.errstate(divide="ignore", invalid="ignore"):
        for a in range(3):
            pos = to[:, a] > 0
            neg = to[:, a] < 0
            ratio = np.where(pos, np.minimum(ratio, hi[a] / to[:, a]), ratio)
            ratio = np.where(neg, np.minimum(ratio, lo[a] / to[:, a]), ratio)
    aa = np.einsum("ij,ij->i", to, to)
    for c, r in SPHERES:
        oc = origin - c
        beta = to @ oc
        cc = oc @ oc - r * r
        disc = beta * beta - aa * cc
        ok = disc >= 0
        sol = (-beta - np.sqrt(np.where(ok, disc, 0.0))) / aa
        ok &= sol >= 0
        ratio = np.where(ok, np.minimum(ratio, sol), ratio)
    return ratio * RAY_LENGTH


def _normal(n, seed, stream):
    g = np.random.Generator(np.random.Philox(key=seed, counter=[0, 0, 0, stream]))
    u1 = 1.0 - g.random(n)
    u2 = g.random(n)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def generate_scan(pose_tq, rings, cols, seed=42, stream=0, noise_sigma=0.01):
    """Returns (N, 3) float32 points in the sensor frame for a sensor at `pose_tq`."""
    d_s = directions(rings, cols)
    d_w = quat_rotate(np.asarray(pose_tq[3:], np.float64), d_s)
    rng = raycast(np.asarray(pose_tq[:3], np.float64), d_w)
    if noise_sigma > 0:
        rng = rng + noise_sigma * _normal(len(rng), seed, stream)
    return (d_s * rng[:, None]).astype(np.float32)


def raycast_from(origins, dirs):
    """raycast with an origin per ray: (N, 3) origins, (N, 3) unit directions (a sensor that moves while it sweeps)."""
    to = dirs * RAY_LENGTH
    ratio = np.ones(len(dirs))
    lo = ROOM_MIN - origins
    hi = ROOM_MIN + ROOM_SIZE - origins
    with np.errstate(divide="ignore", invalid="ignore"):
        for a in range(3):
            pos = to[:, a] > 0
            neg = to[:, a] < 0
            ratio = np.where(pos, np.minimum(ratio, hi[:, a] / to[:, a]), ratio)
            ratio = np.where(neg, np.minimum(ratio, lo[:, a] / to[:, a]), ratio)
    aa = np.einsum("ij,ij->i", to, to)
    for c, r in SPHERES:
        oc = origins - c
        beta = np.einsum("ij,ij->i", to, oc)
        cc = np.einsum("ij,ij->i", oc, oc) - r * r
        disc = beta * beta - aa * cc
        ok = disc >= 0
        sol = (-beta - np.sqrt(np.where(ok, disc, 0.0))) / aa
        ok &= sol >= 0
        ratio = np.where(ok, np.minimum(ratio, sol), ratio)
    return ratio * RAY_LENGTH


def generate_swept_scan(pose_of_column, rings, cols, seed=42, stream=0, noise_sigma=0.01):
    """A scan taken by a MOVING sensor: column c is measured from pose_of_column(c) (t xyz, q wxyz) and returned in
    that pose's sensor frame, as a spinning lidar delivers it -- what per-point unwarping exists for. (N, 3) float32,
    azimuth-major like generate_scan."""
    d_s = directions(rings, cols)
    poses = np.array([pose_of_column(c) for c in range(cols)], np.float64)   # (cols, 7)
    d_w = np.empty_like(d_s)
    origins = np.repeat(poses[:, :3], rings, axis=0)
    for c in range(cols):
        d_w[c * rings:(c + 1) * rings] = quat_rotate(poses[c, 3:], d_s[c * rings:(c + 1) * rings])
    rng = raycast_from(origins, d_w)
    if noise_sigma > 0:
        rng = rng + noise_sigma * _normal(len(rng), seed, stream)
    return (d_s * rng[:, None]).astype(np.float32)


CONFIGS = {"10k": (16, 625), "100k": (50, 2000), "1k": (8, 128), "4k": (16, 256)}


def scan_config(name):
    return CONFIGS[name]
