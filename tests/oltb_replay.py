"""An independent Python statement of OptimizingLocalTrajectoryBuilder (reference:
cartographer/mapping/internal/3d/optimizing_local_trajectory_builder.cc, line numbers below) over the CPU oracle
(pyoracle: grids, TSDF cost functions, Ceres-style solve, voxel filters, InterpolateTransform). Test infrastructure:
tests/test_gpu_cpp_oltb.py feeds it the sensor messages cpp/example_oltb.cc dumped and compares every control point
of every solve with what the C++ adapter (cpp/hg_adapter.h, device solves) reported.

Times are integers (common::Time ticks of 100 ns), poses (t xyz, q wxyz) float64 arrays."""
import bisect
import math
import re
import struct
import sys

import numpy as np

TICKS = 1e7


def from_seconds(s):  # common::FromSeconds: duration_cast truncates toward zero (common/time.cc:30-33)
    return int(s * TICKS)


def to_seconds(d):  # common::ToSeconds (:35-38)
    return d / TICKS


def quat_mul(a, b):
    w, x, y, z = a
    return np.array([w * b[0] - x * b[1] - y * b[2] - z * b[3], w * b[1] + x * b[0] + y * b[3] - z * b[2],
                     w * b[2] + y * b[0] + z * b[1] - x * b[3], w * b[3] + z * b[0] + x * b[2] - y * b[1]])


def quat_rotate(q, v):  # Eigen QuaternionBase::_transformVector
    w, x, y, z = q
    ux, uy, uz = y * v[2] - z * v[1], z * v[0] - x * v[2], x * v[1] - y * v[0]
    tx, ty, tz = ux + ux, uy + uy, uz + uz
    return np.array([v[0] + w * tx + (y * tz - z * ty), v[1] + w * ty + (z * tx - x * tz), v[2] + w * tz + (x * ty - y * tx)])


def pose_mul(a, b):  # Rigid3d operator* (rigid_transform.h:184-190): rotation normalized()
    t = quat_rotate(a[3:], b[:3]) + a[:3]
    q = quat_mul(a[3:], b[3:])
    return np.concatenate([t, q / math.sqrt(float(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]))])


def pose_inv(a):  # Rigid3::inverse (:159-163)
    qc = np.array([a[3], -a[4], -a[5], -a[6]])
    return np.concatenate([quat_rotate(qc, -np.asarray(a[:3], np.float64)), qc])


def get_angle(p):  # transform::GetAngle (transform.h:34-37) == angularDistance to the identity
    return 2.0 * math.atan2(math.sqrt(float(p[4] * p[4] + p[5] * p[5] + p[6] * p[6])), abs(float(p[3])))


class InterpolationBuffer:
    """transform::TransformInterpolationBuffer (transform_interpolation_buffer.cc:40-140)."""

    def __init__(self, po, odometry):
        self.po = po
        self.times = [t for t, _ in odometry]
        self.poses = [p for _, p in odometry]

    def earliest(self):
        return self.times[0]

    def latest(self):
        return self.times[-1]

    def has(self, t):
        return bool(self.times) and self.times[0] <= t <= self.times[-1]

    def _interp(self, i, t):  # between entries i - 1 and i
        duration = to_seconds(self.times[i] - self.times[i - 1])
        return self.po.interpolate_transform(self.poses[i - 1], self.poses[i], to_seconds(t - self.times[i - 1]) / duration)

    def lookup(self, t):
        assert self.has(t)
        end = bisect.bisect_left(self.times, t)
        if self.times[end] == t:
            return self.poses[end]
        return self._interp(end, t)

    def lookup_until_delta(self, start_time, max_translation, max_rotation, max_duration):
        assert self.has(start_time)
        cand = bisect.bisect_left(self.times, start_time)
        start = self.poses[cand] if self.times[cand] == start_time else self._interp(cand, start_time)
        target = 1.0
        tr = rr = dr = 0.0
        while cand + 1 < len(self.times):
            delta = pose_mul(pose_inv(start), self.poses[cand + 1])
            translation_distance = abs(math.sqrt(float(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2])))
            rotation_distance = abs(get_angle(delta))
            delta_time = abs(to_seconds(self.times[cand + 1] - start_time))
            cand += 1
            tr, rr, dr = translation_distance / max_translation, rotation_distance / max_rotation, delta_time / max_duration
            target = max(tr, max(rr, dr))
            if target >= 1.0:
                break
        delta_duration = self.times[cand] - start_time
        corrected = from_seconds(to_seconds(delta_duration) / target) if target > 1.0 else delta_duration
        if target > 1.0:
            tr, rr, dr = tr / target, rr / target, dr / target
        return start_time + corrected, tr, rr, dr


def imu_delta_rotation(imu, start, end):
    """IntegrateImuWithTranslationEuler's rotation (imu_integration.h:99-131): piecewise-constant angular velocity,
    delta *= AngleAxisVectorToRotationQuaternion(w dt) (transform/transform.h:121-135). imu: [(ticks, w[3])]."""
    q = np.array([1.0, 0.0, 0.0, 0.0])
    if not imu or not (start < end):
        return q
    it = 0
    while it + 1 < len(imu) and imu[it + 1][0] <= start:
        it += 1
    cur = start
    while cur < end:
        nxt_imu = imu[it + 1][0] if it + 1 < len(imu) else (1 << 62)
        nxt = min(nxt_imu, end)
        a = np.asarray(imu[it][1]) * to_seconds(nxt - cur)
        sq = float(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])
        scale, w = 0.5, 1.0
        if sq > 1e-8:
            norm = math.sqrt(sq)
            scale, w = math.sin(norm / 2.0) / norm, math.cos(norm / 2.0)
        x, y, z = scale * a
        q = np.array([q[0] * w - q[1] * x - q[2] * y - q[3] * z, q[0] * x + q[1] * w + q[2] * z - q[3] * y,
                      q[0] * y + q[2] * w + q[3] * x - q[1] * z, q[0] * z + q[3] * w + q[1] * y - q[2] * x])
        cur = nxt
        if cur == nxt_imu:
            it += 1
    return q


class Options:
    """trajectory_builder_3d.lua:18-31,56-60,120-146, grid_type = "TSDF"."""
    min_range, max_range = 1.0, 60.0
    num_accumulated_range_data = 1
    voxel_filter_size = 0.15
    high_filter = (2.0, 150.0, 15.0)
    low_filter = (4.0, 200.0, 60.0)
    motion_filter = (0.5, 0.1, 0.004)
    high_resolution, low_resolution, num_range_data = 0.10, 0.45, 160
    high_resolution_grid_weight = low_resolution_grid_weight = 1.0
    velocity_weight = translation_weight = rotation_weight = 1.0
    odometry_translation_weight = odometry_rotation_weight = 1.0
    ct_window_horizon, ct_window_rate = 0.9, 0.1
    initialization_duration = 3.0
    use_adaptive_odometry_weights = True
    use_per_point_unwarping = False
    use_multi_resolution_matching = False
    num_points_per_subdivision = 4
    control_point_sampling = "CONSTANT"
    sampling_max_delta_translation, sampling_max_delta_rotation = 0.2, 0.1
    sampling_min_delta_time, sampling_max_delta_time = 0.025, 0.25
    velocity_in_state = True
    odometry_translation_normalization, odometry_rotation_normalization = 2.0e-2, 1.0e-1


class Submap:
    def __init__(self, po, options, local_pose):
        self.local_pose = local_pose
        self.high = po.Grid(options.high_resolution)
        self.low = po.Grid(options.low_resolution)
        self.num_range_data = 0
        self.finished = False


class OracleOLTB:
    def __init__(self, po, options):
        self.po, self.o = po, options
        self.horizon, self.rate = from_seconds(options.ct_window_horizon), from_seconds(options.ct_window_rate)
        self.init_duration = from_seconds(options.initialization_duration)
        self.have_imu = False
        self.initial_data_time = 0
        self.imu, self.odom, self.clouds, self.cps = [], [], [], []   # cps: dicts time, t[3], q[4], v[3]
        self.submaps = []
        self.motion_total, self.motion_last = 0, None
        self.num_optimizations = self.num_insertions = 0
        self.last_summary = None
        self.last_blocks = []
        self.last_imu_blocks = self.last_odometry_blocks = 0
        self.high_opts = po.InsertOpts()
        self.low_opts = po.InsertOpts(min_range=1.0, max_range=60.0, insertion_ratio=0.1,
                                      normal_computation_horizontal_stride=20, normal_computation_vertical_stride=4)
        # what the adapter inserted (the test hands the dumped range data over so that both maps stay the same map)
        self.forced_range_data = None
        self.map_update_enabled = True   # SetMapUpdateEnabled (:1493-1503: the insertion is still counted and reported)
        self.cloud_errors = []

    # ---- sensor queues (:153-186) ----
    def add_imu(self, t, w):
        if not self.have_imu:
            self.initial_data_time = t
            self.have_imu = True
        self.imu.append((t, np.asarray(w, np.float64)))

    def add_odometry(self, t, pose):
        if not self.have_imu:
            return
        if self.imu and self.imu[0][0] >= t:
            return
        self.odom.append((t, np.asarray(pose, np.float64)))

    # ---- AddRangeData (:188-264) ----
    def add_range_data(self, t, origin, ranges, width):
        po, o = self.po, self.o
        if not self.have_imu or not self.odom:
            return None
        ranges = np.asarray(ranges, np.float32)
        origin = np.asarray(origin, np.float32)
        valid = ~np.isnan(ranges[:, :3]).any(axis=1)
        d = ranges[:, :3] - origin
        rng = np.sqrt(d[:, 0] * d[:, 0] + (d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]))
        keep = valid & (rng >= np.float32(o.min_range)) & (rng <= np.float32(o.max_range))
        points = np.ascontiguousarray(ranges[keep])
        min_ts, max_ts = np.finfo(np.float32).max, np.finfo(np.float32).tiny   # (sic: numeric_limits<float>::min(), :213)
        if len(points):
            max_ts = max(max_ts, float(points[:, 3].max()))
            min_ts = min(min_ts, float(points[:, 3].min()))
        cs = {"time": t, "origin": origin, "original": ranges, "points": points, "width": width,
              "start": t + from_seconds(float(np.float32(min_ts))), "end": t + from_seconds(float(np.float32(max_ts)))}
        if self.initial_data_time > cs["start"]:
            return None
        if self.odom[0][0] > cs["start"]:
            return None
        hf, lf = o.high_filter, o.low_filter
        hi = points[po.adaptive_voxel_filter(hf[0], np.float32(hf[1]) / o.num_accumulated_range_data, hf[2], points)]
        lo = points[po.adaptive_voxel_filter(lf[0], np.float32(lf[1]) / o.num_accumulated_range_data, lf[2], points)]
        cs["high"], cs["high_t"] = np.ascontiguousarray(hi[:, :3]), hi[:, 3].copy()
        cs["low"], cs["low_t"] = np.ascontiguousarray(lo[:, :3]), lo[:, 3].copy()
        self.clouds.append(cs)
        return self.maybe_optimize(t)

    # ---- control points (:266-321, :1596-1656) ----
    def rigid(self, cp):
        return np.concatenate([cp["t"], cp["q"]])

    def predict_state_odom(self, cp, end_time):
        start_time = cp["time"]
        assert any(ti <= start_time for ti, _ in self.imu)
        buf = InterpolationBuffer(self.po, self.odom)

        def lookup(t):
            if buf.has(t):
                return buf.lookup(t)
            return buf.lookup(buf.earliest()) if t < buf.earliest() else buf.lookup(buf.latest())
        previous, current = lookup(start_time), lookup(end_time)
        delta = pose_mul(pose_inv(current), previous)   # (sic)
        dt = to_seconds(end_time - start_time)
        return {"time": end_time, "t": cp["t"] + delta[:3], "q": quat_mul(cp["q"], delta[3:]), "v": (1.0 / dt) * delta[:3]}

    def add_control_point(self, t):
        if not self.cps:
            self.cps.append({"time": t, "t": np.zeros(3), "q": np.array([1.0, 0.0, 0.0, 0.0]), "v": np.zeros(3)})
        elif not self.submaps:
            b = self.cps[-1]
            self.cps.append({"time": t, "t": b["t"].copy(), "q": b["q"].copy(), "v": b["v"].copy()})
        else:
            self.cps.append(self.predict_state_odom(self.cps[-1], t))

    def transform_states(self, tf):  # (:1097-1111)
        for cp in self.cps:
            p = pose_mul(tf, self.rigid(cp))
            cp["t"], cp["q"], cp["v"] = p[:3], p[3:], quat_rotate(tf[3:], cp["v"])

    def remove_obsolete(self):  # (:1076-1095)
        if not self.cps:
            return
        while (self.clouds and len(self.cps) > 1 and self.horizon < self.cps[-1]["time"] - self.cps[0]["time"]
               and self.cps[1]["time"] < self.clouds[0]["start"]):
            self.cps.pop(0)
        while len(self.imu) > 1 and self.imu[1][0] <= self.cps[0]["time"]:
            self.imu.pop(0)
        while len(self.odom) > 1 and self.odom[1][0] <= self.cps[0]["time"]:
            self.odom.pop(0)

    # ---- residual blocks ----
    def add_per_scan_residuals(self, pr, submap):  # (:323-511)
        o = self.o
        nxt = 0
        for cs in self.clouds:
            if cs["time"] > self.cps[-1]["time"]:
                break
            while self.cps[nxt]["time"] <= cs["time"]:
                if nxt + 1 == len(self.cps):
                    break
                nxt += 1
            assert nxt != 0 and self.cps[nxt - 1]["time"] <= cs["time"] <= self.cps[nxt]["time"]
            a, b = nxt - 1, nxt
            duration = to_seconds(self.cps[b]["time"] - self.cps[a]["time"])
            factor = to_seconds(cs["time"] - self.cps[a]["time"]) / duration
            if o.use_multi_resolution_matching:
                if o.high_resolution_grid_weight > 0.0 and len(cs["high"]):
                    scale = o.high_resolution_grid_weight / math.sqrt(float(len(cs["high"])))
                    if factor == 0.0 or factor == 1.0:
                        pa = a if factor == 0.0 else b
                        pr.add_block(cs["high"], [submap.high, submap.low], scale, pa, multi_res=True)
                        self.last_blocks.append((len(cs["high"]), pa, -1, 0.0, 2))
                    else:
                        pr.add_block(cs["high"], [submap.high, submap.low], scale, a, b, factor, multi_res=True)
                        self.last_blocks.append((len(cs["high"]), a, b, factor, 2))
                continue
            on_prev, on_next = self.cps[a]["time"] == cs["time"], self.cps[b]["time"] == cs["time"]
            for tag, cloud, grid, weight in ((0, cs["high"], submap.high, o.high_resolution_grid_weight),
                                             (1, cs["low"], submap.low, o.low_resolution_grid_weight)):
                if not (weight > 0.0 and len(cloud)):
                    continue
                scale = weight / math.sqrt(float(len(cloud)))
                if on_prev:
                    pr.add_block(cloud, [grid], scale, a)
                    self.last_blocks.append((len(cloud), a, -1, 0.0, tag))
                elif on_next:
                    pr.add_block(cloud, [grid], scale, b)
                    self.last_blocks.append((len(cloud), b, -1, 0.0, tag))
                else:
                    pr.add_block(cloud, [grid], scale, a, b, factor)
                    self.last_blocks.append((len(cloud), a, b, factor, tag))

    def _bracket(self, t):
        """The control points around time t, front < t < back (the reference's walking iterator ends on the first
        control point later than t; :550-561), and the clamped interpolation factor."""
        nxt = 1
        while self.cps[nxt]["time"] <= t:
            nxt += 1
        duration = to_seconds(self.cps[nxt]["time"] - self.cps[nxt - 1]["time"])
        return nxt - 1, nxt, min(max(to_seconds(t - self.cps[nxt - 1]["time"]) / duration, 0.0), 1.0)

    def add_per_point_residuals(self, pr, submap):  # (:513-683), one oracle block per subdivision as the reference adds them
        o = self.o
        step = o.num_points_per_subdivision
        front, back = self.cps[0]["time"], self.cps[-1]["time"]
        for cs in self.clouds:
            n = len(cs["high"])
            for start in range(0, n, step):
                end = min(start + step - 1, n - 1)
                center = 0.5 * float(np.float32(cs["high_t"][start]) + np.float32(cs["high_t"][end]))   # float + float, then double
                t = cs["time"] + from_seconds(center)
                if not (front < t < back):
                    continue
                a, b, factor = self._bracket(t)
                scale = o.high_resolution_grid_weight / math.sqrt(float(n))
                if o.use_multi_resolution_matching:
                    pr.add_block(cs["high"][start:end + 1], [submap.high, submap.low], scale, a, b, factor, multi_res=True)
                else:
                    pr.add_block(cs["high"][start:end + 1], [submap.high], scale, a, b, factor)
                self.last_blocks.append((end + 1 - start, a, b, factor, 2 if o.use_multi_resolution_matching else 0))
        if not o.use_multi_resolution_matching and o.low_resolution_grid_weight > 0:
            for cs in self.clouds:
                n = len(cs["low"])
                for i in range(n):
                    t = cs["time"] + from_seconds(float(cs["low_t"][i]))
                    if not (front < t < back):
                        continue
                    a, b, factor = self._bracket(t)
                    pr.add_block(cs["low"][i:i + 1], [submap.low], o.low_resolution_grid_weight / math.sqrt(float(n)), a, b, factor)
                    self.last_blocks.append((1, a, b, factor, 1))

    def add_imu_residuals(self, pr):  # (:928-1007) PREINTEGRATION
        o = self.o
        self.last_imu_blocks = 0
        if o.translation_weight == 0.0 and o.velocity_weight == 0.0 and o.rotation_weight == 0.0:
            return
        assert o.velocity_in_state
        assert any(ti <= self.cps[0]["time"] for ti, _ in self.imu)
        for i in range(1, len(self.cps)):
            dq = imu_delta_rotation(self.imu, self.cps[i - 1]["time"], self.cps[i]["time"])
            pr.add_imu_block(i - 1, i, o.translation_weight, o.velocity_weight, o.rotation_weight,
                             to_seconds(self.cps[i]["time"] - self.cps[i - 1]["time"]), dq)
            self.last_imu_blocks += 1

    def add_odometry_residuals(self, pr):  # (:1009-1074)
        o = self.o
        self.last_odometry_blocks = 0
        if len(self.odom) <= 1:
            return
        buf = InterpolationBuffer(self.po, self.odom)
        for i in range(1, len(self.cps)):
            if not (buf.earliest() <= self.cps[i - 1]["time"] and self.cps[i]["time"] <= buf.latest()):
                continue
            previous, current = buf.lookup(self.cps[i - 1]["time"]), buf.lookup(self.cps[i]["time"])
            delta = pose_mul(pose_inv(current), previous)
            dt = to_seconds(self.cps[i]["time"] - self.cps[i - 1]["time"])
            tw, rw = o.odometry_translation_weight, o.odometry_rotation_weight
            if o.use_adaptive_odometry_weights:
                translation_distance = abs(math.sqrt(float(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2])))
                rotation_distance = abs(get_angle(delta))
                tw = o.odometry_translation_weight / math.sqrt(translation_distance + o.odometry_translation_normalization * dt)
                rw = o.odometry_rotation_weight / math.sqrt(rotation_distance + o.odometry_rotation_normalization * dt)
            pr.add_odometry_block(i - 1, i, tw, rw, delta)
            self.last_odometry_blocks += 1

    # ---- MaybeOptimize (:1113-1413) ----
    def maybe_optimize(self, time):
        po, o = self.po, self.o
        if time - self.initial_data_time < self.init_duration:
            return None
        if len(self.odom) < 2:
            return None
        if not self.cps:
            self.add_control_point(max(self.initial_data_time, self.odom[0][0]))
        added = False
        if o.control_point_sampling == "CONSTANT":
            while self.cps[-1]["time"] + self.rate < self.odom[-1][0]:
                self.add_control_point(self.cps[-1]["time"] + self.rate)
                added = True
        elif o.control_point_sampling == "SYNCED_WITH_RANGE_DATA":
            for cs in self.clouds:
                if self.cps[-1]["time"] < cs["time"] < self.imu[-1][0]:
                    self.add_control_point(cs["time"])
                    added = True
        else:  # ADAPTIVE
            buf = InterpolationBuffer(po, self.odom)
            cand = self.cps[-1]["time"]
            while cand < buf.latest():
                cand, _, _, _ = buf.lookup_until_delta(self.cps[-1]["time"], o.sampling_max_delta_translation,
                                                       o.sampling_max_delta_rotation, o.sampling_max_delta_time)
                if to_seconds(cand - self.cps[-1]["time"]) < o.sampling_min_delta_time:
                    cand = self.cps[-1]["time"] + from_seconds(o.sampling_min_delta_time)
                if cand < buf.latest():
                    self.add_control_point(cand)
                    added = True
        if not added:
            return None
        solved = False
        if self.submaps:
            submap = self.submaps[0]
            inv = pose_inv(submap.local_pose)
            assert abs(abs(inv[3]) - 1.0) < 1e-8
            self.transform_states(inv)
            pr = po.Problem()
            self.last_blocks = []
            for i, cp in enumerate(self.cps):
                pr.add_pose(self.rigid(cp), i == 0)
                if o.velocity_in_state:
                    pr.set_velocity(i, cp["v"], i == 0)
            if o.use_per_point_unwarping:
                self.add_per_point_residuals(pr, submap)
            else:
                self.add_per_scan_residuals(pr, submap)
            self.add_imu_residuals(pr)
            self.add_odometry_residuals(pr)
            self.last_summary = pr.solve()
            self.num_optimizations += 1
            solved = True
            for i, cp in enumerate(self.cps):
                p = pr.get_pose(i)
                cp["t"], cp["q"] = p[:3].copy(), p[3:].copy()
                if o.velocity_in_state:
                    cp["v"] = pr.get_velocity(i).copy()
            self.transform_states(submap.local_pose)
        optimized = self.rigid(self.cps[0])
        time_optimized = self.cps[0]["time"]
        opt_inv = pose_inv(optimized)
        returns, origin = [], np.zeros(3, np.float32)
        width = self.clouds[0]["width"]
        if not self.submaps:
            it = 0
            for cs in self.clouds:   # (:1301-1330) zero motion, the clouds stay queued
                if not (cs["time"] < self.cps[-1]["time"]):
                    continue
                while self.cps[it]["time"] <= cs["time"]:
                    it += 1
                assert 0 < it < len(self.cps)
                a, b = self.cps[it - 1], self.cps[it]
                factor = to_seconds(cs["time"] - a["time"]) / to_seconds(b["time"] - a["time"])
                tf = pose_mul(opt_inv, po.interpolate_transform(self.rigid(a), self.rigid(b), factor)).astype(np.float32)
                returns.append(po.transform_points(tf, cs["original"][:, :3]))
                origin = po.transform_points(tf, cs["origin"][None])[0]
        elif o.use_per_point_unwarping:   # (:1331-1379)
            assert self.cps[0]["time"] <= self.clouds[0]["start"]
            leaving = []
            while (self.clouds and self.horizon < self.cps[-1]["time"] - self.clouds[0]["start"]
                   and self.cps[-1]["time"] > self.clouds[0]["end"]):
                cs = self.clouds.pop(0)
                leaving.append((cs["time"], cs["origin"], cs["points"]))
            if leaving:
                xyz, org, ok = po.unwarp_range_data(np.array([c["time"] for c in self.cps], np.int64),
                                                    np.array([self.rigid(c) for c in self.cps]), leaving)
                assert ok
                returns.append(xyz)
                origin = org
        else:
            assert self.cps[0]["time"] <= self.clouds[0]["time"]
            while self.clouds and self.horizon - self.rate < self.cps[-1]["time"] - self.clouds[0]["time"]:   # (:1382-1404)
                while self.cps[1]["time"] < self.clouds[0]["time"]:
                    self.cps.pop(0)
                a, b, cs = self.cps[0], self.cps[1], self.clouds[0]
                factor = to_seconds(cs["time"] - a["time"]) / to_seconds(b["time"] - a["time"])
                tf = pose_mul(opt_inv, po.interpolate_transform(self.rigid(a), self.rigid(b), factor)).astype(np.float32)
                returns.append(po.transform_points(tf, cs["points"][:, :3]))
                origin = po.transform_points(tf, cs["origin"][None])[0]
                self.clouds.pop(0)
        self.remove_obsolete()
        returns = np.concatenate(returns, 0) if returns else np.zeros((0, 3), np.float32)
        return self.add_accumulated(time_optimized, optimized, origin, returns, width, solved)

    def add_accumulated(self, time, optimized, origin, returns, width, solved):  # (:1415-1514)
        po, o = self.po, self.o
        if len(returns) == 0:
            return None
        filtered = returns[po.voxel_filter(o.voxel_filter_size, returns)]
        if len(filtered) == 0:
            return None
        to_local = optimized.astype(np.float32)
        local_returns = po.transform_points(to_local, returns)
        local_origin = po.transform_points(to_local, origin[None])[0]
        hf, lf = o.high_filter, o.low_filter
        if len(po.adaptive_voxel_filter(hf[0], hf[1], hf[2], filtered)) == 0:
            return None
        if len(po.adaptive_voxel_filter(lf[0], lf[1], lf[2], filtered)) == 0:
            return None
        result = {"time": time, "local_pose": optimized, "inserted": False, "solved": solved}
        # motion filter (motion_filter.cc:40-58)
        self.motion_total += 1
        mt, md, ma = o.motion_filter
        if self.motion_total > 1 and time - self.motion_last[0] <= from_seconds(mt):
            lp = self.motion_last[1]
            if (math.sqrt(float(((optimized[:3] - lp[:3]) ** 2).sum())) <= md and get_angle(pose_mul(pose_inv(optimized), lp)) <= ma):
                return result
        self.motion_last = (time, optimized.copy())
        rot = np.concatenate([np.zeros(3), optimized[3:]])
        lfg = pose_mul(rot, pose_inv(rot))[3:]
        if self.forced_range_data is not None:   # the adapter's float range data (compared with this one's first)
            f_origin, f_returns = self.forced_range_data
            assert f_returns.shape == local_returns.shape
            # (the cloud that initialises the map carries the driver's NaN returns along: the same ones in both)
            assert np.array_equal(np.isnan(f_returns), np.isnan(local_returns))
            self.cloud_errors.append(float(np.nanmax(np.abs(f_returns - local_returns))) if len(local_returns) else 0.0)
            self.cloud_errors.append(float(np.abs(f_origin - local_origin).max()))
            local_origin, local_returns = f_origin, f_returns
        if self.map_update_enabled:
            self.insert_into_submaps(local_origin, local_returns, width, lfg)
        self.num_insertions += 1
        result["inserted"] = True
        return result

    def insert_into_submaps(self, origin, returns, width, lfg):  # ActiveSubmaps3D::InsertData (submap_3d.cc:492-514)
        o = self.o
        if not self.submaps or self.submaps[-1].num_range_data == o.num_range_data:
            if len(self.submaps) >= 2:
                self.submaps.pop(0)
            self.submaps.append(Submap(self.po, o, np.concatenate([origin.astype(np.float64), lfg])))
        for sm in self.submaps:
            inv = pose_inv(sm.local_pose).astype(np.float32)
            sm.high.insert(origin, returns, self.high_opts, width=width, pose_tq=inv)
            sm.low.insert(origin, returns, self.low_opts, width=width, pose_tq=inv)
            sm.num_range_data += 1
        if self.submaps[0].num_range_data == 2 * o.num_range_data:
            self.submaps[0].finished = True


# ---- the C++ builder's dump (cpp/example_oltb.cc) and its comparison with the replay ----
def parse_stdout(text):
    steps, cur = [], None
    for line in text.splitlines():
        m = re.match(r"scan (\d+) time (-?\d+) result (\d) solved (\d) iterations (\d+) termination (\d+) (\d+) queued (\d+) "
                     r"imu_blocks (\d+) odometry_blocks (\d+) residuals (\d+)", line)
        if m:
            v = [int(x) for x in m.groups()]
            cur = {"scan": v[0], "time": v[1], "result": v[2], "solved": v[3], "it": v[4], "term": (v[5], v[6]), "queued": v[7],
                   "imu_blocks": v[8], "odometry_blocks": v[9], "residuals": v[10], "blocks": [], "cps": [], "local_pose": None,
                   "inserted": 0}
            steps.append(cur)
            continue
        m = re.match(r"\s+block (\d+) (-?\d+) (-?\d+) (\S+) (\d)", line)
        if m:
            cur["blocks"].append((int(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(4)), int(m.group(5))))
            continue
        m = re.match(r"\s+cp (-?\d+) pose (.*) vel (.*)", line)
        if m:
            cur["cps"].append((int(m.group(1)), np.array([float(x) for x in m.group(2).split()]),
                               np.array([float(x) for x in m.group(3).split()])))
            continue
        m = re.match(r"\s+local_pose (-?\d+) (.*) inserted (\d+) submaps (\d+)", line)
        if m:
            cur["local_pose"] = (int(m.group(1)), np.array([float(x) for x in m.group(2).split()]))
            cur["inserted"] = int(m.group(3))
            cur["submaps"] = int(m.group(4))
    return steps


def messages(raw):
    off = 4
    while off < len(raw):
        (kind,) = struct.unpack_from("i", raw, off); off += 4
        if kind == 0:
            (t,) = struct.unpack_from("q", raw, off); off += 8
            w = np.frombuffer(raw, np.float64, 3, off).copy(); off += 24
            yield ("imu", t, w)
        elif kind == 1:
            (t,) = struct.unpack_from("q", raw, off); off += 8
            pose = np.frombuffer(raw, np.float64, 7, off).copy(); off += 56
            yield ("odom", t, pose)
        elif kind == 2:
            (t,) = struct.unpack_from("q", raw, off); off += 8
            origin = np.frombuffer(raw, np.float32, 3, off).copy(); off += 12
            (n,) = struct.unpack_from("i", raw, off); off += 4
            pts = np.frombuffer(raw, np.float32, n * 4, off).reshape(n, 4).copy(); off += 16 * n
            yield ("scan", t, origin, pts)
        elif kind == 3:
            (n,) = struct.unpack_from("i", raw, off); off += 4
            if n:
                origin = np.frombuffer(raw, np.float32, 3, off).copy(); off += 12
                ret = np.frombuffer(raw, np.float32, n * 3, off).reshape(n, 3).copy(); off += 12 * n
                yield ("inserted", origin, ret)
            else:
                yield ("inserted", None, None)
        elif kind == 4:
            pose = np.frombuffer(raw, np.float64, 7, off).copy(); off += 56
            (num,) = struct.unpack_from("i", raw, off); off += 4
            grids = []
            for _ in range(2):
                (n,) = struct.unpack_from("i", raw, off); off += 4
                cells = np.frombuffer(raw, np.int32, n * 3, off).reshape(n, 3).copy(); off += 12 * n
                tsd = np.frombuffer(raw, np.uint16, n, off).copy(); off += 2 * n
                weight = np.frombuffer(raw, np.uint16, n, off).copy(); off += 2 * n
                grids.append((cells, tsd, weight))
            yield ("submap", pose, num, grids)
        else:
            raise AssertionError("bad record kind %d" % kind)


MODES = {
    0: dict(),                                                      # the Lua defaults
    1: dict(control_point_sampling="SYNCED_WITH_RANGE_DATA"),
    2: dict(control_point_sampling="ADAPTIVE", use_multi_resolution_matching=True, sampling_max_delta_translation=0.03),
    3: dict(use_per_point_unwarping=True),
    4: dict(use_per_point_unwarping=True),   # + SetMapUpdateEnabled(false) from scan 30 on (example_oltb.cc)
}




def replay_and_compare(po, raw, stdout_text, mode, scans, width=12, check_window=True):
    """Replays the messages example_oltb dumped (`raw`) through OracleOLTB and compares every step with what the C++
    builder printed (`stdout_text`). Returns a dict of what was seen; raises AssertionError where the two differ."""
    import struct
    import time as _time
    rp = sys.modules[__name__]
    gpu = parse_stdout(stdout_text)
    assert len(gpu) == scans
    assert struct.unpack_from("i", raw, 0)[0] == mode
    cpu_seconds = 0.0
    opt = Options()
    opt.initialization_duration = 0.2
    opt.num_range_data = 4
    for k, v in MODES[mode].items():
        setattr(opt, k, v)
    b = OracleOLTB(po, opt)

    msgs = list(messages(raw))
    max_dt = max_dr = max_dv = 0.0
    solves = interpolated = single = 0
    k = 0
    i = 0
    final_submaps = []
    while i < len(msgs):
        m = msgs[i]
        if m[0] == "imu":
            b.add_imu(m[1], m[2])
        elif m[0] == "odom":
            b.add_odometry(m[1], m[2])
        elif m[0] == "submap":
            final_submaps.append(m[1:])
        elif m[0] == "scan":
            ins = msgs[i + 1]
            assert ins[0] == "inserted"
            i += 1
            b.forced_range_data = None if ins[1] is None else (ins[1], ins[2])
            if mode == 4 and k == 30:
                b.map_update_enabled = False
            before = b.num_optimizations
            _t0 = _time.perf_counter()
            res = b.add_range_data(m[1], m[2], m[3], width)
            cpu_seconds += _time.perf_counter() - _t0
            g = gpu[k]
            assert g["time"] == m[1]
            solved = b.num_optimizations - before
            assert g["solved"] == solved, (k, g["solved"], solved)
            assert g["result"] == (1 if res is not None else 0), k
            assert g["queued"] == len(b.clouds), (k, g["queued"], len(b.clouds))
            if solved:
                solves += 1
                so = b.last_summary
                assert (so.num_iterations, so.termination_type, so.termination_reason) == (g["it"], g["term"][0], g["term"][1]), k
                assert g["imu_blocks"] == b.last_imu_blocks and g["odometry_blocks"] == b.last_odometry_blocks, k
                assert g["residuals"] == sum(x[0] for x in b.last_blocks) + 9 * b.last_imu_blocks + 6 * b.last_odometry_blocks, k
                if mode in (3, 4):   # (the device merges the subdivisions between two control points into one block)
                    interpolated += len(b.last_blocks)
                else:
                    assert len(g["blocks"]) == len(b.last_blocks), k
                for gb, ob in zip(g["blocks"] if mode not in (3, 4) else [], b.last_blocks):
                    assert gb[:3] == tuple(ob[:3]) and gb[4] == ob[4], (k, gb, ob)
                    assert gb[3] == ob[3], (k, gb, ob)       # the interpolation factor: same ticks, same division
                    if gb[2] >= 0:
                        interpolated += 1
                        assert 0.0 < gb[3] < 1.0
                    else:
                        single += 1
            if res is not None:
                assert g["local_pose"][0] == res["time"]
                assert g["inserted"] == (1 if res["inserted"] else 0), k
                assert (ins[1] is not None) == res["inserted"]
            # the window as it stands after the step
            assert (not check_window) or [c[0] for c in g["cps"]] == [c["time"] for c in b.cps], k
            for (t, gp, gv), c in (zip(g["cps"], b.cps) if check_window else []):
                max_dt = max(max_dt, float(np.linalg.norm(gp[:3] - c["t"])))
                max_dr = max(max_dr, float(2.0 * np.arccos(min(1.0, abs(float(gp[3:] @ c["q"]))))))
                max_dv = max(max_dv, float(np.linalg.norm(gv - c["v"])))
            k += 1
        i += 1
    assert k == scans
    return {"b": b, "solves": solves, "interpolated": interpolated, "single": single, "max_dt": max_dt, "max_dr": max_dr,
            "max_dv": max_dv, "final_submaps": final_submaps, "cpu_seconds": cpu_seconds}
