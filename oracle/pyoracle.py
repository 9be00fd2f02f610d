"""ctypes binding of the CPU oracle (oracle/libhg_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg. The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HG_ORACLE_SANITIZE=1: the same sources built with -fsanitize=undefined, aborting on the first finding
# (oracle/Makefile: libhg_oracle_ubsan.so; tests/test_oracle_kat.py::test_oracle_under_ubsan runs the known-answer
# tests, an insert, a solve and the unwarping through it)
_SAN = os.environ.get("HG_ORACLE_SANITIZE") == "1"
_LIB_NAME = "libhg_oracle_ubsan.so" if _SAN else "libhg_oracle.so"
_LIB_PATH = os.path.join(_HERE, _LIB_NAME)


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("hg_oracle.hpp", "hg_oracle_capi.cc")):
        subprocess.check_call(["make", "-C", _HERE, "-B", _LIB_NAME],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


class InsertOpts(C.Structure):
    # trajectory_builder_3d.lua:78-93 defaults (high resolution inserter)
    _fields_ = [("relative_truncation_distance", C.c_double),
                ("maximum_weight", C.c_double),
                ("num_free_space_voxels", C.c_int),
                ("project_sdf_distance_to_scan_normal", C.c_int),
                ("weight_function_epsilon", C.c_double),
                ("weight_function_sigma", C.c_double),
                ("min_range", C.c_double),
                ("max_range", C.c_double),
                ("insertion_ratio", C.c_double),
                ("normal_computation_method", C.c_int),
                ("normal_computation_horizontal_stride", C.c_int),
                ("normal_computation_vertical_stride", C.c_int),
                ("reserved", C.c_int)]

    def __init__(self, **kw):
        super().__init__()
        self.relative_truncation_distance = 2.5
        self.maximum_weight = 1000.0
        self.num_free_space_voxels = 0
        self.project_sdf_distance_to_scan_normal = 0
        self.weight_function_epsilon = 1.0
        self.weight_function_sigma = 4.0
        self.min_range = 0.4
        self.max_range = 15.0
        self.insertion_ratio = 1.0
        self.normal_computation_method = 1
        self.normal_computation_horizontal_stride = 5
        self.normal_computation_vertical_stride = 1
        for k, v in kw.items():
            setattr(self, k, v)


class SolverOpts(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int), ("jacobi_scaling", C.c_int),
                ("initial_trust_region_radius", C.c_double),
                ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double),
                ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double)]


class SolverSummary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("final_radius", C.c_double),
                ("num_iterations", C.c_int), ("num_successful_steps", C.c_int),
                ("num_unsuccessful_steps", C.c_int), ("num_cost_evaluations", C.c_int),
                ("num_jacobian_evaluations", C.c_int), ("termination_type", C.c_int),
                ("termination_reason", C.c_int), ("reserved", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    vp, f32, f64, sz = C.c_void_p, C.c_float, C.c_double, C.c_size_t
    P = C.POINTER
    L.hgo_conversion_table.argtypes = [f32, f32, f32, vp]
    L.hgo_converter_create.restype = vp
    L.hgo_converter_create.argtypes = [f32, f32]
    L.hgo_converter_destroy.argtypes = [vp]
    L.hgo_tsd_to_value.restype = C.c_uint16
    L.hgo_tsd_to_value.argtypes = [vp, f32]
    L.hgo_weight_to_value.restype = C.c_uint16
    L.hgo_weight_to_value.argtypes = [vp, f32]
    L.hgo_value_to_tsd.restype = f32
    L.hgo_value_to_tsd.argtypes = [vp, C.c_uint16]
    L.hgo_value_to_weight.restype = f32
    L.hgo_value_to_weight.argtypes = [vp, C.c_uint16]
    L.hgo_grid_create.restype = vp
    L.hgo_grid_create.argtypes = [f32, f32, f32]
    L.hgo_grid_destroy.argtypes = [vp]
    L.hgo_grid_resolution.restype = f32
    L.hgo_grid_resolution.argtypes = [vp]
    L.hgo_grid_max_tsd.restype = f32
    L.hgo_grid_max_tsd.argtypes = [vp]
    L.hgo_grid_cell_index.argtypes = [vp, vp, sz, vp]
    L.hgo_grid_center_of_cell.argtypes = [vp, vp, sz, vp]
    L.hgo_grid_set_cell.restype = C.c_int
    L.hgo_grid_set_cell.argtypes = [vp, C.c_int, C.c_int, C.c_int, f32, f32]
    L.hgo_grid_read_cells.argtypes = [vp, vp, sz, vp, vp]
    L.hgo_grid_get_float.argtypes = [vp, vp, sz, vp, vp, vp]
    L.hgo_grid_count.restype = sz
    L.hgo_grid_count.argtypes = [vp]
    L.hgo_grid_xray.restype = sz
    L.hgo_grid_xray.argtypes = [vp, vp, vp, sz, vp, vp, vp]
    L.hgo_grid_export.restype = sz
    L.hgo_grid_export.argtypes = [vp, vp, vp, vp, sz]
    L.hgo_grid_insert.restype = C.c_int
    L.hgo_grid_insert.argtypes = [vp, P(InsertOpts), vp, vp, sz, sz, vp, vp]
    L.hgo_voxel_filter.restype = sz
    L.hgo_voxel_filter.argtypes = [f32, vp, sz, C.c_int, vp]
    L.hgo_adaptive_voxel_filter.restype = sz
    L.hgo_adaptive_voxel_filter.argtypes = [f32, f32, f32, vp, sz, C.c_int, vp]
    L.hgo_interp_tsd.argtypes = [vp, C.c_int, C.c_int, vp, sz, vp, vp]
    L.hgo_interpolate_transform.argtypes = [vp, vp, f64, vp]
    L.hgo_quaternion_plus.argtypes = [vp, vp, vp]
    L.hgo_unwarp_range_data.restype = C.c_int
    L.hgo_unwarp_range_data.argtypes = [vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp]
    L.hgo_transform_points.argtypes = [vp, vp, sz, vp]
    L.hgo_problem_create.restype = vp
    L.hgo_problem_destroy.argtypes = [vp]
    L.hgo_problem_add_pose.restype = C.c_int
    L.hgo_problem_add_pose.argtypes = [vp, vp, C.c_int]
    L.hgo_problem_set_pose.argtypes = [vp, C.c_int, vp]
    L.hgo_problem_get_pose.argtypes = [vp, C.c_int, vp]
    L.hgo_problem_set_velocity.argtypes = [vp, C.c_int, vp, C.c_int]
    L.hgo_problem_get_velocity.argtypes = [vp, C.c_int, vp]
    L.hgo_problem_add_odometry_block.restype = C.c_int
    L.hgo_problem_add_odometry_block.argtypes = [vp, C.c_int, C.c_int, f64, f64, vp]
    L.hgo_problem_add_imu_block.restype = C.c_int
    L.hgo_problem_add_imu_block.argtypes = [vp, C.c_int, C.c_int, f64, f64, f64, f64, vp]
    L.hgo_problem_add_block.restype = C.c_int
    L.hgo_problem_add_block.argtypes = [vp, vp, sz, vp, C.c_int, C.c_int, f64, C.c_int, C.c_int,
                                        f64]
    L.hgo_problem_num_residuals.restype = C.c_int
    L.hgo_problem_num_residuals.argtypes = [vp]
    L.hgo_problem_num_columns.restype = C.c_int
    L.hgo_problem_num_columns.argtypes = [vp]
    L.hgo_problem_evaluate.argtypes = [vp, vp, vp, vp, vp]
    L.hgo_problem_lookup_stats.argtypes = [vp, vp]
    L.hgo_solver_default_opts.argtypes = [P(SolverOpts)]
    L.hgo_problem_solve.argtypes = [vp, P(SolverOpts), P(SolverSummary)]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def conversion_table(unknown_result, lower, upper):
    out = np.empty(65536, np.float32)
    lib().hgo_conversion_table(unknown_result, lower, upper, _ptr(out))
    return out


class Converter:
    """TSDValueConverter (ref mapping/2d/tsd_value_converter.h)."""

    def __init__(self, max_tsd, max_weight):
        self._h = lib().hgo_converter_create(max_tsd, max_weight)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().hgo_converter_destroy(self._h)
            self._h = None

    def tsd_to_value(self, x):
        return lib().hgo_tsd_to_value(self._h, float(x))

    def weight_to_value(self, x):
        return lib().hgo_weight_to_value(self._h, float(x))

    def value_to_tsd(self, v):
        return lib().hgo_value_to_tsd(self._h, int(v))

    def value_to_weight(self, v):
        return lib().hgo_value_to_weight(self._h, int(v))


class Grid:
    """HybridGridTSDF (ref mapping/3d/hybrid_grid_tsdf.h)."""

    def __init__(self, resolution, relative_truncation_distance=2.5, max_weight=1000.0):
        self._h = lib().hgo_grid_create(resolution, relative_truncation_distance, max_weight)
        self.resolution = np.float32(resolution)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().hgo_grid_destroy(self._h)
            self._h = None

    def cell_index(self, xyz):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        out = np.empty(xyz.shape, np.int32)
        lib().hgo_grid_cell_index(self._h, _ptr(xyz), len(xyz), _ptr(out))
        return out

    def center_of_cell(self, ijk):
        ijk = np.ascontiguousarray(ijk, np.int32).reshape(-1, 3)
        out = np.empty(ijk.shape, np.float32)
        lib().hgo_grid_center_of_cell(self._h, _ptr(ijk), len(ijk), _ptr(out))
        return out

    def set_cell(self, ijk, tsd, weight):
        return lib().hgo_grid_set_cell(self._h, int(ijk[0]), int(ijk[1]), int(ijk[2]),
                                       float(tsd), float(weight))

    def read_cells(self, ijk):
        ijk = np.ascontiguousarray(ijk, np.int32).reshape(-1, 3)
        t = np.empty(len(ijk), np.uint16)
        w = np.empty(len(ijk), np.uint16)
        lib().hgo_grid_read_cells(self._h, _ptr(ijk), len(ijk), _ptr(t), _ptr(w))
        return t, w

    def get_float(self, ijk):
        ijk = np.ascontiguousarray(ijk, np.int32).reshape(-1, 3)
        t = np.empty(len(ijk), np.float32)
        w = np.empty(len(ijk), np.float32)
        k = np.empty(len(ijk), np.uint8)
        lib().hgo_grid_get_float(self._h, _ptr(ijk), len(ijk), _ptr(t), _ptr(w), _ptr(k))
        return t, w, k.astype(bool)

    def count(self):
        return lib().hgo_grid_count(self._h)

    def export(self):
        n = self.count()
        ijk = np.empty((n, 3), np.int32)
        t = np.empty(n, np.uint16)
        w = np.empty(n, np.uint16)
        lib().hgo_grid_export(self._h, _ptr(ijk), _ptr(t), _ptr(w), n)
        return ijk, t, w

    def xray(self, global_submap_pose):
        """X-ray texture (submap_3d.cc:245-276): (cells uint8 [height, width, 2], max_index xy)."""
        pose = np.ascontiguousarray(global_submap_pose, np.float64)
        w, h = C.c_int32(), C.c_int32()
        mx = np.zeros(2, np.int32)
        n = lib().hgo_grid_xray(self._h, _ptr(pose), None, 0, C.byref(w), C.byref(h), _ptr(mx))
        cells = np.empty(n, np.uint8)
        lib().hgo_grid_xray(self._h, _ptr(pose), _ptr(cells), n, C.byref(w), C.byref(h), _ptr(mx))
        return cells.reshape(h.value, w.value, 2), mx

    def insert(self, origin, xyz, opts=None, width=0, pose_tq=None):
        """TSDFRangeDataInserter3D::Insert. Returns (N_in, U)."""
        opts = opts or InsertOpts()
        origin = np.ascontiguousarray(origin, np.float32)
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        pose = None if pose_tq is None else np.ascontiguousarray(pose_tq, np.float32)
        st = np.zeros(2, np.uint64)
        rc = lib().hgo_grid_insert(self._h, C.byref(opts), _ptr(origin), _ptr(xyz), len(xyz),
                                   int(width), _ptr(pose), _ptr(st))
        if rc != 0:
            raise RuntimeError("oracle insert failed rc=%d" % rc)
        return int(st[0]), int(st[1])


def voxel_filter(resolution, pts):
    """VoxelFilter(resolution).Filter: indices of the kept points (first per voxel, input order)."""
    pts = np.ascontiguousarray(pts, np.float32)
    stride = pts.shape[1]
    out = np.empty(len(pts), np.uint32)
    n = lib().hgo_voxel_filter(resolution, _ptr(pts), len(pts), stride, _ptr(out))
    return out[:n].copy()


def adaptive_voxel_filter(max_length, min_num_points, max_range, pts):
    pts = np.ascontiguousarray(pts, np.float32)
    stride = pts.shape[1]
    out = np.empty(len(pts), np.uint32)
    n = lib().hgo_adaptive_voxel_filter(max_length, min_num_points, max_range, _ptr(pts), len(pts),
                                        stride, _ptr(out))
    return out[:n].copy()


def interp_tsd(grids, xyz, multi_res=False):
    xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
    arr = (C.c_void_p * len(grids))(*[g._h for g in grids])
    val = np.empty(len(xyz), np.float64)
    grad = np.empty((len(xyz), 3), np.float64)
    lib().hgo_interp_tsd(arr, len(grids), int(multi_res), _ptr(xyz), len(xyz), _ptr(val),
                         _ptr(grad))
    return val, grad


def interpolate_transform(a, b, factor):
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    out = np.empty(7, np.float64)
    lib().hgo_interpolate_transform(_ptr(a), _ptr(b), float(factor), _ptr(out))
    return out


def unwarp_range_data(control_times, control_poses, clouds):
    """UnwarpAccumulatedRangeData (oltb.cc:1331-1379). clouds: [(time_ticks, origin[3], points[n, 4])].
    Returns (xyz [n, 3] float32, origin [3] float32, time_in_range)."""
    ct = np.ascontiguousarray(control_times, np.int64)
    cp = np.ascontiguousarray(control_poses, np.float64).reshape(-1, 7)
    times = np.ascontiguousarray([c[0] for c in clouds], np.int64)
    origins = np.ascontiguousarray([c[1] for c in clouds], np.float32).reshape(-1, 3)
    pts = [np.ascontiguousarray(c[2], np.float32).reshape(-1, 4) for c in clouds]
    offs = np.zeros(len(clouds) + 1, np.uint64)
    offs[1:] = np.cumsum([len(p) for p in pts])
    allp = np.ascontiguousarray(np.concatenate(pts, 0)) if pts else np.zeros((0, 4), np.float32)
    xyz = np.empty((len(allp), 3), np.float32)
    origin = np.zeros(3, np.float32)
    ok = lib().hgo_unwarp_range_data(_ptr(ct), _ptr(cp), len(ct), _ptr(times), _ptr(offs), _ptr(origins),
                                     len(clouds), _ptr(allp), _ptr(xyz), _ptr(origin))
    return xyz, origin, bool(ok)


def transform_points(tq, xyz):
    """Rigid3f * point for every row (sensor::TransformRangeData, range_data.cc:25-39)."""
    tq = np.ascontiguousarray(tq, np.float32)
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    out = np.empty_like(xyz)
    lib().hgo_transform_points(_ptr(tq), _ptr(xyz), len(xyz), _ptr(out))
    return out


def quaternion_plus(q, delta):
    q = np.ascontiguousarray(q, np.float64)
    delta = np.ascontiguousarray(delta, np.float64)
    out = np.empty(4, np.float64)
    lib().hgo_quaternion_plus(_ptr(q), _ptr(delta), _ptr(out))
    return out


class Problem:
    """ceres::Problem restricted to TSDF scan-matching blocks over pose blocks."""

    def __init__(self):
        self._h = lib().hgo_problem_create()
        self._keep = []

    def __del__(self):
        if getattr(self, "_h", None):
            lib().hgo_problem_destroy(self._h)
            self._h = None

    def add_pose(self, tq, constant=False):
        tq = np.ascontiguousarray(tq, np.float64)
        return lib().hgo_problem_add_pose(self._h, _ptr(tq), int(constant))

    def set_pose(self, idx, tq):
        tq = np.ascontiguousarray(tq, np.float64)
        lib().hgo_problem_set_pose(self._h, idx, _ptr(tq))

    def get_pose(self, idx):
        out = np.empty(7, np.float64)
        lib().hgo_problem_get_pose(self._h, idx, _ptr(out))
        return out

    def set_velocity(self, idx, v, constant=False):
        v = np.ascontiguousarray(v, np.float64)
        lib().hgo_problem_set_velocity(self._h, idx, _ptr(v), int(constant))

    def get_velocity(self, idx):
        out = np.empty(3, np.float64)
        lib().hgo_problem_get_velocity(self._h, idx, _ptr(out))
        return out

    def add_odometry_block(self, a, b, translation_weight, rotation_weight, delta_tq):
        d = np.ascontiguousarray(delta_tq, np.float64)
        return lib().hgo_problem_add_odometry_block(self._h, a, b, float(translation_weight),
                                                    float(rotation_weight), _ptr(d))

    def add_imu_block(self, a, b, translation_weight, velocity_weight, rotation_weight, dt, delta_q):
        d = np.ascontiguousarray(delta_q, np.float64)
        return lib().hgo_problem_add_imu_block(self._h, a, b, float(translation_weight),
                                               float(velocity_weight), float(rotation_weight),
                                               float(dt), _ptr(d))

    def add_block(self, xyz, grids, scaling_factor, pose_a, pose_b=-1, interpolation_ratio=0.0,
                  multi_res=False):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        arr = (C.c_void_p * len(grids))(*[g._h for g in grids])
        self._keep.append((xyz, grids, arr))
        return lib().hgo_problem_add_block(self._h, _ptr(xyz), len(xyz), arr, len(grids),
                                           int(multi_res), float(scaling_factor), pose_a, pose_b,
                                           float(interpolation_ratio))

    def evaluate(self, want_jacobian=True):
        n = lib().hgo_problem_num_residuals(self._h)
        c = lib().hgo_problem_num_columns(self._h)
        cost = C.c_double()
        r = np.empty(n, np.float64)
        J = np.empty((n, c), np.float64) if want_jacobian else None
        g = np.empty(c, np.float64) if want_jacobian else None
        lib().hgo_problem_evaluate(self._h, C.byref(cost), _ptr(r), _ptr(J), _ptr(g))
        return cost.value, r, J, g

    def lookup_stats(self):
        out = np.zeros(2, np.uint64)
        lib().hgo_problem_lookup_stats(self._h, _ptr(out))
        return int(out[0]), int(out[1])

    def solve(self, **kw):
        o = SolverOpts()
        lib().hgo_solver_default_opts(C.byref(o))
        for k, v in kw.items():
            setattr(o, k, v)
        s = SolverSummary()
        lib().hgo_problem_solve(self._h, C.byref(o), C.byref(s))
        return s
