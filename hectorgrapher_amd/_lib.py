"""ctypes loader of the HIP C-ABI library (libhg_mi355x.so, see include/hg_mi355x.h).

There is no CPU fallback: if the library is missing or no GPU is visible the
product path raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HG_LIB_PATH: a diagnostic build of the same library (e.g. with in-kernel stamps, scripts/diag_*.py)
LIB_PATH = os.environ.get("HG_LIB_PATH") or os.path.join(_HERE, "libhg_mi355x.so")

HG_OK = 0
HG_HOST, HG_DEVICE = 0, 1
HG_INSERT_EXACT = 0
HG_INSERT_FAST = 1
KERNELS = {"ray_count": 0, "ray_expand": 1, "sort": 2, "alloc": 3, "apply": 4, "residuals": 5,
           "lm": 6, "scan": 7, "unwarp": 8}

ERRORS = {-1: "HG_ERR_INVALID", -2: "HG_ERR_NO_DEVICE", -3: "HG_ERR_HIP", -4: "HG_ERR_CAPACITY",
          -5: "HG_ERR_UNSUPPORTED", -6: "HG_ERR_RANGE", -7: "HG_ERR_TIME"}

# Every symbol include/hg_mi355x.h declares.
SYMBOLS = [
    "hg_ctx_create", "hg_ctx_destroy", "hg_ctx_synchronize", "hg_ctx_set_option", "hg_ctx_get_option", "hg_ctx_stream", "hg_last_error",
    "hg_version", "hg_prof_enable", "hg_prof_reset", "hg_prof_read", "hg_grid_create", "hg_grid_destroy", "hg_grid_clear", "hg_grid_resolution", "hg_grid_params",
    "hg_grid_set_cells", "hg_grid_read_cells", "hg_grid_count", "hg_grid_export",
    "hg_grid_num_blocks", "hg_grid_window_status", "hg_grid_to_proto", "hg_grid_from_proto", "hg_grid_xray", "hg_grid_block_arrays", "hg_grid_import_blocks", "hg_grid_insert",
    "hg_grid_insert_batch", "hg_pyramid_insert", "hg_pyramid_insert_batch", "hg_grid_status",
    "hg_voxel_filter", "hg_adaptive_voxel_filter", "hg_filter_last_device",
    "hg_problem_create", "hg_problem_destroy", "hg_problem_reset", "hg_problem_add_pose",
    "hg_problem_set_pose", "hg_problem_get_pose", "hg_problem_set_velocity", "hg_problem_get_velocity",
    "hg_problem_add_odometry_block", "hg_problem_add_imu_block", "hg_problem_add_block",
    "hg_problem_add_unwarped_block", "hg_problem_set_block_width",
    "hg_problem_num_residuals", "hg_problem_num_columns", "hg_problem_evaluate",
    "hg_solver_default_opts", "hg_problem_solve", "hg_problem_solve_batch", "hg_problem_solve_batch_async", "hg_problem_solve_async", "hg_problem_fetch",
    "hg_register_scan", "hg_register_scan_mode", "hg_register_scan_batch", "hg_register_scan_sequence", "hg_match_evaluate", "hg_match_solve",
    "hg_pyramid_insert_unwarped", "hg_unwarp_range_data", "hg_unwarp_last_device", "hg_unwarp_status", "hg_register_scan_unwarped",
]


class InsertOpts(C.Structure):
    """hg_insert_opts; defaults = trajectory_builder_3d.lua:78-93 (high-resolution inserter)."""
    _fields_ = [("relative_truncation_distance", C.c_double),
                ("maximum_weight", C.c_double),
                ("num_free_space_voxels", C.c_int32),
                ("project_sdf_distance_to_scan_normal", C.c_int32),
                ("weight_function_epsilon", C.c_double),
                ("weight_function_sigma", C.c_double),
                ("min_range", C.c_double),
                ("max_range", C.c_double),
                ("insertion_ratio", C.c_double),
                ("normal_computation_method", C.c_int32),
                ("normal_computation_horizontal_stride", C.c_int32),
                ("normal_computation_vertical_stride", C.c_int32),
                ("reserved", C.c_int32)]

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


class TimedCloud(C.Structure):
    """hg_timed_cloud: one sensor::TimedPointCloudData of the accumulation."""
    _fields_ = [("time", C.c_int64), ("begin", C.c_uint64), ("count", C.c_uint64),
                ("origin", C.c_float * 3), ("reserved", C.c_float)]


class InsertStats(C.Structure):
    _fields_ = [("num_hits", C.c_uint64), ("num_updates", C.c_uint64),
                ("num_blocks", C.c_uint64), ("flags", C.c_uint64)]


class SolverOpts(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int32), ("jacobi_scaling", C.c_int32),
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
                ("num_iterations", C.c_int32), ("num_successful_steps", C.c_int32),
                ("num_unsuccessful_steps", C.c_int32), ("num_cost_evaluations", C.c_int32),
                ("num_jacobian_evaluations", C.c_int32), ("termination_type", C.c_int32),
                ("termination_reason", C.c_int32), ("reserved", C.c_int32)]


class HgError(RuntimeError):
    pass


def source_digest():
    """sha256[:16] over the library's sources (csrc/*.hip, csrc/*.h, csrc/Makefile, include/*.h): what a committed
    rocprof / PMC profile was taken on. bench.py quotes a profile's HBM traffic only when this still matches."""
    import glob
    import hashlib
    root = os.path.dirname(_HERE)
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")) +
                   [os.path.join(_HERE, "csrc", "Makefile")] + glob.glob(os.path.join(root, "include", "*.h")))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


_lib = None


def load():
    """Loads the library and sets argtypes. Raises if the HIP extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.7 / libhsa-runtime64.
    # Importing torch first makes this library bind to that copy (same SONAME) instead of
    # bringing /opt/rocm's runtime in beside it, which would leave the second one without GPUs.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise HgError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; "
                      "g.build()'` (or make -C hectorgrapher_amd/csrc). There is no CPU fallback."
                      % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, sz, i32, u32, f32, f64 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_float, C.c_double
    P = C.POINTER
    L.hg_ctx_create.argtypes = [i32, vp, P(vp)]
    L.hg_ctx_destroy.argtypes = [vp]
    L.hg_ctx_synchronize.argtypes = [vp]
    L.hg_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_longlong]
    L.hg_ctx_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_longlong)]
    L.hg_ctx_stream.restype = vp
    L.hg_ctx_stream.argtypes = [vp]
    L.hg_last_error.restype = C.c_char_p
    L.hg_version.restype = C.c_char_p
    L.hg_prof_enable.argtypes = [vp, i32]
    L.hg_prof_reset.argtypes = [vp]
    L.hg_prof_read.argtypes = [vp, i32, P(C.c_uint64), P(f64), P(C.c_uint64)]
    L.hg_grid_create.argtypes = [vp, f32, f32, f32, u32, P(vp)]
    L.hg_grid_destroy.argtypes = [vp]
    L.hg_grid_clear.argtypes = [vp]
    L.hg_grid_resolution.restype = f32
    L.hg_grid_resolution.argtypes = [vp]
    L.hg_grid_params.argtypes = [vp, P(f32), P(f32), P(f32), P(u32)]
    L.hg_grid_set_cells.argtypes = [vp, vp, sz, vp, vp]
    L.hg_grid_read_cells.argtypes = [vp, vp, sz, vp, vp]
    L.hg_grid_count.argtypes = [vp, P(sz)]
    L.hg_grid_export.argtypes = [vp, vp, vp, vp, sz, P(sz)]
    L.hg_grid_num_blocks.argtypes = [vp, P(u32)]
    L.hg_grid_window_status.argtypes = [vp, vp]
    L.hg_grid_to_proto.argtypes = [vp, vp, sz, P(sz)]
    L.hg_grid_from_proto.argtypes = [vp, vp, sz, u32, P(vp)]
    L.hg_grid_xray.argtypes = [vp, vp, vp, sz, P(i32), P(i32), vp, P(sz)]
    L.hg_grid_block_arrays.argtypes = [vp, P(vp), P(vp), P(u32)]
    L.hg_grid_import_blocks.argtypes = [vp, vp, vp, u32, i32]
    L.hg_grid_insert.argtypes = [vp, P(InsertOpts), vp, vp, sz, sz, vp, i32, i32, P(InsertStats)]
    L.hg_grid_insert_batch.argtypes = [vp, P(InsertOpts), vp, vp, vp, sz, sz, vp, i32, i32,
                                       P(InsertStats)]
    L.hg_pyramid_insert.argtypes = [vp, vp, i32, vp, vp, sz, sz, vp, i32, i32, vp]
    L.hg_pyramid_insert_batch.argtypes = [vp, vp, i32, vp, vp, vp, sz, sz, vp, i32, i32, vp]
    L.hg_grid_status.argtypes = [vp, P(InsertStats)]
    L.hg_voxel_filter.argtypes = [vp, f32, vp, sz, i32, i32, vp, P(sz)]
    L.hg_adaptive_voxel_filter.argtypes = [vp, f32, f32, f32, vp, sz, i32, i32, vp, P(sz)]
    L.hg_filter_last_device.argtypes = [vp, P(vp), P(vp), P(sz)]
    L.hg_problem_create.argtypes = [vp, P(vp)]
    L.hg_problem_destroy.argtypes = [vp]
    L.hg_problem_reset.argtypes = [vp]
    L.hg_problem_add_pose.argtypes = [vp, vp, i32]
    L.hg_problem_set_pose.argtypes = [vp, i32, vp]
    L.hg_problem_get_pose.argtypes = [vp, i32, vp]
    L.hg_problem_set_velocity.argtypes = [vp, i32, vp, i32]
    L.hg_problem_get_velocity.argtypes = [vp, i32, vp]
    L.hg_problem_add_odometry_block.argtypes = [vp, i32, i32, f64, f64, vp]
    L.hg_problem_add_imu_block.argtypes = [vp, i32, i32, f64, f64, f64, f64, vp]
    L.hg_problem_add_block.argtypes = [vp, vp, sz, i32, vp, i32, i32, f64, i32, i32, f64]
    L.hg_problem_add_unwarped_block.argtypes = [vp, vp, vp, sz, i32, vp, i32, i32, f64, i32, i32]
    L.hg_problem_set_block_width.argtypes = [vp, i32, sz]
    L.hg_problem_num_residuals.argtypes = [vp]
    L.hg_problem_num_columns.argtypes = [vp]
    L.hg_problem_evaluate.argtypes = [vp, vp, vp, vp, vp]
    L.hg_solver_default_opts.argtypes = [P(SolverOpts)]
    L.hg_problem_solve.argtypes = [vp, P(SolverOpts), P(SolverSummary)]
    L.hg_problem_solve_batch.argtypes = [vp, i32, P(SolverOpts), vp]
    L.hg_problem_solve_batch_async.argtypes = [vp, i32, P(SolverOpts)]
    L.hg_problem_solve_async.argtypes = [vp, P(SolverOpts)]
    L.hg_problem_fetch.argtypes = [vp, P(SolverSummary)]
    L.hg_register_scan.argtypes = [vp, P(SolverOpts), i32, vp, vp, i32, vp, vp, sz, sz, i32, vp,
                                   P(SolverSummary)]
    L.hg_register_scan_mode.argtypes = [vp, P(SolverOpts), i32, vp, vp, i32, vp, vp, sz, sz, i32, i32, vp,
                                        P(SolverSummary)]
    L.hg_register_scan_batch.argtypes = [vp, i32, P(SolverOpts), vp, vp, vp, i32, vp, vp, vp, sz, i32, vp, vp]
    L.hg_register_scan_sequence.argtypes = [vp, P(SolverOpts), vp, vp, i32, i32, vp, vp, vp, vp, sz, i32, i32, vp, i32, i32, vp, vp]
    L.hg_match_evaluate.argtypes = [vp, vp, i32, i32, vp, sz, i32, f64, vp, vp, f64, vp, vp, vp, vp]
    L.hg_match_solve.argtypes = [vp, vp, i32, i32, vp, sz, i32, f64, vp, vp, i32, f64,
                                 P(SolverOpts), P(SolverSummary)]
    L.hg_pyramid_insert_unwarped.argtypes = [vp, vp, i32, vp, sz, sz, i32, vp, i32, vp, vp, i32, vp, i32, vp]
    L.hg_unwarp_range_data.argtypes = [vp, vp, sz, i32, vp, i32, vp, vp, i32, i32, vp, vp, vp]
    L.hg_unwarp_last_device.argtypes = [vp, P(vp), P(vp), P(sz)]
    L.hg_unwarp_status.argtypes = [vp]
    L.hg_register_scan_unwarped.argtypes = [vp, P(SolverOpts), vp, vp, i32, vp, sz, sz, i32, vp, i32, vp, vp, i32,
                                            vp, i32, vp, P(SolverSummary)]
    for name in SYMBOLS:
        getattr(L, name)  # raises AttributeError if the ABI is incomplete
    _lib = L
    return L


def check(rc, what=""):
    if rc < 0:
        msg = load().hg_last_error().decode("utf-8", "replace")
        raise HgError("%s failed: %s (%d) %s" % (what, ERRORS.get(rc, "?"), rc, msg))
    return rc
