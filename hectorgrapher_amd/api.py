"""Host-side mirror of the reference interface for the TSDF hot path.

Class and method names follow cartographer::mapping (ref paths relative to
/root/reference/cartographer/):
  HybridGridTSDF            mapping/3d/hybrid_grid_tsdf.h:59-134
  RangeData                 sensor/range_data.h:44-57
  TSDFRangeDataInserter3D   mapping/3d/tsdf_range_data_inserter_3d.{h,cc} (Insert :395)
  Problem                   the ceres::Problem of optimizing_local_trajectory_builder.cc:1238-1291
  CeresScanMatcher3D        mapping/internal/3d/scan_matching/ceres_scan_matcher_3d.cc:72-118
Everything computes on the GPU through the C ABI (include/hg_mi355x.h); numpy
arrays are host buffers, torch CUDA tensors are passed as device pointers.
"""
import contextlib
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import InsertOpts, InsertStats, SolverOpts, SolverSummary, HgError, check  # noqa: F401


def _is_device(a):
    return hasattr(a, "data_ptr") and getattr(a, "is_cuda", False)


def _host(a, dtype, shape_last=None):
    a = np.ascontiguousarray(a, dtype)
    if shape_last is not None:
        a = a.reshape(-1, shape_last)
    return a


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One HIP stream on one device (hg_ctx)."""

    def __init__(self, device=0, stream=None):
        self._L = _lib.load()
        h = C.c_void_p()
        check(self._L.hg_ctx_create(int(device), stream, C.byref(h)), "hg_ctx_create")
        self._h = h
        self.device = int(device)
        self._children = weakref.WeakSet()  # grids / problems: destroyed before the context

    def synchronize(self):
        check(self._L.hg_ctx_synchronize(self._h), "hg_ctx_synchronize")

    def set_option(self, key, value):
        """A tuning / diagnostic switch of the context (hg_ctx_set_option; keys: include/hg_mi355x.h)."""
        check(self._L.hg_ctx_set_option(self._h, key.encode(), int(value)), "hg_ctx_set_option")

    def get_option(self, key):
        v = C.c_longlong()
        check(self._L.hg_ctx_get_option(self._h, key.encode(), C.byref(v)), "hg_ctx_get_option")
        return v.value

    @contextlib.contextmanager
    def option(self, key, value):
        """`with ctx.option("lm_band", 1): ...` -- the switch set inside the block, restored behind it."""
        old = self.get_option(key)
        self.set_option(key, value)
        try:
            yield self
        finally:
            self.set_option(key, old)

    def prof_enable(self, on=True):
        check(self._L.hg_prof_enable(self._h, int(on)), "hg_prof_enable")

    def prof_reset(self):
        check(self._L.hg_prof_reset(self._h), "hg_prof_reset")

    def prof_read(self):
        """{kernel: (launches, total_ms, units)} measured with HIP events on this stream."""
        out = {}
        for name, k in _lib.KERNELS.items():
            n, ms, u = C.c_uint64(), C.c_double(), C.c_uint64()
            check(self._L.hg_prof_read(self._h, k, C.byref(n), C.byref(ms), C.byref(u)),
                  "hg_prof_read")
            out[name] = (n.value, ms.value, u.value)
        return out

    @property
    def stream(self):
        return self._L.hg_ctx_stream(self._h)

    def close(self):
        if getattr(self, "_h", None):
            for child in list(self._children):
                child.close()
            self._L.hg_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RangeData:
    """sensor::RangeData: origin + returns (+ width of the structured cloud)."""

    def __init__(self, origin, returns, width=0):
        self.origin = np.ascontiguousarray(origin, np.float32).reshape(3)
        self.returns = returns
        self.width = int(width)


class HybridGridTSDF:
    def __init__(self, ctx, resolution, relative_truncation_distance=2.5, max_weight=1000.0,
                 max_blocks=1 << 16):
        self._L = _lib.load()
        self.ctx = ctx
        h = C.c_void_p()
        check(self._L.hg_grid_create(ctx._h, resolution, relative_truncation_distance, max_weight,
                                     int(max_blocks), C.byref(h)), "hg_grid_create")
        self._h = h
        ctx._children.add(self)
        self._resolution = np.float32(resolution)
        # codec constants in float32, as the device computes them (tsd_value_converter.cc:22-32)
        f = np.float32
        self.max_tsd = f(relative_truncation_distance) * f(resolution)
        self.min_tsd = -self.max_tsd
        self.max_weight = f(max_weight)
        self.max_blocks = int(max_blocks)
        self.relative_truncation_distance = float(relative_truncation_distance)

    def resolution(self):
        return self._resolution

    def clear(self):
        check(self._L.hg_grid_clear(self._h), "hg_grid_clear")

    def close(self):
        if getattr(self, "_h", None):
            self._L.hg_grid_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- HybridGridBase::GetCellIndex / GetCenterOfCell (hybrid_grid_base.h:428-446) --
    def GetCellIndex(self, point):
        p = np.asarray(point, np.float32) / self._resolution
        # std::lround: half away from zero
        return (np.sign(p) * np.floor(np.abs(p) + np.float32(0.5))).astype(np.int32)

    def GetCenterOfCell(self, index):
        return np.asarray(index, np.int32).astype(np.float32) * self._resolution

    # -- SetCell / value access --
    def SetCell(self, index, tsd, weight):
        self.set_cells(np.asarray(index, np.int32).reshape(1, 3), [tsd], [weight])

    def set_cells(self, ijk, tsd, weight):
        ijk = _host(ijk, np.int32, 3)
        tsd = _host(tsd, np.float32)
        weight = _host(weight, np.float32)
        check(self._L.hg_grid_set_cells(self._h, _p(ijk), len(ijk), _p(tsd), _p(weight)),
              "hg_grid_set_cells")

    def read_cells(self, ijk):
        """Raw (discrete_tsd, discrete_weight) codes of TSDFVoxel."""
        ijk = _host(ijk, np.int32, 3)
        t = np.empty(len(ijk), np.uint16)
        w = np.empty(len(ijk), np.uint16)
        check(self._L.hg_grid_read_cells(self._h, _p(ijk), len(ijk), _p(t), _p(w)),
              "hg_grid_read_cells")
        return t, w

    def _decode(self, code, lower, upper, unknown):
        f = np.float32
        scale = (f(upper) - f(lower)) / f(32766.0)
        v = (code & 0x7FFF).astype(np.float32)
        out = v * scale + (f(lower) - scale)
        return np.where((code & 0x7FFF) == 0, f(unknown), out).astype(np.float32)

    def GetTSD(self, ijk):
        t, _ = self.read_cells(ijk)
        return self._decode(t, self.min_tsd, self.max_tsd, self.min_tsd)

    def GetWeight(self, ijk):
        _, w = self.read_cells(ijk)
        return self._decode(w, 0.0, self.max_weight, 0.0)

    def IsKnown(self, ijk):
        _, w = self.read_cells(ijk)
        return w != 0

    def status(self):
        """Synchronises; returns the last insert's counters and raises on sticky error flags."""
        st = InsertStats()
        check(self._L.hg_grid_status(self._h, C.byref(st)), "hg_grid_status")
        return st

    def count(self):
        n = C.c_size_t()
        check(self._L.hg_grid_count(self._h, C.byref(n)), "hg_grid_count")
        return n.value

    def num_blocks(self):
        n = C.c_uint32()
        check(self._L.hg_grid_num_blocks(self._h, C.byref(n)), "hg_grid_num_blocks")
        return n.value

    def window_status(self):
        """hg_grid_window_status as a dict: blocks, overflow_blocks, window (x, y, z), extent (x, y, z), direct."""
        out = (C.c_uint32 * 9)()
        check(self._L.hg_grid_window_status(self._h, out), "hg_grid_window_status")
        return {"blocks": out[0], "overflow_blocks": out[1], "window": tuple(out[2:5]),
                "extent": tuple(out[5:8]), "direct": bool(out[8])}

    def export(self):
        """(ijk, tsd codes, weight codes) in the reference iterator order (ToProto order)."""
        n = self.count()
        ijk = np.empty((n, 3), np.int32)
        t = np.empty(n, np.uint16)
        w = np.empty(n, np.uint16)
        got = C.c_size_t()
        check(self._L.hg_grid_export(self._h, _p(ijk), _p(t), _p(w), n, C.byref(got)),
              "hg_grid_export")
        return ijk, t, w

    def ToProto(self):
        """Serialised proto::HybridGridTSDF (bytes), as HybridGridTSDF::ToProto().SerializeAsString()."""
        n = C.c_size_t()
        check(self._L.hg_grid_to_proto(self._h, None, 0, C.byref(n)), "hg_grid_to_proto")
        buf = (C.c_uint8 * max(1, n.value))()
        check(self._L.hg_grid_to_proto(self._h, buf, n.value, C.byref(n)), "hg_grid_to_proto")
        return bytes(buf[:n.value])

    def xray(self, global_submap_pose):
        """AddToTextureProto(HybridGridTSDF) (submap_3d.cc:245-276) without the gzip: returns
        (cells uint8 [height, width, 2] = value, alpha; max_index xy int32[2])."""
        pose = np.ascontiguousarray(global_submap_pose, np.float64)
        assert pose.shape == (7,)
        w, h, n = C.c_int32(), C.c_int32(), C.c_size_t()
        mx = np.zeros(2, np.int32)
        check(self._L.hg_grid_xray(self._h, _p(pose), None, 0, C.byref(w), C.byref(h), _p(mx), C.byref(n)),
              "hg_grid_xray")
        cells = np.zeros(n.value, np.uint8)
        if n.value:
            check(self._L.hg_grid_xray(self._h, _p(pose), _p(cells), n.value, C.byref(w), C.byref(h), _p(mx),
                                       C.byref(n)), "hg_grid_xray")
        return cells.reshape(h.value, w.value, 2), mx

    @classmethod
    def FromProto(cls, ctx, data, max_blocks=1 << 16):
        """HybridGridTSDF(const proto::HybridGridTSDF&)."""
        L = _lib.load()
        h = C.c_void_p()
        arr = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data if data else b"\0")
        check(L.hg_grid_from_proto(ctx._h, arr, len(data), int(max_blocks), C.byref(h)), "hg_grid_from_proto")
        g = cls.__new__(cls)
        g._L, g.ctx, g._h = L, ctx, h
        ctx._children.add(g)
        g._resolution = np.float32(L.hg_grid_resolution(h))
        g.max_blocks = int(max_blocks)
        # the proto constructor builds the grid from (resolution, field 8, field 9)
        # (hybrid_grid_tsdf.h:69-83): read the converter constants back from the device grid
        mt, mw = C.c_float(), C.c_float()
        check(L.hg_grid_params(h, None, C.byref(mt), C.byref(mw), None), "hg_grid_params")
        g.max_tsd = np.float32(mt.value)
        g.min_tsd = -g.max_tsd
        g.max_weight = np.float32(mw.value)
        g.relative_truncation_distance = float(g.max_tsd / g._resolution) if g._resolution > 0 else 0.0
        return g

    def block_arrays(self):
        """Device pointers (keys u64[nb], voxels u32[nb*512]) and nb, for the multi-GPU gather."""
        k, v, n = C.c_void_p(), C.c_void_p(), C.c_uint32()
        check(self._L.hg_grid_block_arrays(self._h, C.byref(k), C.byref(v), C.byref(n)),
              "hg_grid_block_arrays")
        return k.value, v.value, n.value

    def import_blocks(self, keys, voxels, num_blocks=None):
        if _is_device(keys):
            nb = int(num_blocks if num_blocks is not None else keys.numel())
            check(self._L.hg_grid_import_blocks(self._h, keys.data_ptr(), voxels.data_ptr(), nb,
                                                _lib.HG_DEVICE), "hg_grid_import_blocks")
        else:
            keys = _host(keys, np.uint64)
            voxels = _host(voxels, np.uint32)
            check(self._L.hg_grid_import_blocks(self._h, _p(keys), _p(voxels), len(keys),
                                                _lib.HG_HOST), "hg_grid_import_blocks")


class TSDFRangeDataInserter3D:
    """RangeDataInserterInterface implementation for TSDF grids."""

    def __init__(self, options=None, mode=_lib.HG_INSERT_EXACT):
        self._L = _lib.load()
        self.options = options or InsertOpts()
        self.mode = mode
        self.last_stats = InsertStats()

    def RequiresStructuredData(self):
        return bool(self.options.project_sdf_distance_to_scan_normal)

    def Insert(self, range_data, grid, pose_tq=None):
        """Insert(range_data, grid). `pose_tq` optionally applies Submap3D::InsertData's
        local_pose().inverse().cast<float>() (submap_3d.cc:436-437) on the device."""
        origin = range_data.origin
        pose = None if pose_tq is None else np.ascontiguousarray(pose_tq, np.float32)
        r = range_data.returns
        if _is_device(r):
            n = r.shape[0]
            ptr, space = r.data_ptr(), _lib.HG_DEVICE
        else:
            r = _host(r, np.float32, 3)
            n = len(r)
            ptr, space = _p(r), _lib.HG_HOST
        st = InsertStats()
        check(self._L.hg_grid_insert(grid._h, C.byref(self.options), _p(origin), ptr, n,
                                     range_data.width, _p(pose), self.mode, space, C.byref(st)),
              "hg_grid_insert")
        self.last_stats = st
        return st

    def InsertBatch(self, origins, returns, scan_offsets, grid, width=0, poses_tq=None):
        """Stream form: scans applied in order in one call (hg_grid_insert_batch)."""
        origins = _host(origins, np.float32, 3)
        offs = _host(scan_offsets, np.uint64)
        poses = None if poses_tq is None else _host(poses_tq, np.float32, 7)
        if _is_device(returns):
            ptr, space = returns.data_ptr(), _lib.HG_DEVICE
        else:
            returns = _host(returns, np.float32, 3)
            ptr, space = _p(returns), _lib.HG_HOST
        st = InsertStats()
        check(self._L.hg_grid_insert_batch(grid._h, C.byref(self.options), _p(origins), ptr,
                                           _p(offs), len(offs) - 1, int(width), _p(poses),
                                           self.mode, space, C.byref(st)), "hg_grid_insert_batch")
        self.last_stats = st
        return st


class VoxelFilter:
    """sensor::VoxelFilter(size).Filter: first point of every voxel, input order preserved."""

    def __init__(self, ctx, size):
        self.ctx, self.size = ctx, float(size)

    def Filter(self, point_cloud):
        """Returns the indices of the kept points (numpy uint32, ascending)."""
        L = _lib.load()
        n = C.c_size_t()
        if _is_device(point_cloud):
            m, stride = point_cloud.shape
            out = np.empty(m, np.uint32)
            check(L.hg_voxel_filter(self.ctx._h, self.size, point_cloud.data_ptr(), m, stride,
                                    _lib.HG_DEVICE, _p(out), C.byref(n)), "hg_voxel_filter")
        else:
            pts = np.ascontiguousarray(point_cloud, np.float32)
            m, stride = pts.shape
            out = np.empty(m, np.uint32)
            check(L.hg_voxel_filter(self.ctx._h, self.size, _p(pts), m, stride, _lib.HG_HOST, _p(out),
                                    C.byref(n)), "hg_voxel_filter")
        return out[:n.value].copy()


class AdaptiveVoxelFilter:
    """sensor::AdaptiveVoxelFilter (max_length, min_num_points, max_range)."""

    def __init__(self, ctx, max_length, min_num_points, max_range):
        self.ctx = ctx
        self.max_length, self.min_num_points, self.max_range = float(max_length), float(min_num_points), float(max_range)

    def Filter(self, point_cloud):
        L = _lib.load()
        n = C.c_size_t()
        if _is_device(point_cloud):
            m, stride = point_cloud.shape
            ptr, space = point_cloud.data_ptr(), _lib.HG_DEVICE
        else:
            point_cloud = np.ascontiguousarray(point_cloud, np.float32)
            m, stride = point_cloud.shape
            ptr, space = _p(point_cloud), _lib.HG_HOST
        out = np.empty(m, np.uint32)
        check(L.hg_adaptive_voxel_filter(self.ctx._h, self.max_length, self.min_num_points,
                                         self.max_range, ptr, m, stride, space, _p(out), C.byref(n)),
              "hg_adaptive_voxel_filter")
        return out[:n.value].copy()


def insert_pyramid(inserters, range_data, grids, pose_tq=None, want_stats=True):
    """Submap3D::InsertData shape (submap_3d.cc:427-452): the same range data goes through
    inserters[l] into grids[l] (high resolution, low resolution, ...) in one fused device pass.
    Returns a list of InsertStats (or None when want_stats is False: fully asynchronous)."""
    L = _lib.load()
    n_l = len(grids)
    opts = (InsertOpts * n_l)(*[i.options for i in inserters])
    garr = (C.c_void_p * n_l)(*[g._h for g in grids])
    origin = range_data.origin
    pose = None if pose_tq is None else np.ascontiguousarray(pose_tq, np.float32)
    r = range_data.returns
    if _is_device(r):
        n, ptr, space = r.shape[0], r.data_ptr(), _lib.HG_DEVICE
    else:
        r = _host(r, np.float32, 3)
        n, ptr, space = len(r), _p(r), _lib.HG_HOST
    st = (InsertStats * n_l)() if want_stats else None
    check(L.hg_pyramid_insert(garr, opts, n_l, _p(origin), ptr, n, range_data.width, _p(pose),
                              inserters[0].mode, space, st), "hg_pyramid_insert")
    return list(st) if want_stats else None


def _timed_clouds(clouds):
    """[(time_ticks, origin[3], points[n, 4])] -> (hg_timed_cloud array, concatenated points, n, memspace).
    A single cloud may hold a CUDA tensor (passed as a device pointer); several clouds are concatenated on
    the host."""
    arr = (_lib.TimedCloud * len(clouds))()
    begin = 0
    for k, (t, origin, pts) in enumerate(clouds):
        arr[k].time = int(t)
        arr[k].begin = begin
        arr[k].count = int(pts.shape[0])
        o = np.asarray(origin, np.float32).reshape(3)
        arr[k].origin[0], arr[k].origin[1], arr[k].origin[2] = float(o[0]), float(o[1]), float(o[2])
        begin += int(pts.shape[0])
    if len(clouds) == 1 and _is_device(clouds[0][2]):
        p = clouds[0][2]
        # the kernels read float4 rows straight from this pointer: anything else must be converted first
        if p.dim() != 2 or p.shape[1] != 4:
            raise ValueError("timed cloud on the device must be [n, 4] (x y z time), got %s" % (tuple(p.shape),))
        import torch
        if p.dtype != torch.float32 or not p.is_contiguous():
            # the conversion runs on torch's current stream, the library's kernels on the context's own: the copy must
            # be complete before its pointer is handed over (read-before-write race)
            p = p.to(torch.float32).contiguous()
            torch.cuda.current_stream(p.device).synchronize()
        return arr, p, p.data_ptr(), begin, _lib.HG_DEVICE
    allp = np.ascontiguousarray(np.concatenate([_host(c[2], np.float32, 4) for c in clouds], 0))
    return arr, allp, allp.ctypes.data, begin, _lib.HG_HOST


def insert_pyramid_unwarped(inserters, clouds, width, control_times, control_poses, grids, pose_tq=None,
                            want_stats=True):
    """hg_pyramid_insert_unwarped: per-point unwarping of the accumulated range data
    (optimizing_local_trajectory_builder.cc:1331-1379), the frame changes of AddAccumulatedRangeData /
    Submap3D::InsertData and the insertion, all on the device. clouds: [(time_ticks, origin[3],
    points[n, 4] = x y z time)]; control_poses[0] is optimized_pose."""
    L = _lib.load()
    n_l = len(grids)
    opts = (InsertOpts * n_l)(*[i.options for i in inserters])
    garr = (C.c_void_p * n_l)(*[g._h for g in grids])
    arr, keep, ptr, n, space = _timed_clouds(clouds)
    ct = np.ascontiguousarray(control_times, np.int64)
    cp = np.ascontiguousarray(control_poses, np.float64).reshape(-1, 7)
    pose = None if pose_tq is None else np.ascontiguousarray(pose_tq, np.float32)
    st = (InsertStats * n_l)() if want_stats else None
    check(L.hg_pyramid_insert_unwarped(garr, opts, n_l, ptr, n, int(width), space, arr, len(clouds), _p(cp), _p(ct),
                                       len(ct), _p(pose), inserters[0].mode, st), "hg_pyramid_insert_unwarped")
    if want_stats:
        del keep        # (the stats read-back has synchronised the stream)
    else:
        grids[0]._keep_unwarp = keep   # the enqueued kernels still read it: held until the next call replaces it
    return list(st) if want_stats else None


def unwarp_range_data(ctx, clouds, control_times, control_poses, frame=0, pose_tq=None):
    """hg_unwarp_range_data: (xyz [n, 3], origin [3]) of the accumulated range data in the tracking frame
    (frame 0) or moved on by control_poses[0].cast<float>() and pose_tq (frame 1)."""
    L = _lib.load()
    arr, keep, ptr, n, space = _timed_clouds(clouds)
    ct = np.ascontiguousarray(control_times, np.int64)
    cp = np.ascontiguousarray(control_poses, np.float64).reshape(-1, 7)
    pose = None if pose_tq is None else np.ascontiguousarray(pose_tq, np.float32)
    xyz = np.empty((n, 3), np.float32)
    origin = np.empty(3, np.float32)
    check(L.hg_unwarp_range_data(ctx._h, ptr, n, space, arr, len(clouds), _p(cp), _p(ct), len(ct), int(frame),
                                 _p(pose), _p(xyz), _p(origin)), "hg_unwarp_range_data")
    del keep
    return xyz, origin


def unwarp_range_data_async(ctx, clouds, control_times, control_poses, frame=0, pose_tq=None):
    """hg_unwarp_range_data without host outputs: the result stays on the device (hg_unwarp_last_device);
    `unwarp_status(ctx)` waits for it and raises HG_ERR_TIME like the synchronous form."""
    L = _lib.load()
    arr, keep, ptr, n, space = _timed_clouds(clouds)
    ct = np.ascontiguousarray(control_times, np.int64)
    cp = np.ascontiguousarray(control_poses, np.float64).reshape(-1, 7)
    pose = None if pose_tq is None else np.ascontiguousarray(pose_tq, np.float32)
    check(L.hg_unwarp_range_data(ctx._h, ptr, n, space, arr, len(clouds), _p(cp), _p(ct), len(ct), int(frame),
                                 _p(pose), None, None), "hg_unwarp_range_data")
    ctx._keep_unwarp = keep  # a device input must outlive the enqueued kernels
    return n


def unwarp_status(ctx):
    """hg_unwarp_status: waits for the context's stream; raises HgError(HG_ERR_TIME) when the last unwarp call
    met a return outside the control points."""
    check(_lib.load().hg_unwarp_status(ctx._h), "hg_unwarp_status")


def unwarp_last_device(ctx):
    """(device pointer of xyz [n, 3], device pointer of the origin, n) of the context's last unwarp call."""
    xyz, org, cnt = C.c_void_p(), C.c_void_p(), C.c_size_t()
    check(_lib.load().hg_unwarp_last_device(ctx._h, C.byref(xyz), C.byref(org), C.byref(cnt)), "hg_unwarp_last_device")
    return xyz.value, org.value, cnt.value


def register_scan_unwarped(problem, inserters, clouds, width, pose_index, control_times, grids, pose_tq=None,
                           **solver_kw):
    """hg_register_scan_unwarped: solve the window, then insert `clouds` unwarped with the SOLVED control poses
    (problem pose pose_index[k] at control_times[k]) without a host round trip. Returns (poses [K, 7], summary)."""
    L = _lib.load()
    o = SolverOpts()
    L.hg_solver_default_opts(C.byref(o))
    for k, v in solver_kw.items():
        setattr(o, k, v)
    n_l = len(grids)
    opts = (InsertOpts * n_l)(*[i.options for i in inserters])
    garr = (C.c_void_p * n_l)(*[g._h for g in grids])
    arr, keep, ptr, n, space = _timed_clouds(clouds)
    ct = np.ascontiguousarray(control_times, np.int64)
    idx = np.ascontiguousarray(pose_index, np.int32)
    pose = None if pose_tq is None else np.ascontiguousarray(pose_tq, np.float32)
    poses = np.empty((len(ct), 7), np.float64)
    summ = SolverSummary()
    check(L.hg_register_scan_unwarped(problem._h, C.byref(o), garr, opts, n_l, ptr, n, int(width), space, arr,
                                      len(clouds), _p(idx), _p(ct), len(ct), _p(pose), inserters[0].mode, _p(poses),
                                      C.byref(summ)), "hg_register_scan_unwarped")
    del keep
    return poses, summ


class Problem:
    """ceres::Problem restricted to TSDF space cost functions over pose blocks."""

    def __init__(self, ctx):
        self._L = _lib.load()
        self.ctx = ctx
        h = C.c_void_p()
        check(self._L.hg_problem_create(ctx._h, C.byref(h)), "hg_problem_create")
        self._h = h
        ctx._children.add(self)
        self._keep = []

    def close(self):
        if getattr(self, "_h", None):
            self._L.hg_problem_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        check(self._L.hg_problem_reset(self._h), "hg_problem_reset")
        self._keep = []

    def add_pose(self, tq, constant=False):
        tq = _host(tq, np.float64)
        return check(self._L.hg_problem_add_pose(self._h, _p(tq), int(constant)), "add_pose")

    def set_pose(self, idx, tq):
        tq = _host(tq, np.float64)
        check(self._L.hg_problem_set_pose(self._h, idx, _p(tq)), "set_pose")

    def get_pose(self, idx):
        out = np.empty(7, np.float64)
        check(self._L.hg_problem_get_pose(self._h, idx, _p(out)), "get_pose")
        return out

    def set_velocity(self, idx, v, constant=False):
        v = _host(v, np.float64)
        check(self._L.hg_problem_set_velocity(self._h, idx, _p(v), int(constant)), "set_velocity")

    def get_velocity(self, idx):
        out = np.empty(3, np.float64)
        check(self._L.hg_problem_get_velocity(self._h, idx, _p(out)), "get_velocity")
        return out

    def add_odometry_block(self, a, b, translation_weight, rotation_weight, delta_tq):
        d = _host(delta_tq, np.float64)
        return check(self._L.hg_problem_add_odometry_block(self._h, a, b, float(translation_weight),
                                                           float(rotation_weight), _p(d)), "add_odometry_block")

    def add_imu_block(self, a, b, translation_weight, velocity_weight, rotation_weight, dt, delta_q):
        d = _host(delta_q, np.float64)
        return check(self._L.hg_problem_add_imu_block(self._h, a, b, float(translation_weight),
                                                      float(velocity_weight), float(rotation_weight),
                                                      float(dt), _p(d)), "add_imu_block")

    def add_block(self, xyz, grids, scaling_factor, pose_a, pose_b=-1, interpolation_ratio=0.0,
                  multi_res=False, width=0):
        """width > 0: the cloud is a structured scan with `width` returns per column (hg_problem_set_block_width:
        a performance hint, see include/hg_mi355x.h)."""
        arr = (C.c_void_p * len(grids))(*[g._h for g in grids])
        if _is_device(xyz):
            n, ptr, space = xyz.shape[0], xyz.data_ptr(), _lib.HG_DEVICE
            self._keep.append(xyz)
        else:
            xyz = _host(xyz, np.float32, 3)
            n, ptr, space = len(xyz), _p(xyz), _lib.HG_HOST
        self._keep.append(grids)
        b = check(self._L.hg_problem_add_block(self._h, ptr, n, space, arr, len(grids),
                                               int(multi_res), float(scaling_factor), pose_a,
                                               pose_b, float(interpolation_ratio)), "add_block")
        if width:
            check(self._L.hg_problem_set_block_width(self._h, b, int(width)), "set_block_width")
        return b

    def set_block_width(self, block, width):
        check(self._L.hg_problem_set_block_width(self._h, int(block), int(width)), "set_block_width")

    def add_unwarped_block(self, xyz, interpolation_ratios, grids, scaling_factor, pose_a, pose_b,
                           multi_res=False):
        """All returns bracketed by control points (pose_a, pose_b), each with its own ratio."""
        arr = (C.c_void_p * len(grids))(*[g._h for g in grids])
        if _is_device(xyz):
            if not _is_device(interpolation_ratios):
                raise HgError("device points need device interpolation ratios (float64)")
            n, ptr, fptr, space = xyz.shape[0], xyz.data_ptr(), interpolation_ratios.data_ptr(), _lib.HG_DEVICE
            self._keep.append((xyz, interpolation_ratios))
        else:
            xyz = _host(xyz, np.float32, 3)
            f = _host(interpolation_ratios, np.float64)
            if len(f) != len(xyz):
                raise HgError("one interpolation ratio per return expected")
            n, ptr, fptr, space = len(xyz), _p(xyz), _p(f), _lib.HG_HOST
        self._keep.append(grids)
        return check(self._L.hg_problem_add_unwarped_block(self._h, ptr, fptr, n, space, arr, len(grids),
                                                           int(multi_res), float(scaling_factor),
                                                           pose_a, pose_b), "add_unwarped_block")

    def num_residuals(self):
        return check(self._L.hg_problem_num_residuals(self._h))

    def num_columns(self):
        return check(self._L.hg_problem_num_columns(self._h))

    def evaluate(self, want_residuals=True):
        """Returns (cost, residuals, gradient, JtJ) in the local parameterisation."""
        n, c = self.num_residuals(), self.num_columns()
        cost = C.c_double()
        r = np.empty(n, np.float64) if want_residuals else None
        g = np.empty(c, np.float64)
        H = np.empty((c, c), np.float64)
        check(self._L.hg_problem_evaluate(self._h, C.byref(cost), _p(r), _p(g), _p(H)), "evaluate")
        return cost.value, r, g, H

    def solve(self, **kw):
        o = SolverOpts()
        self._L.hg_solver_default_opts(C.byref(o))
        for k, v in kw.items():
            setattr(o, k, v)
        s = SolverSummary()
        check(self._L.hg_problem_solve(self._h, C.byref(o), C.byref(s)), "solve")
        return s


def solve_batch(problems, **kw):
    """hg_problem_solve_batch: independent problems of one context solved with shared kernel launches
    (the many scan-to-submap matches of a constraint search). Returns the list of summaries."""
    L = _lib.load()
    o = SolverOpts()
    L.hg_solver_default_opts(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    n = len(problems)
    arr = (C.c_void_p * n)(*[p._h for p in problems])
    summ = (SolverSummary * n)()
    check(L.hg_problem_solve_batch(arr, n, C.byref(o), summ), "hg_problem_solve_batch")
    return list(summ)


def solve_batch_async(problems, **kw):
    """hg_problem_solve_batch_async: the enqueue-only form of solve_batch. The host is free while the batch runs
    (build and enqueue the next batch on another set of problems); collect with fetch_batch(problems)."""
    L = _lib.load()
    o = SolverOpts()
    L.hg_solver_default_opts(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    n = len(problems)
    arr = (C.c_void_p * n)(*[p._h for p in problems])
    check(L.hg_problem_solve_batch_async(arr, n, C.byref(o)), "hg_problem_solve_batch_async")


def fetch_batch(problems):
    """hg_problem_fetch on every problem of a batch enqueued by solve_batch_async. Returns the summaries."""
    L = _lib.load()
    summ = (SolverSummary * len(problems))()
    for i, p in enumerate(problems):
        check(L.hg_problem_fetch(p._h, C.byref(summ[i])), "hg_problem_fetch")
    return list(summ)


def register_scan(problem, pose_index, inserters, range_data, grids, **solver_kw):
    """One registration step on the device (hg_register_scan_mode): solve `problem`, then insert
    `range_data` (tracking frame) into `grids` at the solved pose, in the mode of inserters[0]
    (HG_INSERT_EXACT unless chosen otherwise). Returns (pose, summary)."""
    L = _lib.load()
    o = SolverOpts()
    L.hg_solver_default_opts(C.byref(o))
    for k, v in solver_kw.items():
        setattr(o, k, v)
    n_l = len(grids)
    opts = (InsertOpts * n_l)(*[i.options for i in inserters])
    garr = (C.c_void_p * n_l)(*[g._h for g in grids])
    r = range_data.returns
    if _is_device(r):
        n, ptr, space = r.shape[0], r.data_ptr(), _lib.HG_DEVICE
    else:
        r = _host(r, np.float32, 3)
        n, ptr, space = len(r), _p(r), _lib.HG_HOST
    pose = np.empty(7, np.float64)
    s = SolverSummary()
    check(L.hg_register_scan_mode(problem._h, C.byref(o), pose_index, garr, opts, n_l,
                                  _p(range_data.origin), ptr, n, range_data.width, space, int(inserters[0].mode),
                                  _p(pose), C.byref(s)), "hg_register_scan")
    return pose, s


def register_scan_batch(problems, pose_indices, inserters, scans, pyramids, **solver_kw):
    """hg_register_scan_batch: one registration step for each of len(problems) independent submaps
    with shared launches. scans[j] = RangeData (device tensor returns) of submap j, pyramids[j] its
    list of grids; inserters[l] carries the options of level l. Returns (poses [count, 7], summaries)."""
    L = _lib.load()
    o = SolverOpts()
    L.hg_solver_default_opts(C.byref(o))
    for k, v in solver_kw.items():
        setattr(o, k, v)
    count, n_l = len(problems), len(pyramids[0])
    parr = (C.c_void_p * count)(*[p._h for p in problems])
    garr = (C.c_void_p * (count * n_l))(*[g._h for pyr in pyramids for g in pyr])
    opts = (InsertOpts * n_l)(*[i.options for i in inserters])
    idx = (C.c_int * count)(*[int(i) for i in pose_indices])
    origins = np.ascontiguousarray([s_.origin for s_ in scans], np.float32)
    device = all(_is_device(s_.returns) for s_ in scans)
    keep = [s_.returns if device else _host(s_.returns, np.float32, 3) for s_ in scans]
    ptrs = (C.c_void_p * count)(*[(r.data_ptr() if device else r.ctypes.data) for r in keep])
    ns = (C.c_size_t * count)(*[int(r.shape[0]) for r in keep])
    poses = np.empty((count, 7), np.float64)
    summ = (SolverSummary * count)()
    check(L.hg_register_scan_batch(parr, count, C.byref(o), idx, garr, opts, n_l, _p(origins), ptrs, ns,
                                   int(scans[0].width), _lib.HG_DEVICE if device else _lib.HG_HOST, _p(poses), summ),
          "hg_register_scan_batch")
    return poses, list(summ)


def register_scan_sequence(problem, inserters, scans, guesses, pyramid, scaling, multi_res=True, prof_every=0,
                           prepare_only=False, **solver_kw):
    """hg_register_scan_sequence: len(scans) registration steps of ONE trajectory in one call (step k:
    one free pose = guesses[k], one block over scans[k] with scaling[k], solve, insert at the solved
    pose). Returns (poses [count, 7], summaries). The per-step interpreter work of calling
    register_scan in a loop stays out of the way. prepare_only=True marshals the arguments and returns a
    function that makes the call (a benchmark keeps the marshalling out of its timed region)."""
    L = _lib.load()
    o = SolverOpts()
    L.hg_solver_default_opts(C.byref(o))
    for k, v in solver_kw.items():
        setattr(o, k, v)
    count, n_l = len(scans), len(pyramid)
    garr = (C.c_void_p * n_l)(*[g._h for g in pyramid])
    opts = (InsertOpts * n_l)(*[i.options for i in inserters])
    origins = np.ascontiguousarray([s_.origin for s_ in scans], np.float32).reshape(-1, 3)
    device = all(_is_device(s_.returns) for s_ in scans)
    keep = [s_.returns if device else _host(s_.returns, np.float32, 3) for s_ in scans]
    ptrs = (C.c_void_p * max(1, count))(*[(r.data_ptr() if device else r.ctypes.data) for r in keep])
    ns = (C.c_size_t * max(1, count))(*[int(r.shape[0]) for r in keep])
    sc = np.ascontiguousarray(np.broadcast_to(np.asarray(scaling, np.float64), (count,)))
    gs = np.ascontiguousarray(guesses, np.float64).reshape(-1, 7)
    poses = np.empty((count, 7), np.float64)
    summ = (SolverSummary * max(1, count))()
    mode = inserters[0].mode if hasattr(inserters[0], "mode") else _lib.HG_INSERT_EXACT
    args = (problem._h, C.byref(o), garr, opts, n_l, 1 if multi_res else 0, _p(origins), ptrs, ns, _p(sc),
            int(scans[0].width) if count else 0, _lib.HG_DEVICE if device else _lib.HG_HOST, int(mode), _p(gs), count,
            int(prof_every), _p(poses), summ)
    keep_alive = (o, garr, opts, origins, keep, ptrs, ns, sc, gs)

    def run():
        """The library call alone (arguments are marshalled already); returns (poses, summaries)."""
        check(L.hg_register_scan_sequence(*args), "hg_register_scan_sequence")
        return poses, list(summ)[:count], keep_alive

    if prepare_only:
        return run
    return run()[:2]


def from_seconds(seconds):
    """common::FromSeconds (common/time.cc:30-33): 100 ns ticks, truncated toward zero."""
    return int(float(seconds) * 1e7)


def to_seconds(ticks):
    """common::ToSeconds (common/time.cc:35-38)."""
    return float(ticks) / 1e7


def per_point_subdivisions(point_times, cloud_time, control_times, num_points_per_subdivision):
    """Bracketing of AddPerPointMatchingResiduals (oltb.cc:521-565): subdivisions of
    `num_points_per_subdivision` consecutive returns, timed at the mean of their first and last
    return; a subdivision outside (front, back) of the control points is omitted. Times are
    universal 100 ns ticks (cloud_time, control_times) and seconds relative to the cloud
    (point_times). Returns [(start, end_exclusive, prev_index, next_index, ratio)]."""
    ct = [int(t) for t in control_times]
    n = len(point_times)
    out = []
    step = int(num_points_per_subdivision)
    for start in range(0, n, step):
        end = min(start + step - 1, n - 1)
        # TimedRangefinderPoint::time is a float: the two times are added in float, then 0.5 (double)
        # promotes the sum (oltb.cc:537-542)
        center = 0.5 * float(np.float32(point_times[start]) + np.float32(point_times[end]))
        t = int(cloud_time) + from_seconds(center)
        if not (ct[0] < t < ct[-1]):
            continue
        nxt = 1
        while ct[nxt] <= t:   # first control point later than t (exists: t < back)
            nxt += 1
        duration = to_seconds(ct[nxt] - ct[nxt - 1])
        ratio = min(max(to_seconds(t - ct[nxt - 1]) / duration, 0.0), 1.0)
        out.append((start, end + 1, nxt - 1, nxt, ratio))
    return out


def add_per_point_matching_residuals(problem, pose_ids, control_times, clouds, grids, weight,
                                     num_points_per_subdivision, multi_res=True):
    """use_per_point_unwarping branch of OptimizingLocalTrajectoryBuilder (oltb.cc:513-612): every
    subdivision gets its own interpolation ratio; on the device the subdivisions of one cloud that
    share a control-point pair are ONE unwarped block. clouds: [(cloud_time_ticks, xyz[n,3],
    point_times[n])]. Scaling = weight / sqrt(cloud size) (:573-577). Returns, per block added,
    (cloud index, prev, next, return indices, ratios) in residual order."""
    added = []
    for ci, (cloud_time, xyz, times) in enumerate(clouds):
        xyz = _host(xyz, np.float32, 3)
        subs = per_point_subdivisions(times, cloud_time, control_times, num_points_per_subdivision)
        scale = weight / np.sqrt(float(len(xyz)))
        pairs = {}
        for (s0, s1, a, b, ratio) in subs:
            idx, f = pairs.setdefault((a, b), ([], []))
            idx.extend(range(s0, s1))
            f.extend([ratio] * (s1 - s0))
        for (a, b) in sorted(pairs):
            idx = np.asarray(pairs[(a, b)][0], np.int64)
            f = np.asarray(pairs[(a, b)][1], np.float64)
            problem.add_unwarped_block(xyz[idx], f, grids, scale, pose_ids[a], pose_ids[b],
                                       multi_res=multi_res)
            added.append((ci, a, b, idx, f))
    return added


class CeresScanMatcher3D:
    """Single-pose matcher shape (Match / Evaluate) for TSDF grids.

    `occupied_space_weights[i]` pairs with the i-th (point cloud, grid) entry
    (ceres_scan_matcher_3d.cc:129-156). The translation/rotation prior blocks of
    the reference (:162-172) are not part of the TSDF hot path and are omitted.
    """

    def __init__(self, ctx, occupied_space_weights, **solver_kw):
        self.ctx = ctx
        self.weights = list(occupied_space_weights)
        self.solver_kw = solver_kw

    def _problem(self, initial_pose, clouds_and_grids):
        p = Problem(self.ctx)
        i = p.add_pose(initial_pose)
        for w, (cloud, grid) in zip(self.weights, clouds_and_grids):
            n = cloud.shape[0]
            p.add_block(cloud, [grid], w / np.sqrt(float(n)), i)
        return p, i

    def Match(self, initial_pose_estimate, point_clouds_and_hybrid_grids):
        p, i = self._problem(initial_pose_estimate, point_clouds_and_hybrid_grids)
        summary = p.solve(**self.solver_kw)
        pose = p.get_pose(i)
        p.close()
        return pose, summary

    def Evaluate(self, initial_pose_estimate, point_clouds_and_hybrid_grids):
        p, _ = self._problem(initial_pose_estimate, point_clouds_and_hybrid_grids)
        out = p.evaluate()
        p.close()
        return out
