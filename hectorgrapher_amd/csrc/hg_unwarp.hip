// hg_unwarp.hip — per-point unwarping of the accumulated range data before insertion.
//
// Reference (use_per_point_unwarping branch of OptimizingLocalTrajectoryBuilder::MaybeOptimize,
// mapping/internal/3d/optimizing_local_trajectory_builder.cc:1331-1379): after the window solve every
// return of the clouds that leave the window is moved into the tracking frame of the first control
// point with the pose interpolated AT THE RETURN'S OWN TIME,
//     point_time = cloud.time + FromSeconds(point.time)                                   (:1345-1346)
//     (prev, next) = the control points bracketing point_time                              (:1347-1359)
//     T = InterpolateTransform(prev.pose, next.pose, prev.time, next.time, point_time)     (:1361-1365,
//         transform/timestamped_transform.h:41-65: lerp + Eigen slerp in double)
//     returns.push_back((optimized_pose.inverse() * T).cast<float>() * point)              (:1366-1369)
// NaN returns are kept as they are (:1342-1345), the origin of the accumulated range data is the sensor
// origin of the cloud under the FIRST unwarped return's transform (:1370-1374). AddAccumulatedRangeData
// then moves the range data to the local frame with optimized_pose.cast<float>() (:1437-1440) and
// Submap3D::InsertData to the submap frame with local_pose().inverse().cast<float>()
// (mapping/3d/submap_3d.cc:436-437) before TSDFRangeDataInserter3D::Insert sees it.
//
// Device form: one thread per return does all of that (100k slerps per scan are what a host would
// otherwise do between the solve and the insertion, i.e. the round trip hg_register_scan exists to
// remove) and writes the float cloud the insert kernels read; the origin comes from a one-thread kernel
// behind it (the first non-NaN return is a min-reduction over the returns). The cloud IS written to
// device memory once (12 B per return, against ~550 B per return the binned insertion moves): the three
// pyramid levels and the count and scatter passes of the insertion would otherwise each repeat the fp64
// slerp, and the CLOUD_STRUCTURE normals read neighbouring returns, which must be unwarped as well.
#include <cstring>
#include <vector>

#include "hg_internal.h"

namespace hg {

struct UnwarpCloudDev {
  long long time;            // universal ticks (100 ns)
  unsigned long long begin;  // first point of the cloud
  unsigned long long count;
  float origin[3];
  float pad;
};

constexpr int kUnwarpInlineControl = 48;  // control points / clouds that travel as kernel arguments
constexpr int kUnwarpInlineClouds = 4;

struct UnwarpParams {
  const float* points;        // n x 4: x y z time[s relative to its cloud]
  unsigned long long n;
  const UnwarpCloudDev* clouds;
  int n_clouds;
  const double* poses;        // control poses, pose k at poses + pose_index[k] * pose_stride (t xyz, q wxyz)
  const int* pose_index;      // or nullptr: k
  int pose_stride;
  const long long* times;     // control point times (ticks), ascending
  int n_control;
  int optimized;              // control point whose pose is optimized_pose (the front of the window: 0)
  int to_local;               // apply optimized_pose.cast<float>() (TransformTimedRangeData, :1437-1440)
  int has_post;               // apply post_tq (Submap3D::InsertData's frame change)
  float post_tq[7];
  float* xyz_out;             // n x 3
  float* origin_out;          // 3 floats
  unsigned* first_valid;      // index of the first non-NaN return (initialised to 0xFFFFFFFF)
  unsigned* time_ok;          // the call's own status word: 0xFFFFFFFF, or 0 once a return's time lies outside the
                              // control points (the reference CHECK-fails: nothing of the call may reach a map)
  uint32_t* flag_words[4];    // counters[1] of the grids the cloud goes to (sticky error flags) or nullptr
};

// The small tables of a call as kernel arguments (the usual case: a window's control points and one or two
// clouds): no upload, no staging buffer whose reuse would have to be fenced. The kernels read them with
// per-lane indices straight from the kernel-argument segment.
struct UnwarpInline {
  long long times[kUnwarpInlineControl];
  int pose_index[kUnwarpInlineControl];
  UnwarpCloudDev clouds[kUnwarpInlineClouds];
};
struct UnwarpInlinePoses {
  double poses[kUnwarpInlineControl * 7];
};

struct RigidD {
  double t[3];
  double q[4];  // w x y z
};

// Eigen 3.3 QuaternionBase::_transformVector (generic path), double.
__device__ inline void rotate_d(const double* q, const double* v, double* out) {
  const double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  double ux = qy * v[2] - qz * v[1], uy = qz * v[0] - qx * v[2], uz = qx * v[1] - qy * v[0];
  ux = ux + ux; uy = uy + uy; uz = uz + uz;
  const double cx = qy * uz - qz * uy, cy = qz * ux - qx * uz, cz = qx * uy - qy * ux;
  out[0] = v[0] + qw * ux + cx;
  out[1] = v[1] + qw * uy + cy;
  out[2] = v[2] + qw * uz + cz;
}

// transform/timestamped_transform.h:41-51 with Eigen 3.3 QuaternionBase::slerp (double).
__device__ inline RigidD interpolate_transform(const double* a, const double* b, double factor) {
  RigidD r;
#pragma unroll
  for (int k = 0; k < 3; ++k) r.t[k] = a[k] + (b[k] - a[k]) * factor;
  const double aw = a[3], ax = a[4], ay = a[5], az = a[6];
  const double bw = b[3], bx = b[4], by = b[5], bz = b[6];
  const double one = 1.0 - 2.220446049250313e-16;
  const double d = (ax * bx + ay * by) + (az * bz + aw * bw);  // coeffs are stored (x, y, z, w)
  const double abs_d = fabs(d);
  double s0, s1;
  if (abs_d >= one) {
    s0 = 1.0 - factor;
    s1 = factor;
  } else {
    const double theta = acos(abs_d);
    const double sin_theta = sin(theta);
    s0 = sin((1.0 - factor) * theta) / sin_theta;
    s1 = sin(factor * theta) / sin_theta;
  }
  if (d < 0.0) s1 = -s1;
  r.q[0] = s0 * aw + s1 * bw;
  r.q[1] = s0 * ax + s1 * bx;
  r.q[2] = s0 * ay + s1 * by;
  r.q[3] = s0 * az + s1 * bz;
  return r;
}

// The float transform chain of return `i` (or of the origin of i's cloud): bracket, interpolate,
// (optimized^-1 * T).cast<float>(), then the optional frame changes. Returns false when the time lies
// outside the control points (the reference CHECK-fails there; the caller raises the grids' flag).
__device__ inline bool unwarp_transform(const UnwarpParams& P, long long cloud_time, float point_time, float tq[7]) {
  // common::FromSeconds (common/time.cc:30-33): duration_cast to 100 ns ticks truncates toward zero
  const long long pt = cloud_time + static_cast<long long>(static_cast<double>(point_time) * 1e7);
  // :1347-1354 -- the walk ends at the first control point later than the return (or at the last one)
  int next = 1;
  const int K = P.n_control;
  while (next < K - 1 && P.times[next] <= pt) ++next;
  const int prev = next - 1;
  const long long t0 = P.times[prev], t1 = P.times[next];
  const bool inside = t0 <= pt && pt <= t1;  // CHECK_LE / CHECK_GE (:1358-1359)
  // timestamped_transform.h:59-62, common::ToSeconds = ticks / 1e7 in double
  const double duration = static_cast<double>(t1 - t0) / 1e7;
  const double factor = (static_cast<double>(pt - t0) / 1e7) / duration;
  const double* pa = P.poses + static_cast<size_t>(P.pose_index ? P.pose_index[prev] : prev) * P.pose_stride;
  const double* pb = P.poses + static_cast<size_t>(P.pose_index ? P.pose_index[next] : next) * P.pose_stride;
  const double* po = P.poses + static_cast<size_t>(P.pose_index ? P.pose_index[P.optimized] : P.optimized) * P.pose_stride;
  const RigidD T = interpolate_transform(pa, pb, factor);
  // optimized_pose.inverse() (rigid_transform.h:159-163)
  const double rc[4] = {po[3], -po[4], -po[5], -po[6]};
  double ti[3];
  rotate_d(rc, po, ti);
  ti[0] = -ti[0]; ti[1] = -ti[1]; ti[2] = -ti[2];
  // inverse * T (rigid_transform.h:184-190): rotation normalized()
  double tt[3];
  rotate_d(rc, T.t, tt);
  tt[0] = tt[0] + ti[0]; tt[1] = tt[1] + ti[1]; tt[2] = tt[2] + ti[2];
  const double aw = rc[0], ax = rc[1], ay = rc[2], az = rc[3];
  const double bw = T.q[0], bx = T.q[1], by = T.q[2], bz = T.q[3];
  double qw = aw * bw - ax * bx - ay * by - az * bz;
  double qx = aw * bx + ax * bw + ay * bz - az * by;
  double qy = aw * by + ay * bw + az * bx - ax * bz;
  double qz = aw * bz + az * bw + ax * by - ay * bx;
  const double nrm = sqrt((qx * qx + qy * qy) + (qz * qz + qw * qw));
  qw = qw / nrm; qx = qx / nrm; qy = qy / nrm; qz = qz / nrm;
  tq[0] = static_cast<float>(tt[0]); tq[1] = static_cast<float>(tt[1]); tq[2] = static_cast<float>(tt[2]);
  tq[3] = static_cast<float>(qw); tq[4] = static_cast<float>(qx); tq[5] = static_cast<float>(qy);
  tq[6] = static_cast<float>(qz);
  return inside;
}

// The frame changes behind the unwarping: every return of the range data goes through them, NaN ones too
// (TransformTimedRangeData :1437-1440, TransformRangeData submap_3d.cc:436-437).
__device__ inline void unwarp_frames(const UnwarpParams& P, float& x, float& y, float& z) {
  if (P.to_local) {
    const double* po = P.poses + static_cast<size_t>(P.pose_index ? P.pose_index[P.optimized] : P.optimized) * P.pose_stride;
    float lf[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) lf[k] = static_cast<float>(po[k]);  // Rigid3d::cast<float>()
    transform_point(lf, x, y, z);
  }
  if (P.has_post) transform_point(P.post_tq, x, y, z);
}

__device__ inline int unwarp_find_cloud(const UnwarpParams& P, unsigned long long i) {
  int c = 0;
  while (c + 1 < P.n_clouds && P.clouds[c + 1].begin <= i) ++c;
  return c;
}

__device__ __forceinline__ void unwarp_points_body(const UnwarpParams& P) {
  const unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool in = i < P.n;
  bool valid = false, outside = false;
  if (in) {
    const float4 p = reinterpret_cast<const float4*>(P.points)[i];
    float x = p.x, y = p.y, z = p.z;
    valid = !(isnan(x) || isnan(y) || isnan(z));  // point.position.hasNaN(): pushed back unchanged (:1342-1345)
    if (valid) {
      const UnwarpCloudDev& cl = P.clouds[unwarp_find_cloud(P, i)];
      float tq[7];
      outside = !unwarp_transform(P, cl.time, p.w, tq);
      transform_point(tq, x, y, z);
    }
    unwarp_frames(P, x, y, z);
    P.xyz_out[3 * i] = x;
    P.xyz_out[3 * i + 1] = y;
    P.xyz_out[3 * i + 2] = z;
  }
  // first non-NaN return: one atomic per wavefront that holds a candidate below the current minimum
  const unsigned long long m = __ballot(valid);
  if (m != 0ull && (threadIdx.x % kWave) == static_cast<unsigned>(__ffsll(static_cast<long long>(m)) - 1)) {
    const unsigned idx = static_cast<unsigned>(i);
    if (idx < *reinterpret_cast<volatile unsigned*>(P.first_valid)) atomicMin(P.first_valid, idx);
  }
  if (__ballot(outside) != 0ull && (threadIdx.x % kWave) == 0) {
    atomicAnd(P.time_ok, 0u);
#pragma unroll
    for (int l = 0; l < 4; ++l)
      if (P.flag_words[l]) atomicOr(P.flag_words[l], kFlagTime);
  }
}

// accumulated_range_data_in_tracking.origin = transform * front().origin at the first unwarped return
// (:1370-1374); Vector3f::Zero() when every return is NaN (:1298-1299). Launched over all returns: when the
// points pass found a time outside the control points the reference would have CHECK-failed before anything
// was inserted (:1358-1359), so every return of the failed call is replaced by NaN -- the gates of the
// insertion (and of everything else that reads the device copy) drop NaN returns, the maps stay as they were
// and the caller gets HG_ERR_TIME. The usual case costs one uniform load per workgroup.
__device__ __forceinline__ void unwarp_origin_body(const UnwarpParams& P) {
  const unsigned long long g = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (*P.time_ok == 0u && g < P.n) {
    const float nan = __int_as_float(0x7FC00000);
    P.xyz_out[3 * g] = nan;
    P.xyz_out[3 * g + 1] = nan;
    P.xyz_out[3 * g + 2] = nan;
  }
  if (g != 0) return;
  const unsigned i = *P.first_valid;
  float ox = 0.f, oy = 0.f, oz = 0.f;
  if (i != 0xFFFFFFFFu) {
    const UnwarpCloudDev& cl = P.clouds[unwarp_find_cloud(P, i)];
    float tq[7];
    unwarp_transform(P, cl.time, P.points[4ull * i + 3], tq);
    ox = cl.origin[0]; oy = cl.origin[1]; oz = cl.origin[2];
    transform_point(tq, ox, oy, oz);
  }
  unwarp_frames(P, ox, oy, oz);  // (a zero origin -- every return NaN -- goes through the frame changes as well)
  P.origin_out[0] = ox;
  P.origin_out[1] = oy;
  P.origin_out[2] = oz;
}

// Tables in device memory (more control points or clouds than the inline form holds).
__global__ __launch_bounds__(256) void k_unwarp_points(UnwarpParams P) { unwarp_points_body(P); }
__global__ __launch_bounds__(256) void k_unwarp_origin(UnwarpParams P) { unwarp_origin_body(P); }

// Tables as kernel arguments; the control poses in device memory (what a solve left there) ...
__global__ __launch_bounds__(256) void k_unwarp_points_inline(UnwarpParams P, const UnwarpInline T) {
  P.times = T.times;
  P.pose_index = P.pose_index ? T.pose_index : nullptr;
  P.clouds = T.clouds;
  unwarp_points_body(P);
}
__global__ __launch_bounds__(256) void k_unwarp_origin_inline(UnwarpParams P, const UnwarpInline T) {
  P.times = T.times;
  P.pose_index = P.pose_index ? T.pose_index : nullptr;
  P.clouds = T.clouds;
  unwarp_origin_body(P);
}
// ... or as kernel arguments as well (poses given by the host).
__global__ __launch_bounds__(256) void k_unwarp_points_inline_poses(UnwarpParams P, const UnwarpInline T,
                                                                    const UnwarpInlinePoses X) {
  P.times = T.times;
  P.pose_index = nullptr;
  P.clouds = T.clouds;
  P.poses = X.poses;
  unwarp_points_body(P);
}
__global__ __launch_bounds__(256) void k_unwarp_origin_inline_poses(UnwarpParams P, const UnwarpInline T, const UnwarpInlinePoses X) {
  P.times = T.times;
  P.pose_index = nullptr;
  P.clouds = T.clouds;
  P.poses = X.poses;
  unwarp_origin_body(P);
}

}  // namespace hg

using namespace hg;

// Unwarps `n` timed returns (device or host memory) into the context's unwarp buffer and its origin slot.
// d_poses != nullptr: control poses in device memory (the state a solve left there) with `pose_index` /
// `pose_stride`; else control_poses (host, K x 7) are uploaded. Everything is enqueued on the context's
// stream; nothing synchronises.
int hg::unwarp_enqueue(hg_ctx* c, hg_grid* const* grids, int levels, const float* points, size_t n, int memspace,
                       const hg_timed_cloud* clouds, int n_clouds, const double* control_poses,
                       const double* d_poses, const int* pose_index, int pose_stride,
                       const int64_t* control_times, int n_control, int optimized, int to_local,
                       const float* post_tq) {
  if (!c || !points || n == 0 || n >= 0xFFFFFFFFull || !clouds || n_clouds < 1 || !control_times || n_control < 2 ||
      (!control_poses && !d_poses) || optimized < 0 || optimized >= n_control || levels > 4)
    return HG_ERR_INVALID;
  for (int k = 1; k < n_control; ++k)
    if (control_times[k] <= control_times[k - 1]) {
      set_last_error("control point times must ascend");
      return HG_ERR_INVALID;
    }
  unsigned long long covered = 0;
  for (int k = 0; k < n_clouds; ++k) {
    if (clouds[k].begin != covered) {
      set_last_error("timed clouds must tile [0, n) in order");
      return HG_ERR_INVALID;
    }
    covered += clouds[k].count;
  }
  if (covered != n) {
    set_last_error("timed clouds must tile [0, n) in order");
    return HG_ERR_INVALID;
  }
  hipStream_t s = c->stream;
  HG_HIP_CHECK(hipSetDevice(c->device));
  int rc;
  if ((rc = c->ws_unwarp.reserve(n * 3 * sizeof(float))) != HG_OK) return rc;
  const float* d_points = points;
  if (memspace == HG_HOST) {
    if ((rc = c->ws_unwarp_in.reserve(n * 4 * sizeof(float))) != HG_OK) return rc;
    HG_HIP_CHECK(hipMemcpyAsync(c->ws_unwarp_in.ptr, points, n * 4 * sizeof(float), hipMemcpyHostToDevice, s));
    d_points = c->ws_unwarp_in.as<float>();
  }
  const bool inline_tables = n_control <= kUnwarpInlineControl && n_clouds <= kUnwarpInlineClouds;
  // device layout of the small tables: [clouds | times | pose indices | poses] (only when they do not travel
  // as kernel arguments), then the first-valid word and the origin slot
  const size_t cloud_bytes = sizeof(UnwarpCloudDev) * static_cast<size_t>(n_clouds);
  const size_t time_bytes = sizeof(long long) * static_cast<size_t>(n_control);
  const size_t index_bytes = (sizeof(int) * static_cast<size_t>(n_control) + 7u) & ~size_t(7);
  const size_t pose_bytes = d_poses ? 0 : sizeof(double) * 7u * static_cast<size_t>(n_control);
  const size_t table_bytes = inline_tables ? 0 : cloud_bytes + time_bytes + index_bytes + pose_bytes;
  if ((rc = c->ws_unwarp_tab.reserve(table_bytes + 64)) != HG_OK) return rc;
  char* base = c->ws_unwarp_tab.as<char>();
  UnwarpParams P;
  std::memset(&P, 0, sizeof(P));
  UnwarpInline T;
  UnwarpInlinePoses X;
  if (inline_tables) {
    std::memset(&T, 0, sizeof(T));
    for (int k = 0; k < n_clouds; ++k) {
      T.clouds[k].time = clouds[k].time;
      T.clouds[k].begin = clouds[k].begin;
      T.clouds[k].count = clouds[k].count;
      std::memcpy(T.clouds[k].origin, clouds[k].origin, sizeof(T.clouds[k].origin));
    }
    std::memcpy(T.times, control_times, time_bytes);
    if (pose_index) std::memcpy(T.pose_index, pose_index, sizeof(int) * n_control);
    if (!d_poses) std::memcpy(X.poses, control_poses, pose_bytes);
    P.pose_index = pose_index ? reinterpret_cast<const int*>(1) : nullptr;  // (the kernel points it at its argument)
  } else {
    // the staging vector must outlive the copy: a copy from pageable memory is staged before it returns
    std::vector<unsigned char> host(table_bytes, 0);
    UnwarpCloudDev* hc = reinterpret_cast<UnwarpCloudDev*>(host.data());
    for (int k = 0; k < n_clouds; ++k) {
      hc[k].time = clouds[k].time;
      hc[k].begin = clouds[k].begin;
      hc[k].count = clouds[k].count;
      std::memcpy(hc[k].origin, clouds[k].origin, sizeof(hc[k].origin));
      hc[k].pad = 0.f;
    }
    std::memcpy(host.data() + cloud_bytes, control_times, time_bytes);
    if (pose_index) std::memcpy(host.data() + cloud_bytes + time_bytes, pose_index, sizeof(int) * n_control);
    if (!d_poses) std::memcpy(host.data() + cloud_bytes + time_bytes + index_bytes, control_poses, pose_bytes);
    HG_HIP_CHECK(hipMemcpyAsync(base, host.data(), table_bytes, hipMemcpyHostToDevice, s));
    P.clouds = reinterpret_cast<const UnwarpCloudDev*>(base);
    P.times = reinterpret_cast<const long long*>(base + cloud_bytes);
    P.pose_index = pose_index ? reinterpret_cast<const int*>(base + cloud_bytes + time_bytes) : nullptr;
    if (!d_poses) P.poses = reinterpret_cast<const double*>(base + cloud_bytes + time_bytes + index_bytes);
  }
  HG_HIP_CHECK(hipMemsetAsync(base + table_bytes, 0xFF, 2 * sizeof(unsigned), s));  // first_valid = none yet, time_ok
  P.points = d_points;
  P.n = n;
  P.n_clouds = n_clouds;
  P.n_control = n_control;
  if (d_poses) {
    P.poses = d_poses;
    P.pose_stride = pose_stride;
  } else {
    P.pose_stride = 7;
  }
  P.optimized = optimized;
  P.to_local = to_local;
  P.has_post = post_tq ? 1 : 0;
  if (post_tq) std::memcpy(P.post_tq, post_tq, sizeof(P.post_tq));
  P.xyz_out = c->ws_unwarp.as<float>();
  P.first_valid = reinterpret_cast<unsigned*>(base + table_bytes);
  P.time_ok = reinterpret_cast<unsigned*>(base + table_bytes + 4);
  P.origin_out = reinterpret_cast<float*>(base + table_bytes + 16);
  for (int l = 0; l < levels && l < 4; ++l) P.flag_words[l] = grids && grids[l] ? grids[l]->view.counters + 1 : nullptr;
  {
    ProfScope ps(c, HG_K_UNWARP, n);
    const dim3 grid(static_cast<unsigned>((n + 255) / 256));
    if (inline_tables && d_poses) {
      hipLaunchKernelGGL(k_unwarp_points_inline, grid, dim3(256), 0, s, P, T);
      hipLaunchKernelGGL(k_unwarp_origin_inline, grid, dim3(256), 0, s, P, T);
    } else if (inline_tables) {
      hipLaunchKernelGGL(k_unwarp_points_inline_poses, grid, dim3(256), 0, s, P, T, X);
      hipLaunchKernelGGL(k_unwarp_origin_inline_poses, grid, dim3(256), 0, s, P, T, X);
    } else {
      hipLaunchKernelGGL(k_unwarp_points, grid, dim3(256), 0, s, P);
      hipLaunchKernelGGL(k_unwarp_origin, grid, dim3(256), 0, s, P);
    }
  }
  HG_HIP_CHECK(hipGetLastError());
  c->unwarp_xyz = P.xyz_out;
  c->unwarp_origin = P.origin_out;
  c->unwarp_count = n;
  c->unwarp_time_ok = P.time_ok;
  return HG_OK;
}

// Unwarp + insertion of the result (the cloud is in the grids' frame already, its origin in device memory).
int hg::unwarp_insert(hg_grid* const* grids, const hg_insert_opts* opts, int levels, const float* points, size_t n,
                      size_t width, int memspace, const hg_timed_cloud* clouds, int n_clouds,
                      const double* control_poses, const double* d_poses, const int* pose_index, int pose_stride,
                      const int64_t* control_times, int n_control, const float* pose_tq, int mode,
                      hg_insert_stats* stats) {
  if (!grids || !opts || levels < 1 || levels > 4 || !grids[0] || !control_poses) return HG_ERR_INVALID;
  hg_ctx* c = grids[0]->ctx;
  if (!c) {
    set_last_error("the grid's context has been destroyed");
    return HG_ERR_INVALID;
  }
  int rc = unwarp_enqueue(c, grids, levels, points, n, memspace, clouds, n_clouds, control_poses, d_poses, pose_index,
                          pose_stride, control_times, n_control, 0, 1, pose_tq);
  if (rc != HG_OK) return rc;
  // host-side guess of the origin (sizes the key window of the sort path): the sensor origin of the first
  // cloud at the front control pose; the true origin lies within the window's motion of it
  float approx[3];
  {
    const double* po = control_poses;
    float lf[7];
    for (int k = 0; k < 7; ++k) lf[k] = static_cast<float>(po[k]);
    const float* o = clouds[0].origin;
    const float qw = lf[3], qx = lf[4], qy = lf[5], qz = lf[6];
    float ux = qy * o[2] - qz * o[1], uy = qz * o[0] - qx * o[2], uz = qx * o[1] - qy * o[0];
    ux += ux; uy += uy; uz += uz;
    float a[3] = {o[0] + qw * ux + (qy * uz - qz * uy) + lf[0], o[1] + qw * uy + (qz * ux - qx * uz) + lf[1],
                  o[2] + qw * uz + (qx * uy - qy * ux) + lf[2]};
    if (pose_tq) {
      const float pw = pose_tq[3], px = pose_tq[4], py = pose_tq[5], pz = pose_tq[6];
      float vx = py * a[2] - pz * a[1], vy = pz * a[0] - px * a[2], vz = px * a[1] - py * a[0];
      vx += vx; vy += vy; vz += vz;
      const float b[3] = {a[0] + pw * vx + (py * vz - pz * vy) + pose_tq[0], a[1] + pw * vy + (pz * vx - px * vz) + pose_tq[1],
                          a[2] + pw * vz + (px * vy - py * vx) + pose_tq[2]};
      std::memcpy(a, b, sizeof(a));
    }
    std::memcpy(approx, a, sizeof(approx));
  }
  const uint64_t offsets[2] = {0, n};
  return pyramid_insert_impl(grids, opts, levels, approx, c->unwarp_xyz, offsets, 1, width, nullptr, nullptr, mode,
                             HG_DEVICE, stats, c->unwarp_origin);
}

extern "C" {

int hg_pyramid_insert_unwarped(hg_grid* const* grids, const hg_insert_opts* opts, int levels, const float* points,
                               size_t n, size_t width, int memspace, const hg_timed_cloud* clouds, int n_clouds,
                               const double* control_poses, const int64_t* control_times, int n_control,
                               const float* pose_tq, int mode, hg_insert_stats* stats) {
  return unwarp_insert(grids, opts, levels, points, n, width, memspace, clouds, n_clouds, control_poses, nullptr,
                       nullptr, 7, control_times, n_control, pose_tq, mode, stats);
}

int hg_unwarp_range_data(hg_ctx* ctx, const float* points, size_t n, int memspace, const hg_timed_cloud* clouds,
                         int n_clouds, const double* control_poses, const int64_t* control_times, int n_control,
                         int frame, const float* pose_tq, float* xyz_out, float origin_out[3]) {
  if (!ctx || frame < 0 || frame > 1) return HG_ERR_INVALID;
  int rc = unwarp_enqueue(ctx, nullptr, 0, points, n, memspace, clouds, n_clouds, control_poses, nullptr, nullptr, 7,
                          control_times, n_control, 0, frame, pose_tq);
  if (rc != HG_OK) return rc;
  if (xyz_out)
    HG_HIP_CHECK(hipMemcpyAsync(xyz_out, ctx->unwarp_xyz, n * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (origin_out)
    HG_HIP_CHECK(hipMemcpyAsync(origin_out, ctx->unwarp_origin, 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (xyz_out || origin_out) return hg_unwarp_status(ctx);  // waits for the copies; HG_ERR_TIME as the reference CHECKs
  return HG_OK;
}

int hg_unwarp_status(hg_ctx* ctx) {
  if (!ctx) return HG_ERR_INVALID;
  if (!ctx->unwarp_time_ok) return HG_OK;
  unsigned ok = 0;
  HG_HIP_CHECK(hipSetDevice(ctx->device));
  HG_HIP_CHECK(hipMemcpyAsync(&ok, ctx->unwarp_time_ok, sizeof(ok), hipMemcpyDeviceToHost, ctx->stream));
  HG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return ok ? HG_OK : flags_to_status(kFlagTime);
}

int hg_unwarp_last_device(hg_ctx* ctx, const float** xyz_dev, const float** origin_dev, size_t* count) {
  if (!ctx) return HG_ERR_INVALID;
  if (xyz_dev) *xyz_dev = ctx->unwarp_xyz;
  if (origin_dev) *origin_dev = ctx->unwarp_origin;
  if (count) *count = ctx->unwarp_count;
  return HG_OK;
}

}  // extern "C"
