// hg_match.hip — TSDF scan matching on the device.
//
// Replaces the ceres::Problem that OptimizingLocalTrajectoryBuilder assembles from the TSDF
// space cost functions and solves with ceres::Solve
// (ref mapping/internal/3d/optimizing_local_trajectory_builder.cc:323-511,1238-1291).
//
//   k_tsdf_residuals  per return: world = T*p (fp64, Eigen quaternion formula), trilinear
//                     (multi-resolution) TSDF lookup with the reference's validity branching
//                     (scan_matching/interpolated_tsdf.h:30-116,
//                     interpolated_multi_resolution_tsdf.h:30-137), analytic gradient, the
//                     1x7 row d r / d (t, q) of the interpolated transform, and a wavefront +
//                     workgroup reduction of the 7x7 normal-equation block (28 + 7 + 1 fp64
//                     sums) into one partial per workgroup — no atomics, bitwise reproducible.
//   k_lm              one workgroup: sums the partials, maps every block's 7x7 system through
//                     d(t,q)/d(local parameters) (identity / QuaternionParameterization /
//                     the slerp chain of InterpolateTransform) into the dense normal equations
//                     and runs one step of the Ceres-1.13 trust-region LM state machine
//                     (Jacobi scaling, LM diagonal, Cholesky, step quality, radius update,
//                     tolerances) entirely on the device.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <vector>

#include "hg_internal.h"

namespace hg {

constexpr int kMaxLevels = 4;
constexpr int kMaxPoses = 8;
constexpr int kMaxCols = 6 * kMaxPoses;
constexpr int kMaxBlocks = 32;
constexpr int kAcc = 36;  // 28 (upper triangle of 7x7) + 7 + 1
constexpr int kEvalThreads = 256;

struct PyramidView {
  GridView level[kMaxLevels];
  int levels;
  int multi_res;
};

struct BlockXform {
  double t[3];
  double q[4];       // w x y z of the (interpolated) transform
  double M[7 * 12];  // d(t,q) / d(local params of pose_a [0..5], pose_b [6..11]), row-major
};

struct BlockInfo {
  int pose_a, pose_b;
  double factor;
  double scaling;
  unsigned n;
  unsigned num_wg;
  unsigned partial_offset;  // in units of workgroups
  unsigned row_offset;
  int active;
};

enum { PHASE_INIT = 0, PHASE_CANDIDATE = 1 };
enum { MODE_PREPARE = 0, MODE_STEP = 1, MODE_ASSEMBLE = 2 };

struct LmState {
  int num_poses, num_blocks, ncols, done;
  int iteration, phase, step_is_successful, reuse_diagonal;
  int invalid_steps, termination_type, termination_reason, num_iterations;
  int num_successful, num_unsuccessful, num_cost_evals, num_jac_evals;
  hg_solver_opts opt;
  double radius, decrease_factor;
  double x_cost, cand_cost, model_cost_change, gradient_max_norm, initial_cost;
  double x[kMaxPoses][7];
  double cand[kMaxPoses][7];
  int constant[kMaxPoses];
  int col[kMaxPoses];
  double scale[kMaxCols], diagonal[kMaxCols], g[kMaxCols], step[kMaxCols], delta[kMaxCols];
  double gc[kMaxCols];
  double H[kMaxCols * kMaxCols];
  double Hc[kMaxCols * kMaxCols];
  double work[kMaxCols * kMaxCols];
  BlockInfo blocks[kMaxBlocks];
};

// ------------------------------------------------------------------------------------------
// per-return residual + row
// ------------------------------------------------------------------------------------------
struct D3 {  // value + gradient w.r.t. world (x, y, z); mirrors ceres::Jet<double, 3>
  double a, d0, d1, d2;
};

// InterpolateLinear (interpolated_tsdf.h:30-46 / interpolated_multi_resolution_tsdf.h:30-46)
__device__ inline void interpolate_linear(double both_invalid, const D3& q1, const D3& q2, double w1,
                                          double w2, const D3& r, D3& q, double& w) {
  if (w1 == 0.0 && w2 == 0.0) {
    q = {both_invalid, 0.0, 0.0, 0.0};
    w = 0.0;
  } else if (w1 == 0.0) {
    q = q2;
    w = w2;
  } else if (w2 == 0.0) {
    q = q1;
    w = w1;
  } else {
    const double da = q2.a - q1.a;
    q.a = da * r.a + q1.a;
    q.d0 = (da * r.d0 + (q2.d0 - q1.d0) * r.a) + q1.d0;
    q.d1 = (da * r.d1 + (q2.d1 - q1.d1) * r.a) + q1.d1;
    q.d2 = (da * r.d2 + (q2.d2 - q1.d2) * r.a) + q1.d2;
    w = w1 + w2;
  }
}

// One pyramid level. Returns false when the multi-resolution lookup must fall through to the
// next coarser level (any of the 8 weights is zero).
__device__ inline bool level_tsd(const GridView& g, bool multi, double x, double y, double z, D3& out) {
  const float res = g.resolution;
  // CenterOfLowerVoxel (interpolated_tsdf.h:176-192): float centre, compared against the double
  float cx = static_cast<float>(cell_index_1d(static_cast<float>(x), res)) * res;
  float cy = static_cast<float>(cell_index_1d(static_cast<float>(y), res)) * res;
  float cz = static_cast<float>(cell_index_1d(static_cast<float>(z), res)) * res;
  if (static_cast<double>(cx) > x) cx -= res;
  if (static_cast<double>(cy) > y) cy -= res;
  if (static_cast<double>(cz) > z) cz -= res;
  const double x1 = cx, y1 = cy, z1 = cz;
  const double x2 = cx + res, y2 = cy + res, z2 = cz + res;
  const int ix = cell_index_1d(cx, res), iy = cell_index_1d(cy, res), iz = cell_index_1d(cz, res);

  // 8 voxel codes; corners that share the first corner's block reuse its slot
  uint32_t code[8];
  {
    const bool in0 = cell_in_range(ix, iy, iz);
    const unsigned long long key0 = in0 ? block_key(ix, iy, iz) : ~0ull;
    const uint32_t slot0 = in0 ? find_block(g, key0) : 0xFFFFFFFFu;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      // corner order: c = dx*4 + dy*2 + dz  (111,112,121,122,211,212,221,222)
      const int px = ix + (c >> 2), py = iy + ((c >> 1) & 1), pz = iz + (c & 1);
      uint32_t v = 0u;
      if (cell_in_range(px, py, pz)) {
        const unsigned long long key = block_key(px, py, pz);
        const uint32_t slot = (key == key0) ? slot0 : find_block(g, key);
        if (slot < g.max_blocks)
          v = g.voxels[static_cast<size_t>(slot) * kVoxelsPerBlock + voxel_in_block(px, py, pz)];
      }
      code[c] = v;
    }
  }
  double w[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) w[c] = static_cast<double>(value_to_weight(g, code[c] >> 16));
  double both_invalid;
  if (multi) {
    int invalid = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) invalid += (w[c] == 0.0) ? 1 : 0;
    if (invalid > 0) return false;
    both_invalid = static_cast<double>(g.min_tsd);
  } else {
    bool all_zero = true;
#pragma unroll
    for (int c = 0; c < 8; ++c) all_zero = all_zero && (w[c] == 0.0);
    if (all_zero) {
      out = {static_cast<double>(g.min_tsd), 0.0, 0.0, 0.0};
      return true;
    }
    both_invalid = -0.3;
  }
  D3 q[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
    q[c] = {static_cast<double>(value_to_tsd(g, code[c] & 0xFFFFu)), 0.0, 0.0, 0.0};
  const double ix_inv = 1.0 / (x2 - x1), iy_inv = 1.0 / (y2 - y1), iz_inv = 1.0 / (z2 - z1);
  const D3 nx = {(x - x1) / (x2 - x1), ix_inv, 0.0, 0.0};
  const D3 ny = {(y - y1) / (y2 - y1), 0.0, iy_inv, 0.0};
  const D3 nz = {(z - z1) / (z2 - z1), 0.0, 0.0, iz_inv};
  D3 q11, q12, q21, q22, q1, q2, qq;
  double w11, w12, w21, w22, w1, w2, ww;
  interpolate_linear(both_invalid, q[0], q[1], w[0], w[1], nz, q11, w11);
  interpolate_linear(both_invalid, q[2], q[3], w[2], w[3], nz, q12, w12);
  interpolate_linear(both_invalid, q[4], q[5], w[4], w[5], nz, q21, w21);
  interpolate_linear(both_invalid, q[6], q[7], w[6], w[7], nz, q22, w22);
  interpolate_linear(both_invalid, q11, q12, w11, w12, ny, q1, w1);
  interpolate_linear(both_invalid, q21, q22, w21, w22, ny, q2, w2);
  interpolate_linear(both_invalid, q1, q2, w1, w2, nx, qq, ww);
  out = qq;
  return true;
}

__device__ inline D3 pyramid_tsd(const PyramidView& pv, double x, double y, double z) {
  D3 out;
  if (!pv.multi_res) {
    level_tsd(pv.level[0], false, x, y, z, out);
    return out;
  }
  for (int l = 0; l < pv.levels; ++l)
    if (level_tsd(pv.level[l], true, x, y, z, out)) return out;
  return {static_cast<double>(pv.level[0].min_tsd), 0.0, 0.0, 0.0};
}

__device__ inline void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

// residuals of one block at its current transform + 36 partial sums per workgroup
__global__ __launch_bounds__(kEvalThreads) void k_tsdf_residuals(
    PyramidView pv, const float* __restrict__ xyz, unsigned n, double scaling,
    const BlockXform* __restrict__ xf, double* __restrict__ partials,
    double* __restrict__ residuals, const int* __restrict__ done_flag) {
  if (done_flag && *done_flag) return;
  const unsigned i = blockIdx.x * kEvalThreads + threadIdx.x;
  double acc[kAcc];
#pragma unroll
  for (int k = 0; k < kAcc; ++k) acc[k] = 0.0;
  if (i < n) {
    const double qw = xf->q[0];
    const double u[3] = {xf->q[1], xf->q[2], xf->q[3]};
    const double v[3] = {static_cast<double>(xyz[3 * i]), static_cast<double>(xyz[3 * i + 1]),
                         static_cast<double>(xyz[3 * i + 2])};
    // Eigen QuaternionBase::_transformVector, then + translation (rigid_transform.h:193-197)
    double uv[3], c2[3];
    cross3(u, v, uv);
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    cross3(u, uv, c2);
    const double wx = (v[0] + qw * uv[0] + c2[0]) + xf->t[0];
    const double wy = (v[1] + qw * uv[1] + c2[1]) + xf->t[1];
    const double wz = (v[2] + qw * uv[2] + c2[2]) + xf->t[2];
    const D3 tsd = pyramid_tsd(pv, wx, wy, wz);
    const double r = scaling * tsd.a;
    if (residuals) residuals[i] = r;
    const double g[3] = {scaling * tsd.d0, scaling * tsd.d1, scaling * tsd.d2};
    // d world / d q = [uv | w*duv_k + e_k x uv + u x duv_k], duv_k = 2 (e_k x v)
    double row[7];
    row[0] = g[0]; row[1] = g[1]; row[2] = g[2];
    row[3] = g[0] * uv[0] + g[1] * uv[1] + g[2] * uv[2];
    const double ekv[3][3] = {{0.0, -v[2], v[1]}, {v[2], 0.0, -v[0]}, {-v[1], v[0], 0.0}};
    const double eku[3][3] = {{0.0, -uv[2], uv[1]}, {uv[2], 0.0, -uv[0]}, {-uv[1], uv[0], 0.0}};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double duv[3] = {2.0 * ekv[k][0], 2.0 * ekv[k][1], 2.0 * ekv[k][2]};
      double ud[3];
      cross3(u, duv, ud);
      const double c0 = qw * duv[0] + eku[k][0] + ud[0];
      const double c1 = qw * duv[1] + eku[k][1] + ud[1];
      const double c2k = qw * duv[2] + eku[k][2] + ud[2];
      row[4 + k] = g[0] * c0 + g[1] * c1 + g[2] * c2k;
    }
    int o = 0;
#pragma unroll
    for (int a = 0; a < 7; ++a)
#pragma unroll
      for (int b = a; b < 7; ++b) acc[o++] = row[a] * row[b];
#pragma unroll
    for (int a = 0; a < 7; ++a) acc[28 + a] = row[a] * r;
    acc[35] = r * r;
  }
  // wavefront butterfly, then 4 waves through LDS
  __shared__ double lds[kEvalThreads / kWave][kAcc];
#pragma unroll
  for (int k = 0; k < kAcc; ++k) {
    double s = acc[k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    acc[k] = s;
  }
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < kAcc; ++k) lds[wave][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < kAcc) {
    double s = 0.0;
#pragma unroll
    for (int wv = 0; wv < kEvalThreads / kWave; ++wv) s += lds[wv][threadIdx.x];
    partials[static_cast<size_t>(blockIdx.x) * kAcc + threadIdx.x] = s;
  }
}

// ------------------------------------------------------------------------------------------
// LM state machine
// ------------------------------------------------------------------------------------------
template <int N>
struct DJ {  // forward-mode dual number (ceres::Jet arithmetic)
  double a;
  double v[N];
};
template <int N> __device__ inline DJ<N> dj_const(double a) {
  DJ<N> r; r.a = a;
  for (int i = 0; i < N; ++i) r.v[i] = 0.0;
  return r;
}
template <int N> __device__ inline DJ<N> dj_var(double a, int k) {
  DJ<N> r = dj_const<N>(a);
  r.v[k] = 1.0;
  return r;
}
template <int N> __device__ inline DJ<N> operator+(const DJ<N>& f, const DJ<N>& g) {
  DJ<N> r; r.a = f.a + g.a;
  for (int i = 0; i < N; ++i) r.v[i] = f.v[i] + g.v[i];
  return r;
}
template <int N> __device__ inline DJ<N> operator-(const DJ<N>& f, const DJ<N>& g) {
  DJ<N> r; r.a = f.a - g.a;
  for (int i = 0; i < N; ++i) r.v[i] = f.v[i] - g.v[i];
  return r;
}
template <int N> __device__ inline DJ<N> operator-(const DJ<N>& f) {
  DJ<N> r; r.a = -f.a;
  for (int i = 0; i < N; ++i) r.v[i] = -f.v[i];
  return r;
}
template <int N> __device__ inline DJ<N> operator*(const DJ<N>& f, const DJ<N>& g) {
  DJ<N> r; r.a = f.a * g.a;
  for (int i = 0; i < N; ++i) r.v[i] = f.a * g.v[i] + f.v[i] * g.a;
  return r;
}
template <int N> __device__ inline DJ<N> operator/(const DJ<N>& f, const DJ<N>& g) {
  const double gi = 1.0 / g.a;
  const double fg = f.a * gi;
  DJ<N> r; r.a = fg;
  for (int i = 0; i < N; ++i) r.v[i] = (f.v[i] - fg * g.v[i]) * gi;
  return r;
}
template <int N> __device__ inline DJ<N> dj_sin(const DJ<N>& f) {
  const double c = cos(f.a);
  DJ<N> r; r.a = sin(f.a);
  for (int i = 0; i < N; ++i) r.v[i] = c * f.v[i];
  return r;
}
template <int N> __device__ inline DJ<N> dj_acos(const DJ<N>& f) {
  const double t = -1.0 / sqrt(1.0 - f.a * f.a);
  DJ<N> r; r.a = acos(f.a);
  for (int i = 0; i < N; ++i) r.v[i] = t * f.v[i];
  return r;
}

// QuaternionParameterization (Ceres 1.13 local_parameterization.cc)
__device__ inline void quaternion_plus(const double* x, const double* delta, double* out) {
  const double nd = sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
  if (nd > 0.0) {
    const double sbd = sin(nd) / nd;
    const double q0 = cos(nd), q1 = sbd * delta[0], q2 = sbd * delta[1], q3 = sbd * delta[2];
    out[0] = q0 * x[0] - q1 * x[1] - q2 * x[2] - q3 * x[3];
    out[1] = q0 * x[1] + q1 * x[0] + q2 * x[3] - q3 * x[2];
    out[2] = q0 * x[2] - q1 * x[3] + q2 * x[0] + q3 * x[1];
    out[3] = q0 * x[3] + q1 * x[2] - q2 * x[1] + q3 * x[0];
  } else {
    for (int i = 0; i < 4; ++i) out[i] = x[i];
  }
}
__device__ inline void quaternion_plus_jacobian(const double* x, double* j /*4x3*/) {
  j[0] = -x[1]; j[1] = -x[2]; j[2] = -x[3];
  j[3] = x[0];  j[4] = x[3];  j[5] = -x[2];
  j[6] = -x[3]; j[7] = x[0];  j[8] = x[1];
  j[9] = x[2];  j[10] = -x[1]; j[11] = x[0];
}

// Block transform and its derivative w.r.t. the local parameters of its pose(s) at `poses`.
// Single pose: T = pose_a. Two poses: InterpolateTransform (transform/timestamped_transform.h:41-51)
// = lerp of translations + Eigen 3.3 Quaternion::slerp.
__device__ void prepare_block(const BlockInfo& b, const double (*poses)[7], BlockXform* xf) {
  for (int i = 0; i < 7 * 12; ++i) xf->M[i] = 0.0;
  const double* pa = poses[b.pose_a];
  double pja[12];
  quaternion_plus_jacobian(pa + 3, pja);
  if (b.pose_b < 0) {
    for (int k = 0; k < 3; ++k) xf->t[k] = pa[k];
    for (int k = 0; k < 4; ++k) xf->q[k] = pa[3 + k];
    for (int k = 0; k < 3; ++k) xf->M[k * 12 + k] = 1.0;
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 3; ++c) xf->M[(3 + r) * 12 + 3 + c] = pja[r * 3 + c];
    return;
  }
  const double* pb = poses[b.pose_b];
  double pjb[12];
  quaternion_plus_jacobian(pb + 3, pjb);
  const double f = b.factor;
  for (int k = 0; k < 3; ++k) {
    xf->t[k] = pa[k] + (pb[k] - pa[k]) * f;
    xf->M[k * 12 + k] = 1.0 + (0.0 - 1.0) * f;
    xf->M[k * 12 + 6 + k] = (1.0 - 0.0) * f;
  }
  typedef DJ<8> J;
  // quaternion coefficients as variables: a = (w,x,y,z) -> 0..3, b -> 4..7
  J aw = dj_var<8>(pa[3], 0), ax = dj_var<8>(pa[4], 1), ay = dj_var<8>(pa[5], 2), az = dj_var<8>(pa[6], 3);
  J bw = dj_var<8>(pb[3], 4), bx = dj_var<8>(pb[4], 5), by = dj_var<8>(pb[5], 6), bz = dj_var<8>(pb[6], 7);
  const J t = dj_const<8>(f);
  const double one = 1.0 - 2.220446049250313e-16;
  const J d = (ax * bx + ay * by) + (az * bz + aw * bw);
  const J absd = d.a < 0.0 ? -d : d;
  J s0, s1;
  if (absd.a >= one) {
    s0 = dj_const<8>(1.0) - t;
    s1 = t;
  } else {
    const J theta = dj_acos(absd);
    const J sin_theta = dj_sin(theta);
    s0 = dj_sin((dj_const<8>(1.0) - t) * theta) / sin_theta;
    s1 = dj_sin(t * theta) / sin_theta;
  }
  if (d.a < 0.0) s1 = -s1;
  const J q[4] = {s0 * aw + s1 * bw, s0 * ax + s1 * bx, s0 * ay + s1 * by, s0 * az + s1 * bz};
  for (int r = 0; r < 4; ++r) {
    xf->q[r] = q[r].a;
    for (int c = 0; c < 3; ++c) {
      double sa = 0.0, sb = 0.0;
      for (int j = 0; j < 4; ++j) {
        sa += q[r].v[j] * pja[j * 3 + c];
        sb += q[r].v[4 + j] * pjb[j * 3 + c];
      }
      xf->M[(3 + r) * 12 + 3 + c] = sa;
      xf->M[(3 + r) * 12 + 9 + c] = sb;
    }
  }
}

__device__ inline void pose_plus(const LmState* S, const double (*x)[7], const double* delta,
                                 double (*out)[7]) {
  for (int p = 0; p < S->num_poses; ++p) {
    if (S->constant[p]) {
      for (int k = 0; k < 7; ++k) out[p][k] = x[p][k];
      continue;
    }
    const double* d = delta + S->col[p];
    for (int k = 0; k < 3; ++k) out[p][k] = x[p][k] + d[k];
    quaternion_plus(x[p] + 3, d + 3, out[p] + 3);
  }
}

__device__ bool cholesky_solve(int n, double* A /*n x n, destroyed*/, const double* b, double* x,
                               double* y) {
  for (int j = 0; j < n; ++j) {
    double d = A[j * n + j];
    for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0.0) || !isfinite(d)) return false;
    const double l = sqrt(d);
    A[j * n + j] = l;
    for (int i = j + 1; i < n; ++i) {
      double s = A[i * n + j];
      for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = s / l;
    }
  }
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= A[i * n + k] * y[k];
    y[i] = s / A[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < n; ++k) s -= A[k * n + i] * x[k];
    x[i] = s / A[i * n + i];
  }
  for (int i = 0; i < n; ++i)
    if (!isfinite(x[i])) return false;
  return true;
}

__device__ inline void finish(LmState* S, int type, int reason) {
  S->done = 1;
  S->termination_type = type;
  S->termination_reason = reason;
}

// |x - Plus(x, -g)|_inf in the ambient space (TrustRegionMinimizer::EvaluateGradientAndJacobian)
__device__ double gradient_max_norm(const LmState* S) {
  double m = 0.0;
  for (int p = 0; p < S->num_poses; ++p) {
    if (S->constant[p]) continue;
    const double* g = S->g + S->col[p];
    const double neg[6] = {-g[0], -g[1], -g[2], -g[3], -g[4], -g[5]};
    for (int k = 0; k < 3; ++k) m = fmax(m, fabs(S->x[p][k] - (S->x[p][k] + neg[k])));
    double q[4];
    quaternion_plus(S->x[p] + 3, neg + 3, q);
    for (int k = 0; k < 4; ++k) m = fmax(m, fabs(S->x[p][3 + k] - q[k]));
  }
  return m;
}

// LevenbergMarquardtStrategy::ComputeStep + TrustRegionMinimizer::ComputeTrustRegionStep,
// looping over invalid steps (each one is an iteration). Leaves the next candidate in S->cand
// or terminates.
__device__ void compute_next_candidate(LmState* S) {
  const int n = S->ncols;
  const hg_solver_opts& o = S->opt;
  while (true) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (S->step_is_successful) ++S->num_successful; else ++S->num_unsuccessful;
    if (S->iteration >= o.max_num_iterations) return finish(S, 1, 4);
    if (S->step_is_successful && S->gradient_max_norm <= o.gradient_tolerance) return finish(S, 0, 1);
    if (S->radius <= o.min_trust_region_radius) return finish(S, 0, 5);
    ++S->iteration;
    ++S->num_iterations;
    S->step_is_successful = 0;
    if (!S->reuse_diagonal) {
      for (int k = 0; k < n; ++k) {
        const double s = S->H[k * n + k] * S->scale[k] * S->scale[k];
        S->diagonal[k] = fmin(fmax(s, o.min_lm_diagonal), o.max_lm_diagonal);
      }
    }
    double* A = S->work;
    double* rhs = S->delta;  // reused below
    for (int a = 0; a < n; ++a) {
      for (int b = 0; b < n; ++b) A[a * n + b] = S->H[a * n + b] * S->scale[a] * S->scale[b];
      const double lm = sqrt(S->diagonal[a] / S->radius);
      A[a * n + a] += lm * lm;
      rhs[a] = S->g[a] * S->scale[a];
    }
    double y[kMaxCols];
    bool valid = cholesky_solve(n, A, rhs, S->step, y);
    S->reuse_diagonal = 1;
    double mcc = 0.0;
    if (valid) {
      for (int k = 0; k < n; ++k) S->step[k] = -S->step[k];
      // model_cost_change = -(step.J^T r + step^T J^T J step / 2) on the scaled system
      double lin = 0.0, quad = 0.0;
      for (int a = 0; a < n; ++a) {
        lin += S->step[a] * S->g[a] * S->scale[a];
        double row = 0.0;
        for (int b = 0; b < n; ++b) row += S->H[a * n + b] * S->scale[a] * S->scale[b] * S->step[b];
        quad += S->step[a] * row;
      }
      mcc = -(lin + 0.5 * quad);
      valid = mcc > 0.0;
    }
    if (!valid) {
      if (++S->invalid_steps >= 5) return finish(S, 2, 6);  // max_num_consecutive_invalid_steps
      S->radius = S->radius / S->decrease_factor;
      S->decrease_factor *= 2.0;
      S->reuse_diagonal = 1;
      continue;
    }
    S->invalid_steps = 0;
    S->model_cost_change = mcc;
    for (int k = 0; k < n; ++k) S->delta[k] = S->step[k] * S->scale[k];
    pose_plus(S, S->x, S->delta, S->cand);
    return;
  }
}

// Sums the workgroup partials of every block (all threads), then thread 0 maps them through
// M into Hc / gc / cand_cost.
__device__ void assemble(LmState* S, const BlockXform* xf, const double* partials,
                         double* sums /*LDS [kMaxBlocks*kAcc]*/, double* stripe /*LDS [4][kMaxBlocks*kAcc]*/) {
  const int n = S->ncols;
  // thread (k, j): column k of the 36 sums, stripe j of the workgroup partials; fixed order
  const int stripes = blockDim.x / 64;
  const int k = threadIdx.x % 64, j = threadIdx.x / 64;
  for (int b = 0; b < S->num_blocks; ++b) {
    const BlockInfo& bi = S->blocks[b];
    if (k < kAcc) {
      double s = 0.0;
      if (bi.active)
        for (unsigned wgi = j; wgi < bi.num_wg; wgi += stripes)
          s += partials[(static_cast<size_t>(bi.partial_offset) + wgi) * kAcc + k];
      stripe[(j * kMaxBlocks + b) * kAcc + k] = s;
    }
  }
  __syncthreads();
  for (int b = 0; b < S->num_blocks; ++b) {
    if (threadIdx.x < kAcc) {
      double s = 0.0;
      for (int jj = 0; jj < stripes; ++jj) s += stripe[(jj * kMaxBlocks + b) * kAcc + threadIdx.x];
      sums[b * kAcc + threadIdx.x] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  for (int i = 0; i < n * n; ++i) S->Hc[i] = 0.0;
  for (int i = 0; i < n; ++i) S->gc[i] = 0.0;
  double cost = 0.0;
  for (int b = 0; b < S->num_blocks; ++b) {
    const BlockInfo& bi = S->blocks[b];
    if (!bi.active) continue;
    const double* sm = sums + b * kAcc;
    double A7[7][7];
    int o = 0;
    for (int a = 0; a < 7; ++a)
      for (int c = a; c < 7; ++c) {
        A7[a][c] = sm[o];
        A7[c][a] = sm[o];
        ++o;
      }
    cost += sm[35];
    const double* M = xf[b].M;
    // columns of this block: pose_a -> M[:,0:6], pose_b -> M[:,6:12]
    int cols[12];
    for (int c = 0; c < 12; ++c) cols[c] = -1;
    if (!S->constant[bi.pose_a])
      for (int c = 0; c < 6; ++c) cols[c] = S->col[bi.pose_a] + c;
    if (bi.pose_b >= 0 && !S->constant[bi.pose_b])
      for (int c = 0; c < 6; ++c) cols[6 + c] = S->col[bi.pose_b] + c;
    double AM[7][12];
    for (int a = 0; a < 7; ++a)
      for (int c = 0; c < 12; ++c) {
        double s = 0.0;
        for (int k = 0; k < 7; ++k) s += A7[a][k] * M[k * 12 + c];
        AM[a][c] = s;
      }
    for (int c1 = 0; c1 < 12; ++c1) {
      if (cols[c1] < 0) continue;
      double gsum = 0.0;
      for (int k = 0; k < 7; ++k) gsum += M[k * 12 + c1] * sm[28 + k];
      S->gc[cols[c1]] += gsum;
      for (int c2 = 0; c2 < 12; ++c2) {
        if (cols[c2] < 0) continue;
        double s = 0.0;
        for (int k = 0; k < 7; ++k) s += M[k * 12 + c1] * AM[k][c2];
        S->Hc[cols[c1] * n + cols[c2]] += s;
      }
    }
  }
  S->cand_cost = 0.5 * cost;
}

__global__ __launch_bounds__(256) void k_lm(LmState* S, BlockXform* xf, const double* partials, int mode) {
  __shared__ double sums[kMaxBlocks * kAcc];
  __shared__ double stripe[4 * kMaxBlocks * kAcc];
  if (S->done && mode == MODE_STEP) return;
  if (mode == MODE_PREPARE) {
    // transforms of every block at S->cand
    if (threadIdx.x < S->num_blocks) prepare_block(S->blocks[threadIdx.x], S->cand, &xf[threadIdx.x]);
    return;
  }
  assemble(S, xf, partials, sums, stripe);
  __syncthreads();
  if (mode == MODE_ASSEMBLE) return;
  if (threadIdx.x == 0) {
    const int n = S->ncols;
    const hg_solver_opts& o = S->opt;
    if (S->phase == PHASE_INIT) {
      // IterationZero: EvaluateGradientAndJacobian at x (= cand)
      ++S->num_cost_evals;
      ++S->num_jac_evals;
      S->x_cost = S->cand_cost;
      S->initial_cost = S->cand_cost;
      for (int i = 0; i < n * n; ++i) S->H[i] = S->Hc[i];
      for (int i = 0; i < n; ++i) S->g[i] = S->gc[i];
      for (int k = 0; k < n; ++k)
        S->scale[k] = o.jacobi_scaling ? 1.0 / (1.0 + sqrt(S->H[k * n + k])) : 1.0;
      S->gradient_max_norm = gradient_max_norm(S);
      S->step_is_successful = 1;
      S->num_iterations = 1;
      S->phase = PHASE_CANDIDATE;
      compute_next_candidate(S);
    } else {
      // candidate evaluated
      ++S->num_cost_evals;
      // ParameterToleranceReached
      double sn = 0.0, xn = 0.0;
      for (int p = 0; p < S->num_poses; ++p) {
        if (S->constant[p]) continue;
        for (int k = 0; k < 7; ++k) {
          const double d = S->x[p][k] - S->cand[p][k];
          sn += d * d;
          xn += S->x[p][k] * S->x[p][k];
        }
      }
      sn = sqrt(sn);
      xn = sqrt(xn);
      if (sn <= o.parameter_tolerance * (xn + o.parameter_tolerance)) {
        finish(S, 0, 2);
      } else {
        const double cost_change = S->x_cost - S->cand_cost;
        if (fabs(cost_change) <= o.function_tolerance * S->x_cost) {
          finish(S, 0, 3);
        } else {
          const double relative_decrease = cost_change / S->model_cost_change;
          if (relative_decrease > o.min_relative_decrease) {
            // HandleSuccessfulStep: the candidate's normal equations become x's
            for (int p = 0; p < S->num_poses; ++p)
              for (int k = 0; k < 7; ++k) S->x[p][k] = S->cand[p][k];
            S->x_cost = S->cand_cost;
            for (int i = 0; i < n * n; ++i) S->H[i] = S->Hc[i];
            for (int i = 0; i < n; ++i) S->g[i] = S->gc[i];
            ++S->num_jac_evals;
            S->gradient_max_norm = gradient_max_norm(S);
            S->step_is_successful = 1;
            S->radius = S->radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * relative_decrease - 1.0, 3.0));
            S->radius = fmin(o.max_trust_region_radius, S->radius);
            S->decrease_factor = 2.0;
            S->reuse_diagonal = 0;
          } else {
            S->radius = S->radius / S->decrease_factor;
            S->decrease_factor *= 2.0;
            S->reuse_diagonal = 1;
          }
          compute_next_candidate(S);
        }
      }
    }
  }
  __syncthreads();
  if (!S->done && threadIdx.x < S->num_blocks)
    prepare_block(S->blocks[threadIdx.x], S->cand, &xf[threadIdx.x]);
}

}  // namespace hg

using namespace hg;

struct hg_problem {
  hg_ctx* ctx = nullptr;
  struct Block {
    const float* d_xyz = nullptr;
    void* owned = nullptr;
    size_t n = 0;
    std::vector<hg_grid*> pyramid;
    int multi_res = 0;
    double scaling = 1.0;
    int pose_a = 0, pose_b = -1;
    double factor = 0.0;
  };
  std::vector<Block> blocks;
  std::vector<std::array<double, 7>> poses;
  std::vector<int> constant;
  // device state
  LmState* d_state = nullptr;
  BlockXform* d_xf = nullptr;
  DeviceBuffer partials, residuals;
  LmState h_state;  // staging
};

namespace {

bool block_active(const hg_problem* p, const hg_problem::Block& b) {
  if (b.n == 0) return false;
  if (!p->constant[b.pose_a]) return true;
  return b.pose_b >= 0 && !p->constant[b.pose_b];
}

// Fills h_state's static part and uploads it. cand = x = current poses.
int upload_state(hg_problem* p, const hg_solver_opts* opts) {
  LmState& S = p->h_state;
  std::memset(&S, 0, sizeof(S));
  S.num_poses = static_cast<int>(p->poses.size());
  S.num_blocks = static_cast<int>(p->blocks.size());
  int col = 0;
  for (int i = 0; i < S.num_poses; ++i) {
    for (int k = 0; k < 7; ++k) S.x[i][k] = S.cand[i][k] = p->poses[i][k];
    S.constant[i] = p->constant[i];
    S.col[i] = p->constant[i] ? -1 : col;
    if (!p->constant[i]) col += 6;
  }
  S.ncols = col;
  if (opts) S.opt = *opts; else hg_solver_default_opts(&S.opt);
  S.radius = S.opt.initial_trust_region_radius;
  S.decrease_factor = 2.0;
  S.phase = PHASE_INIT;
  unsigned wg_off = 0, row = 0;
  for (int b = 0; b < S.num_blocks; ++b) {
    const hg_problem::Block& hb = p->blocks[b];
    BlockInfo& bi = S.blocks[b];
    bi.pose_a = hb.pose_a;
    bi.pose_b = hb.pose_b;
    bi.factor = hb.factor;
    bi.scaling = hb.scaling;
    bi.n = static_cast<unsigned>(hb.n);
    bi.active = block_active(p, hb) ? 1 : 0;
    bi.num_wg = bi.active ? (bi.n + kEvalThreads - 1) / kEvalThreads : 0;
    bi.partial_offset = wg_off;
    bi.row_offset = row;
    wg_off += bi.num_wg;
    if (bi.active) row += bi.n;
  }
  int rc = p->partials.reserve(static_cast<size_t>(std::max(1u, wg_off)) * kAcc * sizeof(double));
  if (rc != HG_OK) return rc;
  HG_HIP_CHECK(hipMemcpyAsync(p->d_state, &S, sizeof(S), hipMemcpyHostToDevice, p->ctx->stream));
  return HG_OK;
}

int launch_eval(hg_problem* p, double* d_residuals, bool check_done) {
  hipStream_t s = p->ctx->stream;
  const LmState& S = p->h_state;
  for (int b = 0; b < S.num_blocks; ++b) {
    const BlockInfo& bi = S.blocks[b];
    if (!bi.active) continue;
    const hg_problem::Block& hb = p->blocks[b];
    PyramidView pv;
    std::memset(&pv, 0, sizeof(pv));
    pv.levels = static_cast<int>(hb.pyramid.size());
    pv.multi_res = hb.multi_res;
    for (int l = 0; l < pv.levels; ++l) pv.level[l] = hb.pyramid[l]->view;
    ProfScope ps(p->ctx, HG_K_RESIDUALS, bi.n);
    hipLaunchKernelGGL(k_tsdf_residuals, dim3(bi.num_wg), dim3(kEvalThreads), 0, s, pv, hb.d_xyz,
                       bi.n, bi.scaling, p->d_xf + b,
                       p->partials.as<double>() + static_cast<size_t>(bi.partial_offset) * kAcc,
                       d_residuals ? d_residuals + bi.row_offset : nullptr,
                       check_done ? &p->d_state->done : nullptr);
    HG_HIP_CHECK(hipGetLastError());
  }
  return HG_OK;
}

}  // namespace

extern "C" {

int hg_solver_default_opts(hg_solver_opts* o) {
  if (!o) return HG_ERR_INVALID;
  o->max_num_iterations = 12;  // configuration_files/trajectory_builder_3d.lua:51
  o->jacobi_scaling = 1;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  return HG_OK;
}

int hg_problem_create(hg_ctx* ctx, hg_problem** out) {
  if (!ctx || !out) return HG_ERR_INVALID;
  *out = nullptr;
  HG_HIP_CHECK(hipSetDevice(ctx->device));
  hg_problem* p = new hg_problem();
  p->ctx = ctx;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&p->d_state), sizeof(LmState));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_xf), sizeof(BlockXform) * kMaxBlocks);
  if (e != hipSuccess) {
    set_last_error(std::string("hipMalloc problem: ") + hipGetErrorString(e));
    hg_problem_destroy(p);
    return HG_ERR_HIP;
  }
  *out = p;
  return HG_OK;
}

int hg_problem_destroy(hg_problem* p) {
  if (!p) return HG_ERR_INVALID;
  (void)hipSetDevice(p->ctx->device);
  (void)hipStreamSynchronize(p->ctx->stream);
  for (auto& b : p->blocks)
    if (b.owned) (void)hipFree(b.owned);
  if (p->d_state) (void)hipFree(p->d_state);
  if (p->d_xf) (void)hipFree(p->d_xf);
  p->partials.release();
  p->residuals.release();
  delete p;
  return HG_OK;
}

int hg_problem_reset(hg_problem* p) {
  if (!p) return HG_ERR_INVALID;
  (void)hipStreamSynchronize(p->ctx->stream);
  for (auto& b : p->blocks)
    if (b.owned) (void)hipFree(b.owned);
  p->blocks.clear();
  p->poses.clear();
  p->constant.clear();
  return HG_OK;
}

int hg_problem_add_pose(hg_problem* p, const double tq[7], int constant) {
  if (!p || !tq) return HG_ERR_INVALID;
  if (p->poses.size() >= static_cast<size_t>(kMaxPoses)) {
    set_last_error("too many pose blocks");
    return HG_ERR_CAPACITY;
  }
  std::array<double, 7> a;
  std::memcpy(a.data(), tq, sizeof(double) * 7);
  p->poses.push_back(a);
  p->constant.push_back(constant ? 1 : 0);
  return static_cast<int>(p->poses.size()) - 1;
}

int hg_problem_set_pose(hg_problem* p, int index, const double tq[7]) {
  if (!p || !tq || index < 0 || index >= static_cast<int>(p->poses.size())) return HG_ERR_INVALID;
  std::memcpy(p->poses[index].data(), tq, sizeof(double) * 7);
  return HG_OK;
}

int hg_problem_get_pose(hg_problem* p, int index, double tq[7]) {
  if (!p || !tq || index < 0 || index >= static_cast<int>(p->poses.size())) return HG_ERR_INVALID;
  std::memcpy(tq, p->poses[index].data(), sizeof(double) * 7);
  return HG_OK;
}

int hg_problem_add_block(hg_problem* p, const float* xyz, size_t n, int memspace,
                         hg_grid* const* pyramid, int levels, int multi_res, double scaling_factor,
                         int pose_a, int pose_b, double interpolation_ratio) {
  if (!p || !pyramid || levels < 1 || levels > kMaxLevels || (n && !xyz)) return HG_ERR_INVALID;
  const int np = static_cast<int>(p->poses.size());
  if (pose_a < 0 || pose_a >= np || pose_b >= np) return HG_ERR_INVALID;
  if (p->blocks.size() >= static_cast<size_t>(kMaxBlocks)) {
    set_last_error("too many residual blocks");
    return HG_ERR_CAPACITY;
  }
  if (n > 0xFFFFFFFFull) return HG_ERR_INVALID;
  hg_problem::Block b;
  for (int l = 0; l < levels; ++l) {
    if (!pyramid[l] || pyramid[l]->ctx != p->ctx) return HG_ERR_INVALID;
    // InterpolatedMultiResolutionTSDF ctor CHECK: ascending resolution (:62-67)
    if (l > 0 && !(pyramid[l - 1]->view.resolution < pyramid[l]->view.resolution)) {
      set_last_error("TSDF pyramid must be sorted by ascending voxel size");
      return HG_ERR_INVALID;
    }
    b.pyramid.push_back(pyramid[l]);
  }
  b.n = n;
  b.multi_res = multi_res ? 1 : 0;
  b.scaling = scaling_factor;
  b.pose_a = pose_a;
  b.pose_b = pose_b < 0 ? -1 : pose_b;
  b.factor = interpolation_ratio;
  if (memspace == HG_HOST && n) {
    HG_HIP_CHECK(hipSetDevice(p->ctx->device));
    HG_HIP_CHECK(hipMalloc(&b.owned, n * 3 * sizeof(float)));
    hipError_t e = hipMemcpyAsync(b.owned, xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice, p->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->ctx->stream);
    if (e != hipSuccess) {
      (void)hipFree(b.owned);
      set_last_error(std::string("upload points: ") + hipGetErrorString(e));
      return HG_ERR_HIP;
    }
    b.d_xyz = static_cast<const float*>(b.owned);
  } else {
    b.d_xyz = xyz;
  }
  p->blocks.push_back(b);
  return static_cast<int>(p->blocks.size()) - 1;
}

int hg_problem_num_residuals(hg_problem* p) {
  if (!p) return HG_ERR_INVALID;
  size_t n = 0;
  for (const auto& b : p->blocks)
    if (block_active(p, b)) n += b.n;
  return static_cast<int>(n);
}

int hg_problem_num_columns(hg_problem* p) {
  if (!p) return HG_ERR_INVALID;
  int c = 0;
  for (int k : p->constant)
    if (!k) c += 6;
  return c;
}

int hg_problem_evaluate(hg_problem* p, double* cost, double* residuals, double* gradient, double* JtJ) {
  if (!p) return HG_ERR_INVALID;
  hipStream_t s = p->ctx->stream;
  HG_HIP_CHECK(hipSetDevice(p->ctx->device));
  int rc = upload_state(p, nullptr);
  if (rc != HG_OK) return rc;
  const int nres = hg_problem_num_residuals(p);
  double* d_res = nullptr;
  if (residuals && nres > 0) {
    rc = p->residuals.reserve(sizeof(double) * nres);
    if (rc != HG_OK) return rc;
    d_res = p->residuals.as<double>();
  }
  hipLaunchKernelGGL(k_lm, dim3(1), dim3(256), 0, s, p->d_state, p->d_xf, p->partials.as<double>(), MODE_PREPARE);
  HG_HIP_CHECK(hipGetLastError());
  rc = launch_eval(p, d_res, false);
  if (rc != HG_OK) return rc;
  hipLaunchKernelGGL(k_lm, dim3(1), dim3(256), 0, s, p->d_state, p->d_xf, p->partials.as<double>(), MODE_ASSEMBLE);
  HG_HIP_CHECK(hipGetLastError());
  HG_HIP_CHECK(hipMemcpyAsync(&p->h_state, p->d_state, sizeof(LmState), hipMemcpyDeviceToHost, s));
  if (d_res) HG_HIP_CHECK(hipMemcpyAsync(residuals, d_res, sizeof(double) * nres, hipMemcpyDeviceToHost, s));
  HG_HIP_CHECK(hipStreamSynchronize(s));
  const LmState& S = p->h_state;
  if (cost) *cost = S.cand_cost;
  if (gradient) std::memcpy(gradient, S.gc, sizeof(double) * S.ncols);
  if (JtJ) std::memcpy(JtJ, S.Hc, sizeof(double) * S.ncols * S.ncols);
  return HG_OK;
}

int hg_problem_solve(hg_problem* p, const hg_solver_opts* opts, hg_solver_summary* summary) {
  if (!p) return HG_ERR_INVALID;
  hipStream_t s = p->ctx->stream;
  HG_HIP_CHECK(hipSetDevice(p->ctx->device));
  int rc = upload_state(p, opts);
  if (rc != HG_OK) return rc;
  const LmState& S0 = p->h_state;
  if (S0.ncols == 0) {
    if (summary) {
      std::memset(summary, 0, sizeof(*summary));
      summary->termination_type = 0;
    }
    return HG_OK;
  }
  const int max_it = S0.opt.max_num_iterations;
  hipLaunchKernelGGL(k_lm, dim3(1), dim3(256), 0, s, p->d_state, p->d_xf, p->partials.as<double>(), MODE_PREPARE);
  HG_HIP_CHECK(hipGetLastError());
  for (int it = 0; it <= max_it; ++it) {
    rc = launch_eval(p, nullptr, true);
    if (rc != HG_OK) return rc;
    {
      ProfScope ps(p->ctx, HG_K_LM, 1);
      hipLaunchKernelGGL(k_lm, dim3(1), dim3(256), 0, s, p->d_state, p->d_xf, p->partials.as<double>(), MODE_STEP);
    }
    HG_HIP_CHECK(hipGetLastError());
  }
  HG_HIP_CHECK(hipMemcpyAsync(&p->h_state, p->d_state, sizeof(LmState), hipMemcpyDeviceToHost, s));
  HG_HIP_CHECK(hipStreamSynchronize(s));
  const LmState& S = p->h_state;
  for (int i = 0; i < S.num_poses; ++i) std::memcpy(p->poses[i].data(), S.x[i], sizeof(double) * 7);
  if (summary) {
    summary->initial_cost = S.initial_cost;
    summary->final_cost = S.x_cost;
    summary->final_radius = S.radius;
    summary->num_iterations = S.num_iterations;
    summary->num_successful_steps = S.num_successful;
    summary->num_unsuccessful_steps = S.num_unsuccessful;
    summary->num_cost_evaluations = S.num_cost_evals;
    summary->num_jacobian_evaluations = S.num_jac_evals;
    summary->termination_type = S.done ? S.termination_type : 1;
    summary->termination_reason = S.done ? S.termination_reason : 4;
    summary->reserved = 0;
  }
  return HG_OK;
}

int hg_match_evaluate(hg_ctx* ctx, hg_grid* const* pyramid, int levels, int multi_res,
                      const float* xyz, size_t n, int memspace, double scaling_factor,
                      const double pose0[7], const double* pose1, double interpolation_ratio,
                      double* cost, double* JtJ, double* Jtr, double* residuals) {
  hg_problem* p = nullptr;
  int rc = hg_problem_create(ctx, &p);
  if (rc != HG_OK) return rc;
  const int a = hg_problem_add_pose(p, pose0, 0);
  const int b = pose1 ? hg_problem_add_pose(p, pose1, 0) : -1;
  rc = hg_problem_add_block(p, xyz, n, memspace, pyramid, levels, multi_res, scaling_factor, a, b,
                            interpolation_ratio);
  if (rc >= 0) rc = hg_problem_evaluate(p, cost, residuals, Jtr, JtJ);
  hg_problem_destroy(p);
  return rc;
}

int hg_match_solve(hg_ctx* ctx, hg_grid* const* pyramid, int levels, int multi_res,
                   const float* xyz, size_t n, int memspace, double scaling_factor,
                   double pose0[7], double* pose1, int pose0_constant, double interpolation_ratio,
                   const hg_solver_opts* opts, hg_solver_summary* summary) {
  hg_problem* p = nullptr;
  int rc = hg_problem_create(ctx, &p);
  if (rc != HG_OK) return rc;
  const int a = hg_problem_add_pose(p, pose0, pose0_constant);
  const int b = pose1 ? hg_problem_add_pose(p, pose1, 0) : -1;
  rc = hg_problem_add_block(p, xyz, n, memspace, pyramid, levels, multi_res, scaling_factor, a, b,
                            interpolation_ratio);
  if (rc >= 0) rc = hg_problem_solve(p, opts, summary);
  if (rc == HG_OK) {
    hg_problem_get_pose(p, a, pose0);
    if (pose1) hg_problem_get_pose(p, b, pose1);
  }
  hg_problem_destroy(p);
  return rc;
}

}  // extern "C"
