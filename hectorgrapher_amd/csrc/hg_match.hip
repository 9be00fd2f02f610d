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
//
// This file is compiled TWICE (csrc/Makefile). The plain build holds every entry point of the C ABI and a
// solver whose whole working set lives in LDS: windows of up to 10 control points, 32 TSDF blocks and 20
// odometry / IMU blocks -- the sliding window of trajectory_builder_3d.lua's ct_window_horizon /
// ct_window_rate, and the single-pose registration step. The second build (-DHG_BIG, hg_match_big.o) is the
// same solver with larger limits -- 48 control points (432 columns), 160 TSDF blocks, 96 odometry / IMU
// blocks: the ADAPTIVE / SYNCED_WITH_RANGE_DATA control-point sampling (oltb.cc:1162-1232) and the two
// blocks per scan of use_multi_resolution_matching = false (oltb.cc:392-502) reach ~36 control points and
// ~70 blocks -- whose band matrices, local systems and index tables live in device memory (they no longer
// fit 160 KB of LDS) behind the same code. A problem that outgrows the plain limits is handed to the big
// build by the entry points below (promote()); nothing changes for the problems that fit.
#ifdef HG_BIG
#include "hg_match_big_names.h"  // every global name this file defines gets a _big twin
#endif
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>

#include "hg_internal.h"
#ifndef HG_BIG
#include "hg_match_big.h"  // the big build's entry points (same signatures on hg_problem_big)
#endif

#ifdef HG_BIG
struct hg_problem;  // (= hg_problem_big here)
namespace hg {
const double* big_device_poses(hg_problem* p, int* stride);
void big_orphan(hg_problem* p);
}
#define HG_CAP_NS_OPEN namespace hg { inline namespace big {
#define HG_CAP_NS_CLOSE } }
#else
#define HG_CAP_NS_OPEN namespace hg {
#define HG_CAP_NS_CLOSE }
#endif

HG_CAP_NS_OPEN

constexpr int kMaxLevels = 4;
constexpr int kState = 10;              // per control point: t(3) q(4) v(3)
#ifdef HG_BIG
constexpr int kMaxPoses = 48;
constexpr int kMaxSmall = 96;
constexpr int kMaxBlocks = 160;
#else
constexpr int kMaxPoses = 10;           // ct_window_horizon / ct_window_rate ~ 9 control points
constexpr int kMaxSmall = 20;           // odometry / IMU blocks
constexpr int kMaxBlocks = 32;
#endif
constexpr int kMaxCols = 9 * kMaxPoses;  // 6 pose + 3 velocity columns per control point
// The normal equations are stored as a symmetric band (lower part, row i holds columns
// i-bw..i): blocks couple neighbouring control points only, so the system is block-tridiagonal
// (SURVEY 8a16). n * (bw + 1) <= kHCap: a 90-column window with bw <= 35, or dense up to 56 x 56
// (big build: 432 columns with bw <= 35).
#ifdef HG_BIG
constexpr int kHCap = kMaxCols * 36;
#else
constexpr int kHCap = 3240;
#endif
#ifndef HG_BIG
// Limits of the big build (the plain build checks the caller's adds against them and promotes beyond its own).
constexpr int kBigMaxPoses = 48, kBigMaxSmall = 96, kBigMaxBlocks = 160;
#endif
constexpr int kAcc = 36;  // 28 (upper triangle of 7x7) + 7 + 1
constexpr int kAccU = 91;  // unwarped blocks: 78 (upper triangle of 12x12) + 12 + 1
constexpr int kEvalThreads = 512;
constexpr int kBatchThreads = 256;   // workgroup of the batched residual pass (no LM tail: lean in registers)
constexpr int kLmThreads = 64;   // the LM state machine runs in one wavefront
constexpr int kLmBlock = 512;    // waves 1-7 only help summing the workgroup partials
constexpr int kMaxTiles = 16;    // 256-return tiles one workgroup of the window pass walks at most

struct PyramidView {
  GridView level[kMaxLevels];
  int levels;
  int multi_res;
  const PyramidView* self_mem;  // a copy of this struct in device memory (for code that takes it by address)
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
  unsigned partial_offset;  // in doubles
  unsigned row_offset;
  int active;
  int acc;  // partial sums per workgroup: kAcc, or kAccU for a block with per-return factors
};

enum { PHASE_INIT = 0, PHASE_CANDIDATE = 1 };
enum { MODE_PREPARE = 0, MODE_STEP = 1, MODE_ASSEMBLE = 2 };

// Odometry (RelativeTranslationAndYawCostFunction, 6 residuals) or IMU pre-integration
// (PredictionImuPreintegrationCostFunctor, 9 residuals) block between control points a and b.
struct SmallBlockDev {
  int type, a, b, active;
  unsigned row_offset, pad;
  double w[3];
  double dt;
  double delta[7];
};

// Per small block: its local normal equations over the columns [pose_a 6 | vel_a 3 | pose_b 6 | vel_b 3]:
// lower triangle of J^T J (row-major, entry (r, c), c <= r, at r (r + 1) / 2 + c), then J^T r (18) and
// r^T r -- written by the block's wavefront, staged into LDS as it stands by the LM step.
constexpr int kSmallTri = 18 * 19 / 2;
constexpr int kSmallLoc = kSmallTri + 18 + 1 + 2;  // padded to a multiple of 4 doubles
struct SmallOut {
  double v[kSmallLoc];
};

// Scalars and small vectors of the solver; lives in global memory between launches and in LDS
// while k_lm runs.
struct PinBox;
struct LmHead {
  int num_poses, num_blocks, ncols, done;
  int iteration, phase, step_is_successful, reuse_diagonal;
  int invalid_steps, termination_type, termination_reason, num_iterations;
  int num_successful, num_unsuccessful, num_cost_evals, num_jac_evals;
  hg_solver_opts opt;
  double radius, decrease_factor;
  double x_cost, cand_cost, model_cost_change, gradient_max_norm, initial_cost;
  double x[kMaxPoses][kState];
  double cand[kMaxPoses][kState];
  int constant[kMaxPoses];
  int col[kMaxPoses];
  int vfree[kMaxPoses];  // velocity is a free parameter block
  int vcol[kMaxPoses];
  int num_small;
  int bw;  // half bandwidth of J^T J (max column distance coupled by any block)
  // Block-tridiagonal partition of the columns (0 groups = not of that shape): group k = the free
  // columns of one control point (6 pose and / or 3 velocity columns, contiguous), and every block
  // couples a group with itself or with a NEIGHBOURING group -- the shape of every sliding window
  // (SURVEY 8a16). The factorisation then runs block by block in registers (cholesky_solve_btd).
  int btd_groups;
  int btd_uniform;  // 6 or 9: every group has that many columns and the band holds the full coupling of neighbours
  int btd_cr;       // uniform groups: 2 twisted factorisation (two wavefronts, from both ends), 1 cyclic reduction over the workgroup, 0 chain in one wavefront
  int btd_start[kMaxPoses + 1];
  int btd_size[kMaxPoses + 1];
  SmallBlockDev small[kMaxSmall];
  double scale[kMaxCols], diagonal[kMaxCols], g[kMaxCols], step[kMaxCols], delta[kMaxCols];
  double gc[kMaxCols];
  BlockInfo blocks[kMaxBlocks];
  // host mailbox (mapped pinned memory): the step that terminates the solve stores the head there
  // and then `seq` into its flag word, so the host neither copies nor synchronises the stream
  struct PinBox* box;
  unsigned long long seq;
#ifdef HG_LM_STAMPS
  long long stamps[16];  // diagnostic build only: s_memtime at phase boundaries of the last LM step
#endif
};

// Index tables of a problem's structure, built once per solve (k_lm MODE_PREPARE) and staged into LDS by
// every LM step.
#ifdef HG_BIG
constexpr int kPairMax = 16;  // two blocks per scan (use_multi_resolution_matching = false) double the blocks of a pair
#else
constexpr int kPairMax = 8;
#endif
struct LmTables {
  int colcp[kMaxCols];    // global column -> control point * 16 + slot (0..5 pose, 6..8 velocity)
  int desc[kMaxBlocks + kMaxSmall];  // per block: active << 16 | (pose_b & 255) << 8 | pose_a (TSDF blocks, then small)
  // per pair of control points (p >= q): the blocks that cover both, in block order (TSDF blocks, then
  // small): block | small << 8 | local offset of p's columns << 12 | of q's << 20. A pair that more
  // than kPairMax blocks cover (per-point unwarping: dozens of blocks between two control points) sets
  // pair_overflow and the assembly scans all blocks per entry instead.
  int pair_list[kMaxPoses][kMaxPoses][kPairMax];
  int pair_count[kMaxPoses][kMaxPoses];
  int pair_overflow;
  int pad;
};

// Static gather lists of the assembly (built once per solve next to the tables): for every band entry and
// every column the positions of its <= kGatherMax contributions in the concatenation of the blocks' local
// systems [TSDF blocks: kMaxBlocks x kLoc | small blocks: kMaxSmall x kSmallLoc], in block order; 0xFFFF =
// none. An LM step then assembles the normal equations with independent loads straight from device
// memory -- no staging of the local systems, no per-entry search through the blocks.
constexpr int kGatherMax = kPairMax;
constexpr unsigned kGatherSmallBase = kMaxBlocks * (kAccU + 1);
constexpr int kCrStride = 5 * 81 + 3 * 9 + 5;  // cyclic-reduction workspace per group (cholesky_solve_cr)
struct LmState {
  LmHead h;
  double H[kHCap];   // J^T J at x (unscaled), band storage n x (bw + 1)
  double Hc[kHCap];  // J^T J at the candidate
  LmTables T;
  alignas(16) unsigned short gather_h[kHCap][kGatherMax];
  alignas(16) unsigned short gather_g[kMaxCols][kGatherMax];
#ifdef HG_BIG
  alignas(16) double A[kHCap + 2];     // the scaled system / its factor (LDS in the plain build), + dump slot + padding
  alignas(16) double cr_ws[kCrStride * kMaxPoses];  // cyclic-reduction workspace
#endif
};

struct PinBox {
  LmHead h;                 // uploaded state (host copy) and result of the terminating LM step
  unsigned long long flag;  // == h.seq once the result of solve `seq` is complete
  // compact upload read by k_lm MODE_PREPARE over the host link in ONE round trip: the non-zero
  // 8-byte words of the head and their word indices (~150 of 1170 words; small uncached host reads
  // are slow). The number of words is a kernel argument.
  unsigned long long up_val[sizeof(LmHead) / 8];
  unsigned up_idx[sizeof(LmHead) / 8];
};

#ifndef HG_BIG
typedef __attribute__((address_space(3))) double lds_f64;
#endif
// band storage: entry (i, j), j <= i, i - j <= bw, of a symmetric matrix; W = bw + 1
__device__ __host__ inline int band_index(int i, int j, int W) { return i * W + (W - 1) - (i - j); }
__device__ inline double band_get(const double* B, int i, int j, int W) {
  const int hi = i > j ? i : j, lo = i > j ? j : i;
  return (hi - lo < W) ? B[band_index(hi, lo, W)] : 0.0;
}

// ------------------------------------------------------------------------------------------
// per-return residual + row
// ------------------------------------------------------------------------------------------
struct D3 {  // value + gradient w.r.t. world (x, y, z); mirrors ceres::Jet<double, 3>
  double a, d0, d1, d2;
};

#ifdef HG_EVAL_STAMPS
// diagnostic build only: phase timeline of every workgroup's wave 0 in the residual body (100 MHz)
__device__ unsigned long long g_body_stamps[1024][8];
#define BODY_STAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_body_stamps[blockIdx.x][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BODY_STAMP(i) do {} while (0)
#endif
// Analysis build only (scripts/isa_budget.py: -DHG_ISA_REGIONS -S): comment markers between the phases of a
// return's evaluation, and only the 3-level doubling variant of the lookup instantiated, so that the hot path
// is one straight run of instructions that can be counted per phase.
#ifdef HG_ISA_REGIONS
#define ISA_MARK(name) asm volatile("; HG_REGION " name ::: "memory")
#else
#define ISA_MARK(name) do {} while (0)
#endif

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
    // fused multiply-adds, nested so that the terms that are structurally zero (a derivative slot no
    // stage has touched yet, or the slot of another axis) drop out exactly: the same bits as
    // interp_all_valid, which leaves those terms out
    const double da = q2.a - q1.a;
    q.a = fma(da, r.a, q1.a);
    q.d0 = fma(q2.d0 - q1.d0, r.a, fma(da, r.d0, q1.d0));
    q.d1 = fma(q2.d1 - q1.d1, r.a, fma(da, r.d1, q1.d1));
    q.d2 = fma(q2.d2 - q1.d2, r.a, fma(da, r.d2, q1.d2));
    w = w1 + w2;
  }
}

// Codec constants + geometry + codes of the level a lane interpolates on.
struct LevelSel {
  double x1, y1, z1, x2, y2, z2;
  uint32_t code[8];
  float tsd_scale, tsd_offset, min_tsd, weight_scale, weight_offset;
};

__device__ inline float sel_weight(const LevelSel& s, uint32_t code) {
  const uint32_t v = (code >> 16) & 0x7FFFu;
  return v == 0 ? 0.f : static_cast<float>(v) * s.weight_scale + s.weight_offset;
}
__device__ inline float sel_tsd(const LevelSel& s, uint32_t code) {
  const uint32_t v = code & 0x7FFFu;
  return v == 0 ? s.min_tsd : static_cast<float>(v) * s.tsd_scale + s.tsd_offset;
}

// InterpolatedTSDF::GetTSD (interpolated_tsdf.h:72-116) on the selected level; `both_invalid`
// is -0.3 (single resolution) or the level's getMinTSD() (multi resolution).
__device__ inline D3 interp_selected(const LevelSel& s, double both_invalid, double x, double y, double z) {
  double w[8];
  D3 q[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    w[c] = static_cast<double>(sel_weight(s, s.code[c]));
    q[c] = {static_cast<double>(sel_tsd(s, s.code[c])), 0.0, 0.0, 0.0};
  }
  // Jet / double: Ceres multiplies by the inverse (jet.h operator/(Jet, T))
  const double ix_inv = 1.0 / (s.x2 - s.x1), iy_inv = 1.0 / (s.y2 - s.y1), iz_inv = 1.0 / (s.z2 - s.z1);
  const D3 nx = {(x - s.x1) * ix_inv, ix_inv, 0.0, 0.0};
  const D3 ny = {(y - s.y1) * iy_inv, 0.0, iy_inv, 0.0};
  const D3 nz = {(z - s.z1) * iz_inv, 0.0, 0.0, iz_inv};
  D3 q11, q12, q21, q22, q1, q2, qq;
  double w11, w12, w21, w22, w1, w2, ww;
  interpolate_linear(both_invalid, q[0], q[1], w[0], w[1], nz, q11, w11);
  interpolate_linear(both_invalid, q[2], q[3], w[2], w[3], nz, q12, w12);
  interpolate_linear(both_invalid, q[4], q[5], w[4], w[5], nz, q21, w21);
  interpolate_linear(both_invalid, q[6], q[7], w[6], w[7], nz, q22, w22);
  interpolate_linear(both_invalid, q11, q12, w11, w12, ny, q1, w1);
  interpolate_linear(both_invalid, q21, q22, w21, w22, ny, q2, w2);
  interpolate_linear(both_invalid, q1, q2, w1, w2, nx, qq, ww);
  return qq;
}

// ------------------------------------------------------------------------------------------
// Direct-window lookups: one memory round trip per return.
//
// While every block of a grid sits in its direct slot (no block in the overflow area) and the
// blocks' bounding box fits the window (hg_device.h, GridView), the address of a voxel follows from
// its cell index alone: no hash probe, no key / tag compare. A query outside the window anchored at
// the bounding box minimum cannot hit a block (none exists there) and reads as unknown. The state of
// the window is read from the grid's counters by every wavefront (uniform loads): inserts of the
// same stream may have changed it since the host enqueued the launch, so only the device knows.
// ------------------------------------------------------------------------------------------
typedef unsigned su8 __attribute__((ext_vector_type(8)));
struct DirectRaw {  // counters [8..15] of every level: overflow blocks, bounding-box minimum x y z, maximum x y z, unused
  su8 w[kMaxLevels];  // in SCALAR registers (direct_issue)
};
struct DirectPyramid {
  uint32_t min_b[kMaxLevels][3];  // bounding-box minimum (block coordinates) = window anchor
  bool ok;                        // every level of the lookup is directly addressable
};

// Loads the counters (the point's load is issued in front of them and overlaps the wait).
// Round 4: SCALAR loads (s_load_dwordx8, one per level). As vector loads of one address in every lane they
// went through the vector L1 like any gather -- 16 quads x 6 instructions, a sixth of the tag lookups of a
// wavefront in the batched pass, which that unit bounds -- and came back through 24 v_readfirstlane. The
// scalar cache is invalidated at every kernel start, so the inserts of earlier launches are seen; the
// compiler cannot pick the scalar form itself (it cannot prove that no store of the kernel aliases the counters).
__device__ inline DirectRaw direct_issue(const PyramidView& pv) {
  DirectRaw r;
  const int levels = __builtin_amdgcn_readfirstlane(pv.multi_res ? pv.levels : 1);
  // the addresses are the same in every lane; spelled out for the places where the compiler cannot see that.
  // A level that is not looked up reads level 0's words and is overridden below.
  unsigned long long c[kMaxLevels];
#pragma unroll
  for (int l = 0; l < kMaxLevels; ++l) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(pv.level[l < levels ? l : 0].counters);
    const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(a)));
    const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(a >> 32)));
    c[l] = (static_cast<unsigned long long>(hi) << 32) | lo;
  }
  // One statement issues the loads AND waits for them: the compiler does not know that the registers are
  // still in flight behind a bare s_load and may copy them before the data lands (it did, in the window pass).
  // The wait overlaps the wavefront's first vector loads, which are issued before it.
  static_assert(kMaxLevels == 4, "four scalar loads");
  asm volatile(
      "s_load_dwordx8 %0, %4, 0x20\n\ts_load_dwordx8 %1, %5, 0x20\n\ts_load_dwordx8 %2, %6, 0x20\n\t"
      "s_load_dwordx8 %3, %7, 0x20\n\ts_waitcnt lgkmcnt(0)"
      : "=&s"(r.w[0]), "=&s"(r.w[1]), "=&s"(r.w[2]), "=&s"(r.w[3])
      : "s"(c[0]), "s"(c[1]), "s"(c[2]), "s"(c[3]));
#pragma unroll
  for (int l = 0; l < kMaxLevels; ++l)
    if (l >= levels) r.w[l][0] = 1u;  // never reported direct
  return r;
}
struct DirectFetch {
  float c[3];        // centre of the lower corner voxel per axis (CenterOfLowerVoxel)
  uint32_t s0[3];    // lower corner cell index + kIndexOffset per axis
  uint32_t byte[8];  // byte offsets of the 8 corners inside the direct area
  uint32_t code[8];  // corner order c = dx*4 + dy*2 + dz
  uint32_t face[4];  // the dx = 1 corners of a lane whose x pair crosses a block face (direct_merge)
  bool in;           // both cells of every axis lie in blocks of the window (direct_setup)
};

// Lower-corner cell and the addresses of the 8 corners of one level. The offsets are masked into
// the direct area, so the loads are safe wherever the point lies; whether they MEAN anything is
// decided afterwards (direct_accept).
// The fp32 quotient p / res as cell_index_fast forms it (the bits of the IEEE division).
__device__ inline float cell_quotient_fast(float p, float res, float r) {
  const float q0 = p * r;
  const float rem0 = __builtin_fmaf(-res, q0, p);
  const float q1 = __builtin_fmaf(rem0, r, q0);
  const float rem1 = __builtin_fmaf(-res, q1, p);
  return __builtin_fmaf(rem1, r, q1);
}
// `q` = the quotients p / resolution of the three coordinates (cell_quotient_fast), handed in so that the
// levels of a pyramid whose resolutions double can share them (pyramid_tsd_direct).
// What a direct lookup needs of one level, in scalar registers: fetched from the pyramid description in ONE
// group of scalar loads with one wait, before anything else of the lookup (pin_level). Left to the compiler the
// fields were fetched where they are used -- a dozen scalar loads strewn over the address arithmetic and the
// level selection, each followed by a full wait (and the window bits fetched twice).
struct LevelPin {
  uint32_t vox_lo, vox_hi;  // voxel pool
  uint32_t bits[3];         // window bits per axis
  float res, tsd_scale, tsd_offset;
};
__device__ inline LevelPin pin_level(const GridView& g) {
  LevelPin p;
  const unsigned long long v = reinterpret_cast<unsigned long long>(g.voxels);
  p.vox_lo = __builtin_amdgcn_readfirstlane(static_cast<int>(v));
  p.vox_hi = __builtin_amdgcn_readfirstlane(static_cast<int>(v >> 32));
#pragma unroll
  for (int a = 0; a < 3; ++a) p.bits[a] = __builtin_amdgcn_readfirstlane(static_cast<int>(g.dir_bits[a]));
  p.res = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(g.resolution)));
  p.tsd_scale = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(g.tsd_scale)));
  p.tsd_offset = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(g.tsd_offset)));
  asm volatile("" : "+s"(p.vox_lo), "+s"(p.vox_hi), "+s"(p.bits[0]), "+s"(p.bits[1]), "+s"(p.bits[2]),
                    "+s"(p.res), "+s"(p.tsd_scale), "+s"(p.tsd_offset));
  return p;
}
// a | b | c in one instruction (left to itself the compiler shares two-way ors between the corners: 12 instead of 8)
__device__ inline uint32_t or3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_or3_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// `wmin`: the window anchor of the level (pyramid_tsd_direct); `usable`: the point has a cell at all.
__device__ inline void direct_setup(const LevelPin& g, double x, double y, double z, const float* q,
                                    const uint32_t* wmin, bool usable, DirectFetch& f) {
  const float res = g.res;
  const double w[3] = {x, y, z};
  uint32_t off[3][2];  // BYTE offsets: block part | voxel part of one axis
  uint32_t shift = 9 + 2;
  bool inside = usable;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    // CenterOfLowerVoxel (interpolated_tsdf.h:176-192): float centre, compared against the double
    int i0 = round_to_int(q[a]);
    float c = static_cast<float>(i0) * res;
    // GetCellIndex of the lowered centre (:99-101 via GetWeight/GetTSD) is the index minus one:
    // (i * res -+ res) / res is within 1e-2 of an integer for |i| <= 8192, far from a rounding tie
    if (static_cast<double>(c) > w[a]) { c -= res; i0 -= 1; }
    f.c[a] = c;
    f.s0[a] = static_cast<uint32_t>(i0 + kIndexOffset);
    // Inside the window anchored at the bounding-box minimum (which implies inside the index range)? Outside it
    // no block exists and the voxels read as unknown. Blocks of cells s0 and s0 + 1 both within
    // [wmin, wmin + 2^bits): one unsigned compare of the cell against the window's first cell and its length
    // minus two (the operands are wave-uniform). Decided HERE, next to the address arithmetic that reads the same
    // window bits (round 4): left to the level selection behind the loads, the compiler fetched the bits again
    // from the pyramid description inside nested branches -- nine scalar loads, each awaited, per wavefront.
    const uint32_t first = wmin[a] << 3, span = (8u << g.bits[a]) - 2u;
    inside = inside & ((f.s0[a] - first) <= span);
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      // four instructions per offset (bit-field extract, shift, and, shift-or) and one three-way or per corner:
      // the address arithmetic was a quarter of the batched pass's vector instructions
      const uint32_t s = f.s0[a] + d;
      const uint32_t blk = __builtin_amdgcn_ubfe(s, 3u, g.bits[a]);  // (s >> 3) & mask
      // (the dx = 1 corners are only read by lanes whose x pair crosses a block face: their voxel bits are 0)
      off[a][d] = (a == 0 && d == 1) ? (blk << shift) : ((blk << shift) | ((s & 7u) << (3 * a + 2)));
    }
    shift += g.bits[a];
  }
#pragma unroll
  for (int c = 0; c < 8; ++c)  // direct area < 2^30 voxels
    f.byte[c] = or3(off[0][c >> 2], off[1][(c >> 1) & 1], off[2][c & 1]);
  f.in = inside;
}
// The voxel loads of one level (independent of each other and of the other levels'). The two x
// neighbours of a corner pair are adjacent words unless the pair crosses a block face (x & 7 == 7):
// one 8-byte load fetches both, and only the lanes on a face load their second voxel separately --
// half the requests the cache has to serve. The face loads land in registers of their own and are merged
// by direct_merge AFTER the loads of every level have been issued: written over the pair's second word
// right away (round 3), they made the compiler wait for the pair loads before issuing them -- some lane
// of nearly every wavefront sits on a face -- so the levels' round trips ran one after the other.
// Loads through the global address space with a 32-bit offset from the (uniform) pool base: one address
// register per load and no flat-aperture check; flat loads also count against lgkmcnt and complete out of
// order, which forces full waits.
typedef const __attribute__((address_space(1))) char* gmem_bytes;
__device__ inline gmem_bytes as_global(const void* p) {
  return reinterpret_cast<gmem_bytes>(reinterpret_cast<uintptr_t>(p));
}
__device__ inline void direct_load(const LevelPin& g, DirectFetch& f) {
  gmem_bytes base = reinterpret_cast<gmem_bytes>((static_cast<uintptr_t>(g.vox_hi) << 32) | g.vox_lo);
  const bool face = (f.s0[0] & 7u) == 7u;
#pragma unroll
  for (int k = 0; k < 4; ++k) {  // corner k = (dy, dz) at dx = 0, corner 4 + k at dx = 1
    typedef uint32_t u2 __attribute__((ext_vector_type(2), aligned(4)));
    const u2 v = *reinterpret_cast<const __attribute__((address_space(1))) u2*>(base + f.byte[k]);
    f.code[k] = v.x;
    f.code[4 + k] = v.y;
  }
  // (the face words stay undefined in the other lanes: direct_merge reads them under the same predicate)
  if (face) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      f.face[k] = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t*>(base + f.byte[4 + k]);
  }
}
// (Round 4, measured and dropped: the face words in compacted form -- an exec-masked gather costs the L1 its
// 16-cycle minimum however few lanes take part (scripts/tcp_bench.hip), so the face lanes of a level queued their
// four addresses in LDS and the first 4 x (face lanes) lanes fetched one word each: 3 dense loads instead of 12
// sparse ones, results routed back through LDS. Bit-identical, and slower everywhere -- 64-match batch 24.2k ->
// 23.6k matches/s, headline 3534 -> 3417 scans/s, window 1357 -> 1261: the two LDS round trips and the ballots
// sit on every wavefront's critical path, and the L1 is no longer the only bound.)
__device__ inline void direct_merge(DirectFetch& f) {
  const bool face = (f.s0[0] & 7u) == 7u;
  if (face) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // an opaque move: written as a plain select, the compiler loads the face word straight into the
      // pair's register under the face lanes' mask, which is the wait this function exists to avoid
      uint32_t t;
      asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(f.face[k]));
      f.code[4 + k] = t;
    }
  }
}
// All 8 weights non-zero: weight code (bits 16..30, the marker bit masked as GetWeight does) above 1.
__device__ inline bool all_weights_valid(const uint32_t* code) {
  constexpr uint32_t kAbove1 = 0x7FFE0000u;
  const uint32_t m0 = min(min(code[0] & kAbove1, code[1] & kAbove1), code[2] & kAbove1);
  const uint32_t m1 = min(min(code[3] & kAbove1, code[4] & kAbove1), code[5] & kAbove1);
  const uint32_t m2 = min(min(code[6] & kAbove1, code[7] & kAbove1), m0);
  return min(m1, m2) != 0u;
}
__device__ inline void direct_accept(DirectFetch& f) {
#pragma unroll
  for (int c = 0; c < 8; ++c) f.code[c] = f.in ? f.code[c] : 0u;
}

// InterpolatedTSDF::GetTSD (interpolated_tsdf.h:72-116) when all 8 weights are non-zero: every
// InterpolateLinear takes its interpolating branch, so the validity tests and the weights drop out.
// Same operations in the same order as interp_selected; terms that are exactly zero there (the
// derivative slots no stage has touched yet) are left out, which can only change the sign of a zero.
__device__ inline double rcp_span(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}
__device__ inline D3 interp_all_valid(float res, float tsd_scale, float tsd_offset,
                                      const float* c3, const uint32_t* code, double x, double y, double z) {
  double q[8];
#pragma unroll
  for (int c = 0; c < 8; c += 2) {
    // decode as the LUT does (float multiply, then float add: never fused). A voxel with a non-zero weight
    // code has a non-zero tsd code (SetCell writes both), so the unknown-code case cannot occur on a level
    // that is all valid; callers discard the result otherwise. Two codes per packed multiply / add.
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 t = {static_cast<float>(code[c] & 0x7FFFu), static_cast<float>(code[c + 1] & 0x7FFFu)};
    t = t * f2{tsd_scale, tsd_scale};
    t = t + f2{tsd_offset, tsd_offset};
    q[c] = static_cast<double>(t.x);
    q[c + 1] = static_cast<double>(t.y);
  }
  const double x1 = c3[0], y1 = c3[1], z1 = c3[2];
  const double x2 = c3[0] + res, y2 = c3[1] + res, z2 = c3[2] + res;  // float adds, as the reference
  // Jet / double: Ceres multiplies by the inverse (jet.h operator/(Jet, T)). The inverse of the voxel span
  // (a float sum minus a float: within a few ulps of the resolution, never tiny or huge) by two Newton steps
  // on v_rcp_f64 -- within an ulp of the IEEE quotient in 5 instructions instead of the division's 11;
  // continuous outputs only, as the lerps below.
  const double ix = rcp_span(x2 - x1), iy = rcp_span(y2 - y1), iz = rcp_span(z2 - z1);
  const double nx = (x - x1) * ix, ny = (y - y1) * iy, nz = (z - z1) * iz;
  // the lerps as fused multiply-adds (continuous outputs only: agreement with the reference to rounding)
  // along z: value and d/dz
  double a1[4], dz1[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double da = q[2 * k + 1] - q[2 * k];
    a1[k] = fma(da, nz, q[2 * k]);
    dz1[k] = da * iz;
  }
  // along y
  double a2[2], dy2[2], dz2[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const double da = a1[2 * k + 1] - a1[2 * k];
    a2[k] = fma(da, ny, a1[2 * k]);
    dy2[k] = da * iy;
    dz2[k] = fma(dz1[2 * k + 1] - dz1[2 * k], ny, dz1[2 * k]);
  }
  // along x
  const double da = a2[1] - a2[0];
  D3 r;
  r.a = fma(da, nx, a2[0]);
  r.d0 = da * ix;
  r.d1 = fma(dy2[1] - dy2[0], nx, dy2[0]);
  r.d2 = fma(dz2[1] - dz2[0], nx, dz2[0]);
  return r;
}

#ifdef HG_DIAG_LEVELS
__device__ unsigned long long g_diag_levels[8];  // wavefronts, all-finest wavefronts, lanes, finest lanes, wavefronts with <= 4 others
#endif
// SHARE: 0 = the latency-bound kernels, 1 = the batched kernel (shared division), 2 = the batched kernel over a
// level-partitioned cloud (staged lookup; its own instantiation, so that batches without a partition keep the code
// they had: the staged form costs them 1.5 % per launch), 3 = the batched kernel of the launch that CLASSIFIES the
// returns for the partition (as 1, and reports whether the lane's finest level was valid)
template <int LEVELS, int SHARE>
__device__ inline D3 pyramid_tsd_direct(const PyramidView& pv, const DirectRaw& raw, double x, double y,
                                        double z, bool* ok, bool pred_fast = false, bool* fine_valid = nullptr) {
  // a coordinate this large has no cell; NaN passes and indexes cell 0 like the general path
  const bool usable = !(fabs(x) >= 1e30 || fabs(y) >= 1e30 || fabs(z) >= 1e30);
  LevelPin lp[LEVELS];
#pragma unroll
  for (int l = 0; l < LEVELS; ++l) lp[l] = pin_level(pv.level[l]);
  int multi = __builtin_amdgcn_readfirstlane(pv.multi_res);
  float min_tsd0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(pv.level[0].min_tsd)));
  asm volatile("" : "+s"(multi), "+s"(min_tsd0));
  // every level directly addressable? (direct_issue has waited for the counters; LEVELS is the number of levels
  // the caller looks up)
  DirectPyramid dp;
  dp.ok = true;
#pragma unroll
  for (int l = 0; l < LEVELS; ++l) {
    const su8 w = raw.w[l];
    bool lok = w[0] == 0u;  // no block in the overflow area
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      // a grid without blocks has min = 0xFFFFFFFF, max = 0: the test passes and every lookup reads
      // the all-zero pool, which is the right answer
      lok = lok & ((w[4 + a] - w[1 + a]) < (1u << lp[l].bits[a]));
      dp.min_b[l][a] = w[1 + a];
    }
    dp.ok = dp.ok & lok;
  }
  *ok = dp.ok;
  if (!dp.ok) return {0.0, 0.0, 0.0, 0.0};  // wave-uniform: the caller takes the general path
  DirectFetch f[LEVELS];
  // Staged lookup (round 5; SHARE = the batched kernel, which is bound by vector issue and the L1, not by latency):
  // 78 % of the returns of the bench scene stop at the finest level. A tile that is PREDICTED to hold only such
  // returns (pred_fast: the level partition below put them first) sets up, loads and interpolates ONE level; the
  // finest level's words are awaited, and only when a lane turns out to need more (a wave-uniform branch on a
  // ballot: the pose has moved since the partition) the wavefront goes on to the coarser levels -- a second round
  // trip, rare. Tiles that are not predicted fast take the plain order: all levels in one round trip. Same voxels,
  // same selection, same arithmetic per lane: bit-identical residuals. (Without the prediction -- every wavefront
  // staged, 37 % of them fast with the scan's structure in the lane order -- the second round trip of the other 63 %
  // eats the gain: 143.5 -> 141 us per launch of 64 matches; with all lanes forced to one level the pass takes 107.)
  constexpr bool kStaged = SHARE == 2 && LEVELS > 1;
  bool staged_fast = false;
  {
    // p / res per level and axis. SHARE (the batched kernel, bound by instruction issue): when every level's
    // resolution is the one before doubled (0.05 / 0.10 / 0.20 m: the same mantissa), the quotient of level l
    // is the quotient of level 0 times 2^-l -- every intermediate of the division sequence scales by that
    // power of two, so the bits are the same -- and the division is done once per axis instead of once per
    // axis and level (64 matches: 210 -> 197 us per launch). Two copies of the set-up code, chosen per
    // wavefront; the latency-bound kernels keep the single copy (the window pass lost 2.6 % with both).
    if constexpr (SHARE) {
      const float r0 = lp[0].res;
      bool doubling = true;
#pragma unroll
      for (int l = 1; l < LEVELS; ++l) doubling = doubling & (lp[l].res == r0 * static_cast<float>(1 << l));
#ifdef HG_ISA_REGIONS
      doubling = true;
#endif
      if (doubling) {
        const float p[3] = {static_cast<float>(x), static_cast<float>(y), static_cast<float>(z)};
        const float rr = refined_rcp(r0);
        float q[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) q[a] = cell_quotient_fast(p[a], r0, rr);
        bool level0_done = false;
        if constexpr (kStaged) {
          if (pred_fast) {  // (wave-uniform)
            direct_setup(lp[0], x, y, z, q, dp.min_b[0], usable, f[0]);
            direct_load(lp[0], f[0]);
            direct_merge(f[0]);
            const bool v0 = (multi != 0) & f[0].in & all_weights_valid(f[0].code);
            staged_fast = __ballot(!v0) == 0ull;
            level0_done = true;
#ifdef HG_DIAG_LEVELS
            if ((threadIdx.x & 63u) == 0) {
              atomicAdd(&g_diag_levels[5], 1ull);
              if (!staged_fast) atomicAdd(&g_diag_levels[6], 1ull);
              atomicAdd(&g_diag_levels[7], static_cast<unsigned long long>(__popcll(__ballot(!v0))));
            }
#endif
          }
        }
        if (!staged_fast) {
#pragma unroll
          for (int l = 0; l < LEVELS; ++l) {
            if (l == 0 && level0_done) continue;
            const float scale = 1.0f / static_cast<float>(1 << l);
            const float ql[3] = {q[0] * scale, q[1] * scale, q[2] * scale};
            direct_setup(lp[l], x, y, z, ql, dp.min_b[l], usable, f[l]);
          }
        }
      } else {
        const float p[3] = {static_cast<float>(x), static_cast<float>(y), static_cast<float>(z)};
#pragma unroll
        for (int l = 0; l < LEVELS; ++l) {
          const float res = lp[l].res, rr = refined_rcp(res);
          const float ql[3] = {cell_quotient_fast(p[0], res, rr), cell_quotient_fast(p[1], res, rr), cell_quotient_fast(p[2], res, rr)};
          direct_setup(lp[l], x, y, z, ql, dp.min_b[l], usable, f[l]);
        }
      }
    } else {
      const float p[3] = {static_cast<float>(x), static_cast<float>(y), static_cast<float>(z)};
#pragma unroll
      for (int l = 0; l < LEVELS; ++l) {
        const float res = lp[l].res, rr = refined_rcp(res);
        const float ql[3] = {cell_quotient_fast(p[0], res, rr), cell_quotient_fast(p[1], res, rr), cell_quotient_fast(p[2], res, rr)};
        direct_setup(lp[l], x, y, z, ql, dp.min_b[l], usable, f[l]);
      }
    }
  }
  BODY_STAMP(1);
  BODY_STAMP(2);
  ISA_MARK("cells+addresses|loads");
  if (!staged_fast) {  // (wave-uniform; always taken by the kernels that are not staged)
    const bool have0 = kStaged && pred_fast;  // the finest level is already in (its lanes turned out to need more)
#pragma unroll
    for (int l = 0; l < LEVELS; ++l)
      if (!(l == 0 && have0)) direct_load(lp[l], f[l]);
#pragma unroll
    for (int l = 0; l < LEVELS; ++l)
      if (!(l == 0 && have0)) direct_merge(f[l]);
  }
  BODY_STAMP(3);
  ISA_MARK("loads|select");
  if (!multi) {
    direct_accept(f[0]);
    const GridView& g = pv.level[0];
    LevelSel s;
    s.x1 = f[0].c[0]; s.y1 = f[0].c[1]; s.z1 = f[0].c[2];
    s.x2 = f[0].c[0] + g.resolution; s.y2 = f[0].c[1] + g.resolution; s.z2 = f[0].c[2] + g.resolution;
    bool all_zero = true;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      s.code[c] = f[0].code[c];
      all_zero = all_zero && ((s.code[c] >> 16) & 0x7FFFu) <= 1u;
    }
    s.tsd_scale = g.tsd_scale; s.tsd_offset = g.tsd_offset; s.min_tsd = g.min_tsd;
    s.weight_scale = g.weight_scale; s.weight_offset = g.weight_offset;
    const D3 r = interp_selected(s, -0.3, x, y, z);
    if (all_zero) return {static_cast<double>(g.min_tsd), 0.0, 0.0, 0.0};  // :86-89
    return r;
  }
  // first level whose 8 weights are all non-zero (interpolated_multi_resolution_tsdf.h:99-106).
  // The per-level codec constants are pinned to scalar registers (readfirstlane): left as plain
  // kernel-argument loads the compiler built a lookup table of them in PRIVATE memory and indexed it
  // per lane (eight scratch stores per lane at kernel start: 3 MB of writes per launch).
  float c3[3] = {f[0].c[0], f[0].c[1], f[0].c[2]};
  uint32_t code[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) code[c] = f[0].code[c];
  float res = lp[0].res, tsd_scale = lp[0].tsd_scale, tsd_offset = lp[0].tsd_offset;
  bool found = staged_fast;  // (every lane's finest level is valid: nothing to select)
  if constexpr (SHARE == 3) { if (fine_valid) *fine_valid = f[0].in & all_weights_valid(f[0].code); }
  if (!staged_fast) {
#pragma unroll
    for (int l = 0; l < LEVELS; ++l) {
      // valid: inside the window (outside it no block exists: the voxels read as unknown, whatever the
      // toroidal slot holds) and all 8 weights non-zero
      const bool valid = f[l].in & all_weights_valid(f[l].code);
      if (l > 0) {
        const bool take = !found && valid;
#pragma unroll
        for (int c = 0; c < 8; ++c) code[c] = take ? f[l].code[c] : code[c];
#pragma unroll
        for (int a = 0; a < 3; ++a) c3[a] = take ? f[l].c[a] : c3[a];
        res = take ? lp[l].res : res;
        tsd_scale = take ? lp[l].tsd_scale : tsd_scale;
        tsd_offset = take ? lp[l].tsd_offset : tsd_offset;
      }
      found = found || valid;
    }
  }
#ifdef HG_DIAG_LEVELS
  {
    // wavefronts whose active lanes all stop at the finest level / lanes that do
    const bool v0 = f[0].in & all_weights_valid(f[0].code);
    const unsigned long long act = __ballot(true), m0 = __ballot(v0);
    if ((threadIdx.x & 63u) == static_cast<unsigned>(__ffsll(static_cast<long long>(act)) - 1)) {
      atomicAdd(&g_diag_levels[0], 1ull);
      atomicAdd(&g_diag_levels[1], m0 == act ? 1ull : 0ull);
      atomicAdd(&g_diag_levels[2], static_cast<unsigned long long>(__popcll(act)));
      atomicAdd(&g_diag_levels[3], static_cast<unsigned long long>(__popcll(m0)));
      atomicAdd(&g_diag_levels[4], __popcll(act & ~m0) <= 4 ? 1ull : 0ull);
    }
  }
#endif
  ISA_MARK("select|interpolation");
  const D3 r = interp_all_valid(res, tsd_scale, tsd_offset, c3, code, x, y, z);
  ISA_MARK("interpolation|row");
  if (!found) return {static_cast<double>(min_tsd0), 0.0, 0.0, 0.0};  // :136
  return r;
}

// interp_selected for the general path: the same seven InterpolateLinear calls in the same order,
// looped over arrays that live in private memory (run-time indices) so that the function needs a few
// dozen registers instead of 160.
__device__ inline D3 interp_selected_compact(const LevelSel& s, double both_invalid, double x, double y, double z) {
  D3 node[8];
  double wt[8];
#pragma unroll 1
  for (int c = 0; c < 8; ++c) {
    wt[c] = static_cast<double>(sel_weight(s, s.code[c]));
    node[c] = {static_cast<double>(sel_tsd(s, s.code[c])), 0.0, 0.0, 0.0};
  }
  const double ix_inv = 1.0 / (s.x2 - s.x1), iy_inv = 1.0 / (s.y2 - s.y1), iz_inv = 1.0 / (s.z2 - s.z1);
  D3 axis[3];
  axis[0] = {(z - s.z1) * iz_inv, 0.0, 0.0, iz_inv};
  axis[1] = {(y - s.y1) * iy_inv, 0.0, iy_inv, 0.0};
  axis[2] = {(x - s.x1) * ix_inv, ix_inv, 0.0, 0.0};
  int count = 8;
#pragma unroll 1
  for (int stage = 0; stage < 3; ++stage) {
    count >>= 1;
#pragma unroll 1
    for (int k = 0; k < count; ++k) {
      D3 q;
      double w;
      interpolate_linear(both_invalid, node[2 * k], node[2 * k + 1], wt[2 * k], wt[2 * k + 1], axis[stage], q, w);
      node[k] = q;
      wt[k] = w;
    }
  }
  return node[0];
}

// General path (a block in the overflow area, or a map wider than the window): hash lookups, one
// corner after the other. It runs only for grids that have outgrown their window, so it is written
// for a small register footprint, not for speed: the kernel's register allocation is the maximum over
// everything it can call, and the direct path must not pay for this one. Not inlined; it reads the
// pyramid from its copy in device memory (a reference to the caller's kernel argument would force that
// into private memory and turn every field access of the hot path into a scratch load; by value it
// would travel in a hundred registers).
__device__ __attribute__((noinline)) D3 pyramid_tsd_general(const PyramidView* pv_mem, double x, double y, double z) {
  const PyramidView& pv = *pv_mem;
  const int levels = pv.multi_res ? pv.levels : 1;
  LevelSel s;
  bool found = false;
#pragma unroll 1
  for (int l = 0; l < levels; ++l) {
    const GridView& g = pv.level[l];
    const float res = g.resolution;
    // CenterOfLowerVoxel (interpolated_tsdf.h:176-192), as direct_setup
    int i0[3] = {cell_index_1d(static_cast<float>(x), res), cell_index_1d(static_cast<float>(y), res),
                 cell_index_1d(static_cast<float>(z), res)};
    float c[3] = {static_cast<float>(i0[0]) * res, static_cast<float>(i0[1]) * res, static_cast<float>(i0[2]) * res};
    if (static_cast<double>(c[0]) > x) { c[0] -= res; i0[0] -= 1; }
    if (static_cast<double>(c[1]) > y) { c[1] -= res; i0[1] -= 1; }
    if (static_cast<double>(c[2]) > z) { c[2] -= res; i0[2] -= 1; }
    uint32_t code[8];
    bool valid = true;
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
      code[k] = load_voxel(g, i0[0] + (k >> 2), i0[1] + ((k >> 1) & 1), i0[2] + (k & 1));
      valid = valid && ((code[k] >> 16) & 0x7FFFu) > 1u;
    }
    if (!found && (valid || l == 0 || !pv.multi_res)) {  // level 0 stands in until a level is found
      s.x1 = c[0]; s.y1 = c[1]; s.z1 = c[2];
      s.x2 = c[0] + res; s.y2 = c[1] + res; s.z2 = c[2] + res;
#pragma unroll
      for (int k = 0; k < 8; ++k) s.code[k] = code[k];
      s.tsd_scale = g.tsd_scale; s.tsd_offset = g.tsd_offset; s.min_tsd = g.min_tsd;
      s.weight_scale = g.weight_scale; s.weight_offset = g.weight_offset;
    }
    found = found || valid;
    if (!pv.multi_res) {
      bool all_zero = true;
#pragma unroll
      for (int k = 0; k < 8; ++k) all_zero = all_zero && ((code[k] >> 16) & 0x7FFFu) <= 1u;
      const D3 r = interp_selected_compact(s, -0.3, x, y, z);
      if (all_zero) return {static_cast<double>(g.min_tsd), 0.0, 0.0, 0.0};  // :86-89
      return r;
    }
    if (__ballot(!found) == 0ull) break;
  }
  const D3 r = interp_selected_compact(s, static_cast<double>(s.min_tsd), x, y, z);
  if (!found) return {static_cast<double>(pv.level[0].min_tsd), 0.0, 0.0, 0.0};  // :136
  return r;
}

template <int SHARE = 0>
__device__ inline D3 pyramid_tsd(const PyramidView& pv, const DirectRaw& raw, double x, double y, double z,
                                 bool pred_fast = false, bool* fine_valid = nullptr) {
  const int levels = pv.multi_res ? pv.levels : 1;
  bool ok;
  D3 r;
#ifdef HG_ISA_REGIONS
  (void)levels;
  r = pyramid_tsd_direct<3, SHARE>(pv, raw, x, y, z, &ok);
  if (ok) return r;
  return pyramid_tsd_general(pv.self_mem, x, y, z);
#endif
#ifdef HG_FORCE_L1  // (timing experiment only: every lookup stops at the finest level -- wrong results)
  r = pyramid_tsd_direct<1, SHARE>(pv, raw, x, y, z, &ok);
  if (ok) return r;
#endif
  switch (levels) {  // wave-uniform
    case 1: r = pyramid_tsd_direct<1, SHARE>(pv, raw, x, y, z, &ok, pred_fast, fine_valid); break;
    case 2: r = pyramid_tsd_direct<2, SHARE>(pv, raw, x, y, z, &ok, pred_fast, fine_valid); break;
    case 3: r = pyramid_tsd_direct<3, SHARE>(pv, raw, x, y, z, &ok, pred_fast, fine_valid); break;
    default: r = pyramid_tsd_direct<4, SHARE>(pv, raw, x, y, z, &ok, pred_fast, fine_valid); break;
  }
  if (ok) return r;
  if constexpr (SHARE == 3) *fine_valid = false;  // (a pool that is not directly addressable has no staged form)
  return pyramid_tsd_general(pv.self_mem, x, y, z);
}

__device__ inline void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

// One return at transform (t, q): row8 = [d r / d(t, q) (7) | r]. The world point follows Eigen's
// QuaternionBase::_transformVector, then + translation (rigid_transform.h:193-197).
template <int SHARE = 0>
__device__ __forceinline__ void return_row(const PyramidView& pv, const DirectRaw& dp, const double* t,
                                           const double* q, const double* v, double scaling, double* row8,
                                           bool pred_fast = false, bool* fine_valid = nullptr) {
  const double qw = q[0];
  const double u[3] = {q[1], q[2], q[3]};
  double uv[3], c2[3];
  cross3(u, v, uv);
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  cross3(u, uv, c2);
  const double wx = (v[0] + qw * uv[0] + c2[0]) + t[0];
  const double wy = (v[1] + qw * uv[1] + c2[1]) + t[1];
  const double wz = (v[2] + qw * uv[2] + c2[2]) + t[2];
  ISA_MARK("transform|cells+addresses");
  const D3 tsd = pyramid_tsd<SHARE>(pv, dp, wx, wy, wz, pred_fast, fine_valid);
  const double r = scaling * tsd.a;
  const double g[3] = {scaling * tsd.d0, scaling * tsd.d1, scaling * tsd.d2};
  // d world / d q = [uv | qw duv_k + e_k x uv + u x duv_k], duv_k = 2 (e_k x v). Written out per k with
  // the structural zeros of e_k x . dropped and the sums as fused multiply-adds: nothing downstream of
  // the world point takes a discrete decision, and the row only has to agree with the reference's Jet
  // evaluation to rounding (tests: 1e-9 relative on J^T J); the world point itself keeps Eigen's
  // operation order, because cell indices and validity branches are derived from it.
  row8[0] = g[0]; row8[1] = g[1]; row8[2] = g[2];
  row8[3] = fma(g[2], uv[2], fma(g[1], uv[1], g[0] * uv[0]));
  const double v2[3] = {v[0] + v[0], v[1] + v[1], v[2] + v[2]};
  {
    // k = 0: duv = (0, -2 v2, 2 v1); e0 x uv = (0, -uv2, uv1); u x duv = (u1 duv2 - u2 duv1, -u0 duv2, u0 duv1)
    const double d1 = -v2[2], d2 = v2[1];
    const double c0 = fma(u[1], d2, -(u[2] * d1));
    const double c1 = fma(qw, d1, -uv[2]) - u[0] * d2;
    const double c2k = fma(qw, d2, uv[1]) + u[0] * d1;
    row8[4] = fma(g[2], c2k, fma(g[1], c1, g[0] * c0));
  }
  {
    // k = 1: duv = (2 v2, 0, -2 v0); e1 x uv = (uv2, 0, -uv0); u x duv = (u1 duv2, u2 duv0 - u0 duv2, -u1 duv0)
    const double d0 = v2[2], d2 = -v2[0];
    const double c0 = fma(qw, d0, uv[2]) + u[1] * d2;
    const double c1 = fma(u[2], d0, -(u[0] * d2));
    const double c2k = fma(qw, d2, -uv[0]) - u[1] * d0;
    row8[5] = fma(g[2], c2k, fma(g[1], c1, g[0] * c0));
  }
  {
    // k = 2: duv = (-2 v1, 2 v0, 0); e2 x uv = (-uv1, uv0, 0); u x duv = (-u2 duv1, u2 duv0, u0 duv1 - u1 duv0)
    const double d0 = -v2[1], d1 = v2[0];
    const double c0 = fma(qw, d0, -uv[1]) - u[2] * d1;
    const double c1 = fma(qw, d1, uv[0]) + u[2] * d0;
    const double c2k = fma(u[0], d1, -(u[1] * d0));
    row8[6] = fma(g[2], c2k, fma(g[1], c1, g[0] * c0));
  }
  row8[7] = r;
  ISA_MARK("row|tile");
}

// A wavefront executes its LDS instructions in order, so lanes that exchange data through LDS with
// lanes of the SAME wavefront only need the compiler not to move accesses across the point. (The LM
// state machine runs in one wavefront; the X tiles of the X^T X reduction are per wavefront.)
__device__ inline void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Partial sums travel from every workgroup to the one that runs the LM step. Stored write-through
// (sc1) and read around the vector L1 (sc1), they need no release / acquire fence pair on the way
// (MI355X_MICROARCH.md, "Valid forms": every store of the handed-off bytes sc1 and drained by the
// storing wave's s_waitcnt vmcnt(0) before the workgroup's arrival is counted; every load of them sc1,
// by the wavefront whose counter add came last after that add has returned, by the others after a
// workgroup barrier that wavefront joins).
__device__ inline void store_partial(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline double load_partial(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Round 5, the single-pose chain: partial sums that are their own flag (cdna_hip_programming.md Guideline 16, R2:
// "the data IS the flag"). A double travels as two 8-byte granules {tag = this launch's epoch, 32 bits of the value},
// each written by ONE aligned 8-byte write-through store; the reader re-reads a granule until its tag is the epoch.
// No acknowledgement wait, no ticket, no reload behind the ticket: the hand-over was store drain (0.7 us) + returning
// atomic (0.8 us) + loads (1.2 us) per launch, now it is the visibility of the last workgroup's stores plus the
// re-read that sees them. Epochs count launches per problem (never 0, never reused: the buffer starts zeroed).
__device__ inline void store_partial_tagged(unsigned long long* g, double v, unsigned epoch) {
  const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(v));
  const unsigned long long tag = static_cast<unsigned long long>(epoch) << 32;
  __hip_atomic_store(g, tag | (bits & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(g + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// both granules of a double by ONE 16-byte load around the L1 (each 8-byte half is a granule of its own, validated by
// its own tag; two 8-byte loads per double made the first sweep twice as long -- 8-byte accesses run at 0.54-0.70x the
// 16-byte rate). Fourteen loads and their wait in ONE asm statement: the compiler knows nothing of an asm load's
// latency, so no result may leave the statement before the wait. (A buffer load with the sc1 bit through
// __builtin_amdgcn_raw_buffer_load_b128 never saw the tags: measured, dropped.)
typedef unsigned long long hg_u64x2 __attribute__((ext_vector_type(2)));
constexpr int kSweep = 14;
__device__ __forceinline__ void load_granule_pairs(hg_u64x2 (&v)[kSweep], const unsigned long long* const (&g)[kSweep]) {
  asm volatile(
      "global_load_dwordx4 %0, %14, off sc1\n global_load_dwordx4 %1, %15, off sc1\n"
      "global_load_dwordx4 %2, %16, off sc1\n global_load_dwordx4 %3, %17, off sc1\n"
      "global_load_dwordx4 %4, %18, off sc1\n global_load_dwordx4 %5, %19, off sc1\n"
      "global_load_dwordx4 %6, %20, off sc1\n global_load_dwordx4 %7, %21, off sc1\n"
      "global_load_dwordx4 %8, %22, off sc1\n global_load_dwordx4 %9, %23, off sc1\n"
      "global_load_dwordx4 %10, %24, off sc1\n global_load_dwordx4 %11, %25, off sc1\n"
      "global_load_dwordx4 %12, %26, off sc1\n global_load_dwordx4 %13, %27, off sc1\n"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]),
        "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13])
      : "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4]), "v"(g[5]), "v"(g[6]), "v"(g[7]), "v"(g[8]), "v"(g[9]), "v"(g[10]),
        "v"(g[11]), "v"(g[12]), "v"(g[13])
      : "memory");
}
[[maybe_unused]] constexpr unsigned kBcastReplicas = 32;  // copies of the persistent solve's hand-back (k_tsdf_residuals_single_persist)
__device__ inline hg_u64x2 load_granule_pair16(const unsigned long long* g) {  // one 16-byte load around the L1, waited for
  hg_u64x2 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=&v"(v) : "v"(g) : "memory");
  return v;
}
__device__ inline double uniform_f64(double x) {  // a value every lane holds, moved to scalar registers
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
  return __hiloint2double(hi, lo);
}
__device__ inline hg_u64x2 load_granule_pair(const unsigned long long* g) {  // (the re-reads: two 8-byte loads)
  return hg_u64x2{__hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                  __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
}

// Which return a lane takes. The vector L1 prices a gather by the distinct 128-byte lines the four lanes of
// every aligned quad touch (scripts/tcp_bench.hip: one cycle per line and quad, no merging beyond adjacent
// lanes), and the voxel lookups of the batched and window passes are bound by exactly that (round 4: L1 busy 91 %,
// 0.71 tag lookups per cycle and CU, VALU 40 %). A structured scan is stored azimuth-major (`width` = returns per
// column, sensor/range_data.h): adjacent returns are VERTICAL neighbours, whose voxels differ in z -- the slowest
// dimension of a block -- so the lanes of a quad hit four different lines per load (3.6 / 2.8 / 2.2 lines per quad
// at 0.05 / 0.10 / 0.20 m on the bench scene). When the caller has told the width (hg_problem_set_block_width,
// the width argument of hg_register_scan*), lane order is remapped: four adjacent lanes take the same ring of four
// adjacent columns -- horizontal neighbours 0.18 degrees apart, the same or neighbouring x / y cells: 1.55 / 1.30 /
// 1.33 lines per quad. A permutation of the summation order only: residuals keep their positions.
struct ScanOrder {
  unsigned width, group, full;  // group = 4 * width returns (four columns); returns >= full keep their order
  float inv_group;
};
__device__ inline ScanOrder make_scan_order(unsigned n, unsigned width) {
  ScanOrder so;
  so.width = width;
  so.group = (width != 0u && n < (1u << 24)) ? 4u * width : 0u;
  so.inv_group = so.group ? 1.0f / static_cast<float>(so.group) : 0.f;
  unsigned q = static_cast<unsigned>(static_cast<float>(n) * so.inv_group);
  if (so.group && q * so.group > n) --q;
  so.full = q * so.group;
  return so;
}
__device__ inline unsigned scan_index(const ScanOrder& so, unsigned i) {
  if (so.group == 0u || i >= so.full) return i;
  // i / group by a float reciprocal (i < 2^24: exact conversion) and one correction step either way
  unsigned q = static_cast<unsigned>(static_cast<float>(i) * so.inv_group);
  int r = static_cast<int>(i - q * so.group);
  if (r < 0) { r += static_cast<int>(so.group); --q; }
  else if (r >= static_cast<int>(so.group)) { r -= static_cast<int>(so.group); ++q; }
  return q * so.group + (static_cast<unsigned>(r) & 3u) * so.width + (static_cast<unsigned>(r) >> 2);
}
// The return's coordinates as ONE 12-byte load (three 4-byte loads cost three passes through the L1).
__device__ inline void load_point(const float* __restrict__ xyz, unsigned i, double* v) {
  typedef float f3 __attribute__((ext_vector_type(3), aligned(4)));
  const f3 p = *reinterpret_cast<const __attribute__((address_space(1))) f3*>(as_global(xyz) + 12ull * i);
  v[0] = static_cast<double>(p.x);
  v[1] = static_cast<double>(p.y);
  v[2] = static_cast<double>(p.z);
}

// The transform (t, q) of a block from device memory by SCALAR loads: the address is the same in every lane, but
// the compiler cannot prove that no store of the kernel aliases it and emits four 16-byte vector loads of one
// address -- four passes through the vector L1 per wavefront (round 4: that unit and the VALU share the bound of
// the batched pass). Written by an earlier launch (k_lm* / the LM tail); the scalar cache starts every kernel
// invalidated. Issue and wait in one statement (see direct_issue).
__device__ inline void load_transform_uniform(const double* tq, double* out7) {
  typedef unsigned su4 __attribute__((ext_vector_type(4)));
  typedef unsigned su2 __attribute__((ext_vector_type(2)));
  const unsigned long long a = reinterpret_cast<unsigned long long>(tq);
  const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(a)));
  const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(a >> 32)));
  const unsigned long long ua = (static_cast<unsigned long long>(hi) << 32) | lo;
  su8 w0; su4 w1; su2 w2;
  asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x20\n\ts_load_dwordx2 %2, %3, 0x30\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&s"(w0), "=&s"(w1), "=&s"(w2) : "s"(ua));
  out7[0] = __hiloint2double(static_cast<int>(w0[1]), static_cast<int>(w0[0]));
  out7[1] = __hiloint2double(static_cast<int>(w0[3]), static_cast<int>(w0[2]));
  out7[2] = __hiloint2double(static_cast<int>(w0[5]), static_cast<int>(w0[4]));
  out7[3] = __hiloint2double(static_cast<int>(w0[7]), static_cast<int>(w0[6]));
  out7[4] = __hiloint2double(static_cast<int>(w1[1]), static_cast<int>(w1[0]));
  out7[5] = __hiloint2double(static_cast<int>(w1[3]), static_cast<int>(w1[2]));
  out7[6] = __hiloint2double(static_cast<int>(w2[1]), static_cast<int>(w2[0]));
}

// residuals of one block at its current transform + 36 partial sums per workgroup
template <int THREADS = kEvalThreads, int SHARE = (THREADS == 256 ? 1 : 0)>
__device__ __forceinline__ void tsdf_residuals_body(
    const PyramidView& pv, const float* __restrict__ xyz, unsigned n, double scaling,
    const BlockXform* __restrict__ xf, double* __restrict__ partials,
    double* __restrict__ residuals, double (*xs)[kWave][8], double (*cs)[64], unsigned wg,
    const double* pose_tq = nullptr /* the transform when it does not come from xf (first launch) */,
    unsigned width = 0, unsigned tiles = 1 /* tiles of THREADS returns per workgroup (the batched kernel) */,
    unsigned fast_n = 0 /* the first fast_n returns are expected to stop at the finest level (level partition) */,
    unsigned epoch = 0 /* != 0: the partial sums go out as tagged granules (store_partial_tagged), `partials` = the granule buffer */,
    const double* pre_v = nullptr, const DirectRaw* pre_dp = nullptr /* the lane's return and the window counters, loaded
        by the caller (the persistent solve evaluates the same returns against the same map many times) */,
    unsigned char* flags_out = nullptr, unsigned* wave_counts = nullptr /* SHARE == 3: per return (lane order) whether it
        stopped at the finest level, and how many of every wavefront's 64 did (the level partition's classification) */) {
  const ScanOrder order = make_scan_order(n, width);
  const unsigned first_i0 = wg * tiles * THREADS + threadIdx.x;
  BODY_STAMP(0);
  ISA_MARK("entry|transform");
  // (the point load is issued in front of the scalar loads of direct_issue, whose wait it then overlaps)
  unsigned i = scan_index(order, first_i0 < n ? first_i0 : 0u);
  double v[3];
  if (pre_v) { v[0] = pre_v[0]; v[1] = pre_v[1]; v[2] = pre_v[2]; }
  else load_point(xyz, i, v);
  const DirectRaw dp = pre_dp ? *pre_dp : direct_issue(pv);
  double tq[7];
  if (pose_tq) {
#pragma unroll
    for (int k = 0; k < 7; ++k) tq[k] = pose_tq[k];
  } else {
    static_assert(offsetof(BlockXform, q) == 3 * sizeof(double), "t and q are contiguous");
    load_transform_uniform(xf->t, tq);
  }
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 cacc = {0.0, 0.0, 0.0, 0.0};
  const int mj = lane & 15, mk = lane >> 4;
  // The batched kernel (THREADS == 256) gives a workgroup `tiles` tiles of 256 returns: the accumulators of the
  // matrix cores run through them, so the prologue, the reduction over the wavefronts and the 36 partial sums per
  // workgroup (which the step kernel has to read back) are paid once per `tiles` tiles; the next tile's return is
  // fetched while the current one is evaluated.
  for (unsigned tile = 0; tile < tiles; ++tile) {
    const unsigned i0 = first_i0 + tile * THREADS;
    if (tile > 0 && i0 - threadIdx.x >= n) break;  // uniform
    double row8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const unsigned i_cur = i;
    const double vc[3] = {v[0], v[1], v[2]};
    if (tile + 1 < tiles) {
      const unsigned i1 = i0 + THREADS;
      i = scan_index(order, i1 < n ? i1 : 0u);
      load_point(xyz, i, v);
    }
    bool fine = false;
    if (i0 < n) {
      // (THREADS == 256: the batched kernel)
      return_row<SHARE>(pv, dp, tq, tq + 3, vc, scaling, row8, i0 - threadIdx.x + THREADS <= fast_n, &fine);
      if (residuals) residuals[i_cur] = row8[7];
    }
    if constexpr (SHARE == 3) {
      if (i0 < n) flags_out[i0] = fine ? 1 : 0;
      const unsigned long long m = __ballot(fine);
      // (every wavefront of a 256-return group that holds a return reports, the empty ones of the last group 0)
      if (lane == 0 && i0 - threadIdx.x < n) wave_counts[i0 >> 6] = static_cast<unsigned>(__popcll(m));
    }
    if (tile > 0) wave_sync();  // the operand reads of the tile before are done
    {
      typedef double d2 __attribute__((ext_vector_type(2)));
      d2* dst = reinterpret_cast<d2*>(&xs[wave][lane][0]);
#pragma unroll
      for (int c = 0; c < 4; ++c) dst[c] = d2{row8[2 * c], row8[2 * c + 1]};
    }
    wave_sync();  // xs[wave] is written and read by this wavefront only
    {
      const int half = (mj >> 3) * 32, col = mj & 7;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const double a = xs[wave][half + 4 * s + mk][col];
        cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, cacc, 0, 0, 0);
      }
    }
  }
  BODY_STAMP(4);
  // J^T J accumulation on the matrix cores: per wavefront X = [row | r] is 64 x 8 (padded to 16
  // columns); 16 x v_mfma_f64_16x16x4_f64 accumulate X^T X, whose upper-left 8 x 8 block holds
  // J^T J (7x7), J^T r (column 7) and r^T r. Operand layout: lane l feeds A[i = l%16][k = l/16] and
  // B[k = l/16][j = l%16] — here the same element X[4s + l/16][l%16]; D[l/16 + 4v][l%16] comes back
  // in accumulator register v (verified against the CPU oracle by the parity tests).
  // (Round 4: the 16-wide operand carries TWO row groups -- columns 0..7 the returns 0..31 of the wavefront,
  // columns 8..15 the returns 32..63 -- so 8 instead of 16 MFMAs (64 cycles each, dependent) form X^T X: its
  // upper-left 8 x 8 block is the sum over the first 32 returns, the lower-right one over the other 32, the
  // off-diagonal blocks (cross terms) are dropped.)
  wave_sync();  // the operand reads are done: the wavefront's X tile takes the second block
  {
    // D[l/16 + 4v][l%16] sits in accumulator register v: lanes with mj < 8 hold rows mk, mk + 4 of the first
    // block (registers 0, 1), lanes with mj >= 8 rows mk, mk + 4 of the second (registers 2, 3)
    double* second = &xs[wave][0][0];
    double* dst = (mj < 8) ? &cs[wave][0] : second;
    const int col = mj & 7;
    dst[mk * 8 + col] = (mj < 8) ? cacc[0] : cacc[2];
    dst[(mk + 4) * 8 + col] = (mj < 8) ? cacc[1] : cacc[3];
  }
  __syncthreads();
  if (threadIdx.x < kAcc) {
    // partial layout: 28 upper-triangle entries of the 7x7 block (row-major), 7 x J^T r, r^T r
    int a, b;
    if (threadIdx.x < 28) {
      int t = threadIdx.x;
      a = 0;
      while (t >= 7 - a) { t -= 7 - a; ++a; }
      b = a + t;
    } else if (threadIdx.x < 35) {
      a = threadIdx.x - 28;
      b = 7;
    } else {
      a = 7;
      b = 7;
    }
    double s = 0.0;
#pragma unroll
    for (int wv = 0; wv < THREADS / kWave; ++wv) s += cs[wv][a * 8 + b] + xs[wv][0][a * 8 + b];
    if (epoch != 0u)  // (uniform)
      store_partial_tagged(reinterpret_cast<unsigned long long*>(partials) + (static_cast<size_t>(wg) * kAcc + threadIdx.x) * 2, s, epoch);
    else
      store_partial(&partials[static_cast<size_t>(wg) * kAcc + threadIdx.x], s);
  }
  BODY_STAMP(5);
  ISA_MARK("tile|exit");
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
#pragma clang fp contract(fast)  // (LM side only: no discrete decision downstream; see lm_step_single)
  const double nd = sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
  if (nd > 0.0) {
    // (Round 6, measured and dropped: sin(nd) / nd and cos(nd) as Taylor series below 0.5 rad -- 18 multiply-adds
    // instead of the library's argument reduction and an fp64 division, twice on the serial path of every LM step. A
    // launch per evaluation gained 1 % (3809 -> 3848 scans/s), the persistent solve LOST 2.5 % (3951 -> 3854): that
    // kernel is out of registers and the polynomials' constants became spills on the serial path
    // (scripts/r06_headline_variants.sh). Series in one form only would end the bit-for-bit equality of the two.)
    double sn, q0;
    sincos(nd, &sn, &q0);  // (one argument reduction for both)
    const double sbd = sn / nd;
    const double q1 = sbd * delta[0], q2 = sbd * delta[1], q3 = sbd * delta[2];
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
// Eigen 3.3 Quaternion::slerp(t, b) of a (Geometry/Quaternion.h) in Jet arithmetic; variables: the
// coefficients (w,x,y,z) of a -> 0..3, of b -> 4..7.
__device__ inline void slerp_jets(const double* qa, const double* qb, double f, DJ<8>* q) {
  typedef DJ<8> J;
  J aw = dj_var<8>(qa[0], 0), ax = dj_var<8>(qa[1], 1), ay = dj_var<8>(qa[2], 2), az = dj_var<8>(qa[3], 3);
  J bw = dj_var<8>(qb[0], 4), bx = dj_var<8>(qb[1], 5), by = dj_var<8>(qb[2], 6), bz = dj_var<8>(qb[3], 7);
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
  q[0] = s0 * aw + s1 * bw;
  q[1] = s0 * ax + s1 * bx;
  q[2] = s0 * ay + s1 * by;
  q[3] = s0 * az + s1 * bz;
}

// Single pose: T = pose_a. Two poses: InterpolateTransform (transform/timestamped_transform.h:41-51)
// = lerp of translations + Eigen 3.3 Quaternion::slerp.
// xf->M must be zero on entry (only the structural non-zeros are written).
__device__ __forceinline__ void prepare_block(const BlockInfo& b, const double (*poses)[kState], BlockXform* xf) {
  if (b.acc == kAccU) return;  // per-return factors: the residual kernel interpolates itself
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
  DJ<8> q[4];
  slerp_jets(pa + 3, pb + 3, f, q);
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

// ------------------------------------------------------------------------------------------
// Window pass (every problem but the single-pose registration step): lean residual bodies without a
// solver tail. A workgroup of 256 threads walks `tiles` consecutive 256-return tiles of ONE block and
// keeps X^T X in the MFMA accumulators across them, so a window of nine 100k-point scans is ~700
// resident workgroups (one round on the chip, three per CU) that leave 36 sums each, instead of 1800
// workgroups of 512 threads at one per CU. The partials cross a kernel boundary to k_lm.
// ------------------------------------------------------------------------------------------
struct EvalBlock {
  PyramidView pv;
  const float* xyz;
  const double* factor;  // per-return interpolation ratios, or nullptr
  double scaling;
  unsigned n, wg_begin, num_wg, partial_offset, row_offset;
  int pose_a, pose_b, index;
  unsigned width;  // returns per column of a structured scan (0 = unknown: lanes take returns in order)
  unsigned pad;
};

// Per-scan blocks (one transform per block; the map to the local parameters is applied after the
// reduction, in k_lm): X = [d r / d(t, q) | r], 64 x 8 per wavefront and tile.
__device__ __forceinline__ void window_body_plain(const EvalBlock& eb, const BlockXform* __restrict__ xf,
                                                  double* __restrict__ partials, double* __restrict__ residuals,
                                                  unsigned wg, unsigned tiles, unsigned char* smem) {
  constexpr int kWaves = kBatchThreads / kWave;
  double (*xs)[kWave][8] = reinterpret_cast<double (*)[kWave][8]>(smem);
  double (*cs)[64] = reinterpret_cast<double (*)[64]>(smem + kWaves * kWave * 8 * sizeof(double));
  const PyramidView& pv = eb.pv;
  const DirectRaw dp = direct_issue(pv);
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const int mj = lane & 15, mk = lane >> 4;
  typedef double d4 __attribute__((ext_vector_type(4)));
  typedef double d2 __attribute__((ext_vector_type(2)));
  d4 cacc = {0.0, 0.0, 0.0, 0.0};
  const unsigned n = eb.n;
  const double scaling = eb.scaling;
  const ScanOrder order = make_scan_order(n, eb.width);
  const unsigned first = xcd_chunk(wg, eb.num_wg) * tiles;
  double tq[7];
  load_transform_uniform(xf->t, tq);
  // The return of tile t + 1 is fetched while tile t is evaluated (round 4): a wavefront works through its tiles
  // one after the other, and each began with the round trip of its point load in front of the voxel round trip.
  typedef float f3 __attribute__((ext_vector_type(3), aligned(4)));
  auto fetch = [&](unsigned tile, unsigned* index) {
    const unsigned i0 = (first + tile) * kBatchThreads + threadIdx.x;
    const unsigned i = scan_index(order, (tile < tiles && i0 < n) ? i0 : 0u);
    *index = i;
    return *reinterpret_cast<const __attribute__((address_space(1))) f3*>(as_global(eb.xyz) + 12ull * i);
  };
  unsigned i_next;
  f3 p_next = fetch(0, &i_next);
  for (unsigned tile = 0; tile < tiles; ++tile) {
    const unsigned base = (first + tile) * kBatchThreads;
    if (base >= n) break;  // uniform
    const unsigned i0 = base + threadIdx.x;
    const unsigned i = i_next;
    const f3 p = p_next;
    p_next = fetch(tile + 1, &i_next);
    double row8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (i0 < n) {
      const double v[3] = {static_cast<double>(p.x), static_cast<double>(p.y), static_cast<double>(p.z)};
      return_row(pv, dp, tq, tq + 3, v, scaling, row8);
      if (residuals) residuals[i] = row8[7];
    }
    d2* dst = reinterpret_cast<d2*>(&xs[wave][lane][0]);
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[c] = d2{row8[2 * c], row8[2 * c + 1]};
    wave_sync();  // xs[wave] is written and read by this wavefront only
    {
      // two row groups in the 16-wide operand: 8 MFMAs per tile (tsdf_residuals_body)
      const int half = (mj >> 3) * 32, col = mj & 7;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const double a = xs[wave][half + 4 * s + mk][col];
        cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, cacc, 0, 0, 0);
      }
    }
    wave_sync();  // the operand reads are done before the next tile overwrites the tile
  }
  {
    double* second = &xs[wave][0][0];
    double* dst = (mj < 8) ? &cs[wave][0] : second;
    const int col = mj & 7;
    dst[mk * 8 + col] = (mj < 8) ? cacc[0] : cacc[2];
    dst[(mk + 4) * 8 + col] = (mj < 8) ? cacc[1] : cacc[3];
  }
  __syncthreads();
  if (threadIdx.x < kAcc) {
    // partial layout: 28 upper-triangle entries of the 7x7 block (row-major), 7 x J^T r, r^T r
    int a, b;
    if (threadIdx.x < 28) {
      int t = threadIdx.x;
      a = 0;
      while (t >= 7 - a) { t -= 7 - a; ++a; }
      b = a + t;
    } else if (threadIdx.x < 35) {
      a = threadIdx.x - 28;
      b = 7;
    } else {
      a = 7;
      b = 7;
    }
    double sum = 0.0;
#pragma unroll
    for (int wv = 0; wv < kWaves; ++wv) sum += cs[wv][a * 8 + b] + xs[wv][0][a * 8 + b];
    store_partial(&partials[static_cast<size_t>(wg) * kAcc + threadIdx.x], sum);
  }
}

// Block with one interpolation factor per return (per-point unwarping: the subdivision blocks of
// AddPerPointMatchingResiduals, optimizing_local_trajectory_builder.cc:513-612, and
// InterpolatedTSDFPerPointSpaceCostFunction3D): every lane interpolates its own transform between
// control points a and b and forms its Jacobian row directly over the 12 local columns
// [pose_a 6 | pose_b 6]; X = [row12 | r] (64 x 13, padded to 16) goes through the same MFMA X^T X
// reduction. 91 partial sums per workgroup: upper triangle of the 12 x 12 block, J^T r, r^T r.
__device__ __forceinline__ void window_body_unwarp(const EvalBlock& eb, const double* pa, const double* pb,
                                                   double* __restrict__ partials, double* __restrict__ residuals,
                                                   unsigned wg, unsigned tiles, unsigned char* smem) {
  constexpr int kWaves = kBatchThreads / kWave;
  double (*xs)[kWave][16] = reinterpret_cast<double (*)[kWave][16]>(smem);
  double (*cs)[256] = reinterpret_cast<double (*)[256]>(smem + kWaves * kWave * 16 * sizeof(double));
  const PyramidView& pv = eb.pv;
  const DirectRaw dp = direct_issue(pv);
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const int mj = lane & 15, mk = lane >> 4;
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 cacc = {0.0, 0.0, 0.0, 0.0};
  const unsigned n = eb.n;
  const double scaling = eb.scaling;
  const ScanOrder order = make_scan_order(eb.n, eb.width);
  const unsigned first = xcd_chunk(wg, eb.num_wg) * tiles;
  for (unsigned tile = 0; tile < tiles; ++tile) {
    const unsigned base = (first + tile) * kBatchThreads;
    if (base >= n) break;  // uniform
    const unsigned i0 = base + threadIdx.x;
    double row[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) row[k] = 0.0;
    if (i0 < n) {
      const unsigned i = scan_index(order, i0);
      const double f = eb.factor[i];
      double pja[12], pjb[12];
      quaternion_plus_jacobian(pa + 3, pja);
      quaternion_plus_jacobian(pb + 3, pjb);
      double t[3], q[4];
      for (int k = 0; k < 3; ++k) t[k] = pa[k] + (pb[k] - pa[k]) * f;
      DJ<8> qj[4];
      slerp_jets(pa + 3, pb + 3, f, qj);
      for (int r = 0; r < 4; ++r) q[r] = qj[r].a;
      double v[3];
      load_point(eb.xyz, i, v);
      double row8[8];
      return_row(pv, dp, t, q, v, scaling, row8);
      if (residuals) residuals[i] = row8[7];
      const double ma = 1.0 + (0.0 - 1.0) * f, mb = (1.0 - 0.0) * f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        row[c] = row8[c] * ma;
        row[6 + c] = row8[c] * mb;
        double ja = 0.0, jb = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double sa = 0.0, sb = 0.0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            sa += qj[r].v[j] * pja[j * 3 + c];
            sb += qj[r].v[4 + j] * pjb[j * 3 + c];
          }
          ja += row8[3 + r] * sa;
          jb += row8[3 + r] * sb;
        }
        row[3 + c] = ja;
        row[9 + c] = jb;
      }
      row[12] = row8[7];
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) xs[wave][lane][c] = c < 13 ? row[c] : 0.0;
    wave_sync();
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const double a = xs[wave][4 * s + mk][mj];
      cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, cacc, 0, 0, 0);
    }
    wave_sync();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) cs[wave][(mk + 4 * r) * 16 + mj] = cacc[r];  // D[l/16 + 4v][l%16]
  __syncthreads();
  if (threadIdx.x < kAccU) {
    int a, b;
    if (threadIdx.x < 78) {
      int t = threadIdx.x;
      a = 0;
      while (t >= 12 - a) { t -= 12 - a; ++a; }
      b = a + t;
    } else if (threadIdx.x < 90) {
      a = threadIdx.x - 78;
      b = 12;
    } else {
      a = 12;
      b = 12;
    }
    double sum = 0.0;
#pragma unroll
    for (int wv = 0; wv < kWaves; ++wv) sum += cs[wv][a * 16 + b];
    store_partial(&partials[static_cast<size_t>(wg) * kAccU + threadIdx.x], sum);
  }
}

// Tail of a block's LAST workgroup (a ticket per block counts them): sums the block's workgroup
// partials and maps them to the block's local normal equations over [pose_a 6 | pose_b 6] --
// J^T J = M^T A7 M, J^T r = M^T b7 with M = d(t, q) / d(local parameters) of the block's transform
// (prepare_block) for per-scan blocks; blocks with a ratio per return already are in that form. The
// blocks of a window finish at about the same time, so these tails run side by side, and k_lm reads 91
// numbers per block instead of a thousand partials. Hand-over of the partials as in single_eval: sc1
// stores drained by every storing wavefront, one counted arrival per workgroup behind the barrier, sc1
// loads by the workgroup whose arrival was the last.
template <bool UNWARP>
__device__ __forceinline__ void window_block_tail(const EvalBlock& eb, const double* __restrict__ partials,
                                                  unsigned* ticket, const BlockXform* __restrict__ xf,
                                                  double* __restrict__ loc_out, double* lds) {
  constexpr int ACC = UNWARP ? kAccU : kAcc;
  constexpr int STRIPES = kBatchThreads / ACC;
  __shared__ int s_last;
  const int t = threadIdx.x;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's partial stores have left
  __syncthreads();
  if (t == 0) {
    const unsigned arrived = atomicAdd(ticket, 1u);
    s_last = (arrived == eb.num_wg - 1u) ? 1 : 0;
    if (s_last) *ticket = 0u;  // ready for the next launch
  }
  __syncthreads();  // the arrival count has returned to wave 0 before any wave loads a partial
  if (!s_last) return;
  {
    const int j = t / ACC, k = t - j * ACC;
    if (j < STRIPES) {
      double acc = 0.0;
      const double* p = partials + k;
      for (unsigned w = j; w < eb.num_wg; w += 16 * STRIPES) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const unsigned idx = w + u * STRIPES;
          v[u] = idx < eb.num_wg ? load_partial(&p[static_cast<size_t>(idx) * ACC]) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
      }
      lds[j * ACC + k] = acc;
    }
  }
  __syncthreads();
  double* sm = lds + STRIPES * ACC;
  if (t < ACC) {
    double sum = 0.0;
#pragma unroll
    for (int jj = 0; jj < STRIPES; ++jj) sum += lds[jj * ACC + t];
    sm[t] = sum;
  }
  __syncthreads();
  if (UNWARP) {
    if (t < kAccU) loc_out[t] = sm[t];
    return;
  }
  const double* M = xf->M;
  if (t < 78) {
    // entry (lo, hi) of the upper triangle; evaluated as the lower-triangle entry (c1 = hi, c2 = lo):
    // column c1 of M contracted with A7, then with column c2
    int lo = 0, e = t;
    while (e >= 12 - lo) { e -= 12 - lo; ++lo; }
    const int c1 = lo + e, c2 = lo;
    double m1[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) m1[k] = M[k * 12 + c1];
    double v = 0.0;
#pragma unroll
    for (int l = 0; l < 7; ++l) {
      double tt = 0.0;
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const int a = k < l ? k : l, b = k < l ? l : k;
        tt += m1[k] * sm[a * 7 - (a * (a - 1)) / 2 + (b - a)];
      }
      v += tt * M[l * 12 + c2];
    }
    loc_out[t] = v;
  } else if (t < 90) {
    const int c1 = t - 78;
    double gs = 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) gs += M[k * 12 + c1] * sm[28 + k];
    loc_out[t] = gs;
  } else if (t == 90) {
    loc_out[90] = sm[35];
  }
}

// Everything below runs in ONE wavefront: scalars are computed redundantly by every lane,
// vectors/matrices live in LDS and their loops are spread over the lanes.
#ifdef HG_LM_STAMPS
#define HG_STAMP(S, i) do { if (threadIdx.x == 0 && (S).h.stamps[15]) (S).h.stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HG_STAMP(S, i) do {} while (0)
#endif

constexpr int kLoc = kAccU + 1;   // per TSDF block: 78 (upper triangle of its 12 x 12 local system) + 12 + 1, padded

#ifdef HG_BIG
// Big build: the band matrices, the staged local systems and the index tables live in device memory
// (LmState and its work area, L2-resident: 3 x 124 KB of bands alone); the head and the vectors stay in LDS.
// A workgroup's own global stores are visible to its waves behind a barrier (one CU, one vector L1).
struct LmShared {
  LmHead h;
  double* H;
  double* Hc;
  double* A;
  double rhs[kMaxCols], y[kMaxCols];
  double invd[kMaxCols];
  double (*loc)[kLoc];
  double (*small)[kSmallLoc];
  LmTables* Tp;
  double red[kMaxPoses + 8];
  int solve_ok;
  int h_changed;
  LmState* step_G;        // what lm_back needs of lm_front's arguments (nothing stays live in the kernel across the calls)
  BlockXform* step_xf;
};
#define LM_T(S) (*(S).Tp)
typedef double lds_f64;  // the block solvers take generic pointers here (A is in device memory)
typedef int lds_i32;
#else
struct LmShared {
  LmHead h;
  double H[kHCap];
  double Hc[kHCap];
  double A[kHCap + 1];    // + a dump slot for predicated-off stores of the block solver
  double rhs[kMaxCols], y[kMaxCols];
  double invd[kMaxCols];  // reciprocals of the Cholesky diagonal
  alignas(16) double loc[kMaxBlocks][kLoc];  // local normal equations of the TSDF blocks (k_window_residuals tail)
  alignas(16) double small[kMaxSmall][kSmallLoc];   // the same of the odometry / IMU blocks (lower triangle, row-major)
  LmTables T;
  double red[kMaxPoses + 8];  // per-pose partial results (+ scalar slots)
  int solve_ok;           // wavefront 0's factorisation succeeded
  int h_changed;          // this step replaced H (stored back at its end)
  LmState* step_G;        // what lm_back needs of lm_front's arguments (nothing stays live in the kernel across the calls:
  BlockXform* step_xf;    // the copies of the arguments it kept in vector registers were spilled and reloaded around them)
};
#define LM_T(S) ((S).T)
#endif
// ONE instance for the kernels that run the general step and for the functions they call (lm_front / lm_back): at
// file scope, so that the functions address it as LDS without a generic pointer
__shared__ LmShared g_lm_shared;
constexpr int kRedScalar = kMaxPoses + 5;  // slot of S.red that carries a scalar between barriers

__device__ inline double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ inline double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// Candidate = Plus(x, delta): thread p handles control point p. Workgroup-wide (ends with a barrier).
__device__ __forceinline__ void pose_plus(const LmHead& h, const double (*x)[kState], const double* delta,
                                 double (*out)[kState]) {
  if (static_cast<int>(threadIdx.x) < h.num_poses) {
    const int p = threadIdx.x;
    for (int k = 0; k < kState; ++k) out[p][k] = x[p][k];
    if (!h.constant[p]) {
      const double* d = delta + h.col[p];
      for (int k = 0; k < 3; ++k) out[p][k] = x[p][k] + d[k];
      double q[4];
      quaternion_plus(x[p] + 3, d + 3, q);
      for (int k = 0; k < 4; ++k) out[p][3 + k] = q[k];
    }
    if (h.vfree[p]) {
      const double* d = delta + h.vcol[p];
      for (int k = 0; k < 3; ++k) out[p][7 + k] = x[p][7 + k] + d[k];
    }
  }
  __syncthreads();
}

// Right-looking band Cholesky + column-oriented substitutions on an LDS band matrix (W = bw + 1).
// Entries outside the band stay exactly zero in the dense factorisation, so skipping them changes
// nothing; term order per entry equals the sequential left-looking form (k ascending).
// Right-looking band Cholesky, one LDS round trip per column: every lane reads the pivot, the
// (unscaled) column entries and the targets of its NE entries of the trailing triangle
// {(r, k): 0 <= k <= r < bw} (entry p = lane + 64 e), then the updates and the scaled column are
// written. (a * inv) * (b * inv) has the same bits as the product of the stored scaled entries.
// band_index(j + 1 + r, j) etc. are j * W + constant, hoisted out of the column loop.
template <int NE>
__device__ __forceinline__ bool band_factor(int n, int W, lds_f64* A, lds_f64* invd, int lane) {
  const int bw = W - 1, Wm = W - 1;
  const int tri = bw * (bw + 1) / 2;
  int o_li[NE], o_lk[NE], o_tg[NE], er[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int pidx = lane + 64 * e;
    int r = static_cast<int>((sqrtf(8.0f * static_cast<float>(pidx) + 1.0f) - 1.0f) * 0.5f);
    while (r * (r + 1) / 2 > pidx) --r;
    while ((r + 1) * (r + 2) / 2 <= pidx) ++r;
    const int k = pidx - r * (r + 1) / 2;
    const bool in = pidx < tri;
    er[e] = in ? r : 0x7FFFFFFF;  // entry exists in column j iff er < m
    o_li[e] = in ? (1 + r) * Wm + Wm : Wm;
    o_lk[e] = in ? (1 + k) * Wm + Wm : Wm;
    o_tg[e] = in ? (1 + r) * Wm + Wm + 1 + k : Wm;
  }
  const int o_col = (1 + lane) * Wm + Wm;
  const int last_full = (n - 1 - bw > 0 ? n - 1 - bw : 0) * W;  // j * W of the last full-band column
  int jW = 0;
  for (int j = 0; j < n; ++j, jW += W) {
    const double d = A[jW + Wm];  // band_index(j, j)
    if (!(d > 0.0) || !isfinite(d)) return false;  // uniform
    const int m = min(bw, n - j - 1);  // rows below the diagonal inside the band
    // entries outside this column's triangle (er >= m: the last bw columns) read in-range
    // addresses of an earlier column and are discarded
    const int rb = min(jW, last_full);
    double li[NE], lk[NE], tg[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int base = er[e] < m ? jW : rb;
      li[e] = A[base + o_li[e]];
      lk[e] = A[base + o_lk[e]];
      tg[e] = A[base + o_tg[e]];
    }
    const double cj = A[min(jW + o_col, kHCap - 1)];
    const double inv = rsqrt(d);  // l = d * inv, rows *= inv
#pragma unroll
    for (int e = 0; e < NE; ++e)
      if (er[e] < m) A[jW + o_tg[e]] = tg[e] - (li[e] * inv) * (lk[e] * inv);
    if (lane < m) A[jW + o_col] = cj * inv;
    if (lane == 0) {
      A[jW + Wm] = d * inv;
      invd[j] = inv;
    }
    wave_sync();
  }
  return true;
}

// noinline: its register allocation must not be squeezed by the 12 x 12 register solver. The
// matrices live in LDS; the address-space-qualified pointers keep the accesses ds_read / ds_write
// across the call (a generic pointer would turn every access into a flat load + full waitcnt).
__device__ __attribute__((noinline)) bool cholesky_solve_wave(int n, int W, lds_f64* A, const lds_f64* b,
                                                             lds_f64* x, lds_f64* y, lds_f64* invd,
                                                             int lane, long long* dbg = nullptr) {
  const int bw = W - 1;
  if (bw > 40) {
    // dense systems up to 56 x 56 (blocks coupling far-apart control points): plain loop
    for (int j = 0; j < n; ++j) {
      const double d = A[band_index(j, j, W)];
      if (!(d > 0.0) || !isfinite(d)) return false;
      const double inv = rsqrt(d);
      const int m = min(bw, n - j - 1);
      wave_sync();
      if (lane == 0) { A[band_index(j, j, W)] = d * inv; invd[j] = inv; }
      for (int r = lane; r < m; r += kLmThreads) A[band_index(j + 1 + r, j, W)] *= inv;
      wave_sync();
      for (int idx = lane; idx < m * m; idx += kLmThreads) {
        const int i = j + 1 + idx / m, k = j + 1 + idx % m;
        if (k <= i) A[band_index(i, k, W)] -= A[band_index(i, j, W)] * A[band_index(k, j, W)];
      }
      wave_sync();
    }
  } else {
    const int tri = bw * (bw + 1) / 2;
    const int ne = (tri + 63) / 64;  // triangle entries per lane (uniform)
    bool ok;
    switch (ne) {
      case 1: ok = band_factor<1>(n, W, A, invd, lane); break;
      case 2: ok = band_factor<2>(n, W, A, invd, lane); break;
      case 3: ok = band_factor<3>(n, W, A, invd, lane); break;
      case 4: ok = band_factor<4>(n, W, A, invd, lane); break;
      case 5: case 6: ok = band_factor<6>(n, W, A, invd, lane); break;
      case 7: case 8: ok = band_factor<8>(n, W, A, invd, lane); break;
      default: ok = band_factor<13>(n, W, A, invd, lane); break;  // bw <= 40: 820 entries
    }
    if (!ok) return false;
  }
  if (dbg && lane == 0) dbg[9] = __builtin_amdgcn_s_memtime();
  // Substitutions with the vector in registers: lane l holds entries l and l + 64 (n <= 128); the
  // pivot entry travels by v_readlane, so the dependent chain per row is a few ALU instructions
  // instead of two LDS round trips. L's entries for the next row are loaded one row ahead; their
  // addresses are linear in the row (band_index(r, i) = r * (W - 1) + (W - 1) + i).
  (void)y;
  const int Wm1 = W - 1;
  double v0 = lane < n ? b[lane] : 0.0, v1 = lane + 64 < n ? b[lane + 64] : 0.0;
  const double iv0 = lane < n ? invd[lane] : 0.0, iv1 = lane + 64 < n ? invd[lane + 64] : 0.0;
  const int r0 = lane, r1 = lane + 64;
  {
    // forward, column oriented: after y_i is final, rows i < r <= i + bw subtract L[r][i] * y_i
    const int f0 = r0 * Wm1 + Wm1, f1 = r1 * Wm1 + Wm1;  // + i
    auto col = [&](int i, double& c0, double& c1) {
      c0 = (static_cast<unsigned>(r0 - i - 1) < static_cast<unsigned>(bw) && r0 < n) ? A[f0 + i] : 0.0;
      c1 = (static_cast<unsigned>(r1 - i - 1) < static_cast<unsigned>(bw) && r1 < n) ? A[f1 + i] : 0.0;
    };
    double c0, c1;
    col(0, c0, c1);
    const int n_lo = min(n, 64);
    for (int i = 0; i < n_lo; ++i) {
      double n0, n1;
      col(i + 1, n0, n1);
      const double yi = readlane_f64(v0, i) * readlane_f64(iv0, i);
      v0 = (lane == i) ? yi : v0 - c0 * yi;  // c = 0 outside the band / above the pivot row
      v1 -= c1 * yi;
      c0 = n0;
      c1 = n1;
    }
    for (int i = 64; i < n; ++i) {
      double n0, n1;
      col(i + 1, n0, n1);
      const double yi = readlane_f64(v1, i - 64) * readlane_f64(iv1, i - 64);
      v1 = (lane == i - 64) ? yi : v1 - c1 * yi;
      c1 = n1;
      (void)n0;
    }
  }
  if (dbg && lane == 0) dbg[10] = __builtin_amdgcn_s_memtime();
  {
    // backward: after x_i is final, rows i - bw <= r < i subtract L[i][r] * x_i;
    // band_index(i, r) = i * (W - 1) + (W - 1) + r
    auto rowv = [&](int i, double& c0, double& c1) {
      const int base = i * Wm1 + Wm1;
      c0 = (static_cast<unsigned>(i - r0 - 1) < static_cast<unsigned>(bw)) ? A[base + r0] : 0.0;
      c1 = (static_cast<unsigned>(i - r1 - 1) < static_cast<unsigned>(bw)) ? A[base + r1] : 0.0;
    };
    double c0 = 0.0, c1 = 0.0;
    if (n > 0) rowv(n - 1, c0, c1);
    for (int i = n - 1; i >= 64; --i) {
      double n0, n1;
      rowv(i - 1, n0, n1);
      const double xi = readlane_f64(v1, i - 64) * readlane_f64(iv1, i - 64);
      v1 = (lane == i - 64) ? xi : v1 - c1 * xi;
      v0 -= c0 * xi;
      c0 = n0;
      c1 = n1;
    }
    for (int i = min(n, 64) - 1; i >= 0; --i) {
      double n0 = 0.0, n1 = 0.0;
      if (i > 0) rowv(i - 1, n0, n1);
      const double xi = readlane_f64(v0, i) * readlane_f64(iv0, i);
      v0 = (lane == i) ? xi : v0 - c0 * xi;
      c0 = n0;
      (void)n1;
    }
  }
  bool ok = isfinite(v0) && isfinite(v1);
  ok = __all(ok);
  if (lane < n) x[lane] = v0;
  if (lane + 64 < n) x[lane + 64] = v1;
  wave_sync();
  return ok;
}

// Block-tridiagonal systems (LmHead::btd_*): groups of up to 9 columns, every coupling inside a group
// or between neighbouring groups. One wavefront, block by block, the band matrix in LDS updated in
// place (L over A, as the band path leaves it):
//   1. the diagonal block is loaded by every lane (uniform LDS addresses) and factorised in registers,
//      redundantly -- no cross-lane traffic inside the serial chain of its columns;
//   2. lane r solves row r of the sub-diagonal block against it;
//   3. lane e = (r, c) subtracts row r . row c from the next diagonal block (the Schur complement).
// The band path spends ~900 cycles per COLUMN on LDS round trips (81 columns of a ten-control-point
// window: 30 us, 49 us with its substitutions); here a BLOCK costs ~2000 cycles. Groups smaller than 9
// are padded with identity rows in registers. Entries outside the band are structural zeros: they are
// read as 0 and never written (fill-in stays inside the band).
#ifndef HG_BIG
typedef __attribute__((address_space(3))) int lds_i32;
#endif
// Entry (i, j), j <= i, of the band matrix when `on`, else 0 -- as an unconditional load from an address
// that is always valid plus a select, so that a run of reads stays straight-line code with all of its
// LDS loads in flight (as conditional loads every entry became a branch with its own wait).
__device__ inline double band_read(const lds_f64* A, int i, int j, int W, bool on = true) {
  const bool in = on && (i - j < W);
  const double v = A[in ? band_index(i, j, W) : 0];
  return in ? v : 0.0;
}
__device__ __attribute__((noinline)) bool cholesky_solve_btd(int W, lds_f64* A, const lds_f64* b, lds_f64* x,
                                                            lds_f64* y, lds_f64* invd, const lds_i32* start,
                                                            const lds_i32* size, int groups, int lane) {
  constexpr int MB = 9;
  // lane -> entry (er, ec), ec <= er, of the lower triangle of a 9 x 9 block
  int er = static_cast<int>((sqrtf(8.0f * static_cast<float>(lane) + 1.0f) - 1.0f) * 0.5f);
  while (er * (er + 1) / 2 > lane) --er;
  while ((er + 1) * (er + 2) / 2 <= lane) ++er;
  const int ec = lane - er * (er + 1) / 2;
  auto load_diag = [&](int s0, int m, double (*L)[MB]) {
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j)
      {
        const bool in = i < m && j < m;
        const double v = band_read(A, s0 + i, s0 + j, W, in);
        L[i][j] = in ? v : (i == j ? 1.0 : 0.0);
      }
  };
  bool ok = true;
  for (int g = 0; g < groups; ++g) {
    const int s0 = start[g], m = size[g];
    const bool has_next = g + 1 < groups;
    const int s1 = has_next ? start[g + 1] : 0, m1 = has_next ? size[g + 1] : 0;
    double L[MB][MB], inv[MB];
    load_diag(s0, m, L);
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      double d = L[j][j];
#pragma unroll
      for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
      ok = ok && (d > 0.0) && isfinite(d);
      inv[j] = rsqrt(d);
      L[j][j] = d * inv[j];
#pragma unroll
      for (int i = j + 1; i < MB; ++i) {
        double v = L[i][j];
#pragma unroll
        for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
        L[i][j] = v * inv[j];
      }
    }
    if (!ok) return false;  // uniform: every lane holds the same block
    if (lane == 0) {
      // stores that do not apply go to the dump slot A[kHCap]: straight-line code instead of 54 branches
#pragma unroll
      for (int i = 0; i < MB; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j)
          A[(i < m && j < m && i - j < W) ? band_index(s0 + i, s0 + j, W) : kHCap] = L[i][j];
        if (i < m) invd[s0 + i] = inv[i];
      }
    }
    if (has_next) {
      // row `lane` of the sub-diagonal block: X L^T = B
      double X[MB];
#pragma unroll
      for (int c = 0; c < MB; ++c) X[c] = band_read(A, s1 + lane, s0 + c, W, lane < m1 && c < m);
#pragma unroll
      for (int k = 0; k < MB; ++k) {
        double v = X[k];
#pragma unroll
        for (int j = 0; j < k; ++j) v -= X[j] * L[k][j];
        X[k] = v * inv[k];
      }
#pragma unroll
      for (int c = 0; c < MB; ++c)
        A[(lane < m1 && c < m && (s1 + lane) - (s0 + c) < W) ? band_index(s1 + lane, s0 + c, W) : kHCap] = X[c];
      wave_sync();
      if (lane < MB * (MB + 1) / 2 && er < m1 && er - ec < W) {
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < MB; ++k)
          sum += band_read(A, s1 + er, s0 + k, W, k < m) * band_read(A, s1 + ec, s0 + k, W, k < m);
        A[band_index(s1 + er, s1 + ec, W)] -= sum;
      }
    }
    wave_sync();
  }
  // forward substitution L y = b, block by block: lane r forms its row's right-hand side, then every
  // lane solves the diagonal block in registers
  for (int g = 0; g < groups; ++g) {
    const int s0 = start[g], m = size[g];
    const int sp = g > 0 ? start[g - 1] : 0, mp = g > 0 ? size[g - 1] : 0;
    if (lane < m) {
      double t = b[s0 + lane];
#pragma unroll
      for (int c = 0; c < MB; ++c) t -= band_read(A, s0 + lane, sp + c, W, c < mp) * y[sp + (c < mp ? c : 0)];
      y[s0 + lane] = t;
    }
    wave_sync();
    double L[MB][MB], yy[MB];
    load_diag(s0, m, L);
#pragma unroll
    for (int k = 0; k < MB; ++k) {
      const double v = y[s0 + (k < m ? k : 0)];
      yy[k] = k < m ? v : 0.0;
    }
    double dinv[MB];
#pragma unroll
    for (int k = 0; k < MB; ++k) {
      const double v = invd[s0 + (k < m ? k : 0)];
      dinv[k] = k < m ? v : 1.0;
    }
#pragma unroll
    for (int k = 0; k < MB; ++k) {
      double v = yy[k];
#pragma unroll
      for (int j = 0; j < k; ++j) v -= L[k][j] * yy[j];
      yy[k] = v * dinv[k];
    }
    wave_sync();
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < MB; ++k)
        if (k < m) y[s0 + k] = yy[k];
    }
    wave_sync();
  }
  // backward substitution L^T x = y
  for (int g = groups - 1; g >= 0; --g) {
    const int s0 = start[g], m = size[g];
    const bool has_next = g + 1 < groups;
    const int sn = has_next ? start[g + 1] : 0, mn = has_next ? size[g + 1] : 0;
    if (lane < m) {
      double t = y[s0 + lane];
#pragma unroll
      for (int r = 0; r < MB; ++r) t -= band_read(A, sn + r, s0 + lane, W, r < mn) * x[sn + (r < mn ? r : 0)];
      x[s0 + lane] = t;
    }
    wave_sync();
    double L[MB][MB], xx[MB];
    load_diag(s0, m, L);
#pragma unroll
    for (int k = 0; k < MB; ++k) {
      const double v = x[s0 + (k < m ? k : 0)];
      xx[k] = k < m ? v : 0.0;
    }
    double dinv[MB];
#pragma unroll
    for (int k = 0; k < MB; ++k) {
      const double v = invd[s0 + (k < m ? k : 0)];
      dinv[k] = k < m ? v : 1.0;
    }
#pragma unroll
    for (int k = MB - 1; k >= 0; --k) {
      double v = xx[k];
#pragma unroll
      for (int j = k + 1; j < MB; ++j) v -= L[j][k] * xx[j];
      xx[k] = v * dinv[k];
      ok = ok && isfinite(xx[k]);
    }
    wave_sync();
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < MB; ++k)
        if (k < m) x[s0 + k] = xx[k];
    }
    wave_sync();
  }
  return ok;
}

// The same for the usual window: every group has MB columns (6 pose, or 6 pose + 3 velocity) and the band
// holds the whole 2 MB x 2 MB coupling of neighbouring groups (bw >= 2 MB - 1), so nothing is padded or
// predicated and entry (i, j) sits at i * (W - 1) + (W - 1) + j. The forward substitution rides along
// with the factorisation: the right-hand side is treated as one more row below each sub-diagonal block
// (its Schur update by spare lanes, its triangular solve in registers next to the block's), which
// leaves only the backward pass as a second sweep over the blocks.
template <int MB>
__device__ __attribute__((noinline)) bool cholesky_solve_btd_full(int W, lds_f64* A, const lds_f64* b, lds_f64* x,
                                                                 lds_f64* y, lds_f64* invd, int groups, int lane) {
  const int Wm = W - 1;
  constexpr int kTri = MB * (MB + 1) / 2;
  int er = static_cast<int>((sqrtf(8.0f * static_cast<float>(lane) + 1.0f) - 1.0f) * 0.5f);
  while (er * (er + 1) / 2 > lane) --er;
  while ((er + 1) * (er + 2) / 2 <= lane) ++er;
  const int ec = lane - er * (er + 1) / 2;
  const int rl = lane - 48;  // lanes 48 .. 48 + MB - 1 carry the right-hand side of the next group
  if (lane < MB) y[lane] = b[lane];
  wave_sync();
  bool ok = true;
  for (int g = 0; g < groups; ++g) {
    const int s0 = g * MB, s1 = s0 + MB;
    const bool has_next = g + 1 < groups;
    const int base = s0 * Wm + Wm + s0;  // entry (s0, s0)
    double L[MB][MB], inv[MB], yy[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) L[i][j] = A[base + i * Wm + j];
#pragma unroll
    for (int k = 0; k < MB; ++k) yy[k] = y[s0 + k];
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      double d = L[j][j];
#pragma unroll
      for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
      ok = ok && (d > 0.0) && isfinite(d);
      inv[j] = rsqrt(d);
      L[j][j] = d * inv[j];
#pragma unroll
      for (int i = j + 1; i < MB; ++i) {
        double v = L[i][j];
#pragma unroll
        for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
        L[i][j] = v * inv[j];
      }
    }
    if (!ok) return false;  // uniform
    // y_g = L^-1 (b_g - L_{g,g-1} y_{g-1}): the bracket was left in y[] by the previous group
#pragma unroll
    for (int k = 0; k < MB; ++k) {
      double v = yy[k];
#pragma unroll
      for (int j = 0; j < k; ++j) v -= L[k][j] * yy[j];
      yy[k] = v * inv[k];
    }
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < MB; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) A[base + i * Wm + j] = L[i][j];
        invd[s0 + i] = inv[i];
        y[s0 + i] = yy[i];
      }
    }
    if (has_next) {
      const int row = (s1 + (lane < MB ? lane : 0)) * Wm + Wm + s0;  // entry (s1 + lane, s0)
      double X[MB];
#pragma unroll
      for (int c = 0; c < MB; ++c) X[c] = A[row + c];
#pragma unroll
      for (int k = 0; k < MB; ++k) {
        double v = X[k];
#pragma unroll
        for (int j = 0; j < k; ++j) v -= X[j] * L[k][j];
        X[k] = v * inv[k];
      }
      if (lane < MB) {
#pragma unroll
        for (int c = 0; c < MB; ++c) A[row + c] = X[c];
      }
      wave_sync();
      if (lane < kTri) {
        const int ra = (s1 + er) * Wm + Wm + s0, rb = (s1 + ec) * Wm + Wm + s0;
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < MB; ++k) sum += A[ra + k] * A[rb + k];
        A[(s1 + er) * Wm + Wm + s1 + ec] -= sum;
      } else if (rl >= 0 && rl < MB) {
        const int ra = (s1 + rl) * Wm + Wm + s0;
        double t = b[s1 + rl];
#pragma unroll
        for (int k = 0; k < MB; ++k) t -= A[ra + k] * yy[k];
        y[s1 + rl] = t;
      }
    }
    wave_sync();
  }
  // backward substitution L^T x = y
  for (int g = groups - 1; g >= 0; --g) {
    const int s0 = g * MB, sn = s0 + MB;
    const bool has_next = g + 1 < groups;
    if (lane < MB) {
      double t = y[s0 + lane];
      if (has_next) {
#pragma unroll
        for (int r = 0; r < MB; ++r) t -= A[(sn + r) * Wm + Wm + s0 + lane] * x[sn + r];
      }
      x[s0 + lane] = t;
    }
    wave_sync();
    const int base = s0 * Wm + Wm + s0;
    double L[MB][MB], xx[MB], dinv[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) L[i][j] = A[base + i * Wm + j];
#pragma unroll
    for (int k = 0; k < MB; ++k) {
      xx[k] = x[s0 + k];
      dinv[k] = invd[s0 + k];
    }
#pragma unroll
    for (int k = MB - 1; k >= 0; --k) {
      double v = xx[k];
#pragma unroll
      for (int j = k + 1; j < MB; ++j) v -= L[j][k] * xx[j];
      xx[k] = v * dinv[k];
      ok = ok && isfinite(xx[k]);
    }
    wave_sync();
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < MB; ++k) x[s0 + k] = xx[k];
    }
    wave_sync();
  }
  return ok;
}

// The same system by block CYCLIC REDUCTION over the whole workgroup. cholesky_solve_btd_full is a chain of
// `groups` block steps in one wavefront (81 columns: nine steps of ~7k cycles); the groups at the even
// positions of the chain do not touch each other, so they are eliminated AT THE SAME TIME by different
// wavefronts, the odd ones form a chain of half the length, and so on: 9 -> 4 -> 2 -> 1 -> 0, four levels.
// Per level, wavefront w takes the group p at position 2w with its kept neighbours l (before) and r (after):
//   phase 1  L_p = chol(D_p) in registers (every lane), y_p = L_p^-1 b_p; lanes 0.. solve the rows of
//            X_l = A_lp L_p^-T, lanes 16.. those of X_r = A_rp L_p^-T; the new coupling of r with l is
//            -X_r X_l^T (fill);
//   phase 2  every kept group q (wavefront per group) subtracts X X^T of its eliminated neighbours from
//            D_q and X y from b_q;
// then the levels are unwound: x_p = L_p^-T (y_p - X_l^T x_l - X_r^T x_r). Same factorisation up to the
// elimination order (a symmetric permutation of the system), hence the same solution to rounding.
// Workspace: dense blocks in LDS (`ws`, kCrStride doubles per group).
// (kCrStride = 5 * 81 + 3 * 9 + 5: D/L, E, Xl, Xr, fill scratch | b/y, x, inv)
template <int MB>
__device__ __forceinline__ bool cholesky_solve_cr(int W, const double* A, const double* b, double* x, double* ws,
                                                  int groups, int* ok_flag) {
  constexpr int MM = MB * MB;
  constexpr int oD = 0, oE = 81, oXl = 162, oXr = 243, oB = 405, oX = 414, oI = 423;
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int wave = tid / kWave, lane = tid % kWave;
  const int Wm = W - 1;
  // dense copies of the diagonal blocks (lower triangle), the couplings with the next group, the rhs
  for (int i = tid; i < groups * MM; i += nthreads) {
    const int g = i / MM, e = i - g * MM, r = e / MB, c = e - r * MB;
    const int s0 = g * MB;
    if (c <= r) ws[g * kCrStride + oD + e] = A[(s0 + r) * Wm + Wm + s0 + c];
    if (g + 1 < groups) ws[g * kCrStride + oE + e] = A[(s0 + MB + r) * Wm + Wm + s0 + c];  // rows: next group, cols: g
  }
  for (int i = tid; i < groups * MB; i += nthreads) ws[(i / MB) * kCrStride + oB + i % MB] = b[i];
  if (tid == 0) *ok_flag = 1;
  __syncthreads();
  // The chain of level v holds the groups (k + 1) 2^v - 1, k = 0 .. (groups >> v) - 1: the kept (odd)
  // positions of the level before.
  auto chain_at = [](int v, int k) { return ((k + 1) << v) - 1; };
  auto chain_len = [groups](int v) { return groups >> v; };
  int levels = 0;
  while (chain_len(levels) > 0) ++levels;
  int er = static_cast<int>((sqrtf(8.0f * static_cast<float>(lane) + 1.0f) - 1.0f) * 0.5f);
  while (er * (er + 1) / 2 > lane) --er;
  while ((er + 1) * (er + 2) / 2 <= lane) ++er;
  const int ec = lane - er * (er + 1) / 2;
  for (int v = 0; v < levels; ++v) {
    const int len_v = chain_len(v);
    const int n_el = (len_v + 1) / 2, n_keep = len_v / 2;
    if (wave < n_el) {  // ---- phase 1: eliminate the group at position 2 * wave
      const int p = chain_at(v, 2 * wave);
      const int l = 2 * wave - 1 >= 0 ? chain_at(v, 2 * wave - 1) : -1;
      const int r = 2 * wave + 1 < len_v ? chain_at(v, 2 * wave + 1) : -1;
      double* wp = ws + p * kCrStride;
      double L[MB][MB], inv[MB], yy[MB];
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) L[i][j] = wp[oD + i * MB + j];
#pragma unroll
      for (int k = 0; k < MB; ++k) yy[k] = wp[oB + k];
      bool ok = true;
#pragma unroll
      for (int j = 0; j < MB; ++j) {
        double d = L[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        ok = ok && (d > 0.0) && isfinite(d);
        inv[j] = rsqrt(d);
        L[j][j] = d * inv[j];
#pragma unroll
        for (int i = j + 1; i < MB; ++i) {
          double t = L[i][j];
#pragma unroll
          for (int k = 0; k < j; ++k) t -= L[i][k] * L[j][k];
          L[i][j] = t * inv[j];
        }
      }
      if (!ok && lane == 0) *ok_flag = 0;
#pragma unroll
      for (int k = 0; k < MB; ++k) {
        double t = yy[k];
#pragma unroll
        for (int j = 0; j < k; ++j) t -= L[k][j] * yy[j];
        yy[k] = t * inv[k];
      }
      // rows of X_l (lanes 0 .. MB - 1: row i of A_lp = column i of the coupling block of l, whose rows are
      // p's) and of X_r (lanes 16 .. 16 + MB - 1: row i of the coupling block of p, whose rows are r's)
      {
        const bool right = lane >= 16;
        const int i = lane & 15;
        const bool on = i < MB && lane < 32 && (right ? r >= 0 : l >= 0);
        const double* src = right ? wp + oE : ws + (l >= 0 ? l : 0) * kCrStride + oE;
        const int base0 = on ? (right ? i * MB : i) : 0, stride = right ? 1 : MB;
        double X[MB];
#pragma unroll
        for (int k = 0; k < MB; ++k) X[k] = src[base0 + k * stride];
#pragma unroll
        for (int k = 0; k < MB; ++k) {
          double t = X[k];
#pragma unroll
          for (int j = 0; j < k; ++j) t -= X[j] * L[k][j];
          X[k] = t * inv[k];
        }
        if (on) {
          double* dst = wp + (right ? oXr : oXl) + i * MB;
#pragma unroll
          for (int k = 0; k < MB; ++k) dst[k] = X[k];
        }
      }
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < MB; ++i) {
#pragma unroll
          for (int j = 0; j <= i; ++j) wp[oD + i * MB + j] = L[i][j];
          wp[oI + i] = inv[i];
          wp[oB + i] = yy[i];
        }
      }
      wave_sync();
      if (l >= 0 && r >= 0) {  // fill: coupling of r (rows) with l (columns), replaces l's coupling with p
        double* fe = ws + l * kCrStride + oE;
        for (int e = lane; e < MM; e += kWave) {
          const int i = e / MB, j = e - i * MB;
          double sum = 0.0;
#pragma unroll
          for (int k = 0; k < MB; ++k) sum += wp[oXr + i * MB + k] * wp[oXl + j * MB + k];
          fe[e] = -sum;
        }
      }
    }
    __syncthreads();
    if (wave < n_keep) {  // ---- phase 2: the kept group at position 2 * wave + 1 takes its neighbours' updates
      const int q = chain_at(v, 2 * wave + 1);
      const int pp = chain_at(v, 2 * wave);
      const int pn = 2 * wave + 2 < len_v ? chain_at(v, 2 * wave + 2) : -1;
      double* wq = ws + q * kCrStride;
      const double* xa = ws + pp * kCrStride + oXr;                       // q is pp's right neighbour
      const double* xb = ws + (pn >= 0 ? pn : pp) * kCrStride + oXl;      // and pn's left neighbour
      const double* ya = ws + pp * kCrStride + oB;
      const double* yb = ws + (pn >= 0 ? pn : pp) * kCrStride + oB;
      if (lane < MB * (MB + 1) / 2) {
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < MB; ++k) sum += xa[er * MB + k] * xa[ec * MB + k];
        if (pn >= 0) {
#pragma unroll
          for (int k = 0; k < MB; ++k) sum += xb[er * MB + k] * xb[ec * MB + k];
        }
        wq[oD + er * MB + ec] -= sum;
      } else if (lane >= 48 && lane < 48 + MB) {
        const int i = lane - 48;
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < MB; ++k) sum += xa[i * MB + k] * ya[k];
        if (pn >= 0) {
#pragma unroll
          for (int k = 0; k < MB; ++k) sum += xb[i * MB + k] * yb[k];
        }
        wq[oB + i] -= sum;
      }
    }
    __syncthreads();
  }
  // ---- unwind: x_p = L_p^-T (y_p - X_l^T x_l - X_r^T x_r), last level first
  for (int v = levels - 1; v >= 0; --v) {
    const int len_v = chain_len(v);
    const int n_el = (len_v + 1) / 2;
    if (wave < n_el) {
      const int p = chain_at(v, 2 * wave);
      const int l = 2 * wave - 1 >= 0 ? chain_at(v, 2 * wave - 1) : -1;
      const int r = 2 * wave + 1 < len_v ? chain_at(v, 2 * wave + 1) : -1;
      double* wp = ws + p * kCrStride;
      if (lane < MB) {
        double t = wp[oB + lane];
        if (l >= 0) {
          const double* xl = ws + l * kCrStride + oX;
#pragma unroll
          for (int i = 0; i < MB; ++i) t -= wp[oXl + i * MB + lane] * xl[i];
        }
        if (r >= 0) {
          const double* xr = ws + r * kCrStride + oX;
#pragma unroll
          for (int i = 0; i < MB; ++i) t -= wp[oXr + i * MB + lane] * xr[i];
        }
        wp[oX + lane] = t;
      }
      wave_sync();
      double L[MB][MB], xx[MB], dinv[MB];
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) L[i][j] = wp[oD + i * MB + j];
#pragma unroll
      for (int k = 0; k < MB; ++k) {
        xx[k] = wp[oX + k];
        dinv[k] = wp[oI + k];
      }
#pragma unroll
      for (int k = MB - 1; k >= 0; --k) {
        double t = xx[k];
#pragma unroll
        for (int j = k + 1; j < MB; ++j) t -= L[j][k] * xx[j];
        xx[k] = t * dinv[k];
      }
      wave_sync();
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < MB; ++k) wp[oX + k] = xx[k];
      }
    }
    __syncthreads();
  }
  bool ok = *ok_flag != 0;
  for (int i = tid; i < groups * MB; i += nthreads) {
    const double v = ws[(i / MB) * kCrStride + oX + i % MB];
    x[i] = v;
    if (!isfinite(v)) *ok_flag = 0;  // benign race: same value
  }
  __syncthreads();
  return ok && *ok_flag != 0;
}

#include "hg_btd.h"  // cholesky_solve_twisted
static_assert(kTwWs <= kCrStride, "the twisted factorisation shares the cyclic-reduction workspace");

// Small systems: every lane factorises its own register copy (no LDS round trips, no barriers);
// left-looking term order; one reciprocal per column instead of a division per entry.
template <int N>
__device__ __attribute__((noinline)) bool cholesky_solve_regs(const lds_f64* A_lds, int W, const lds_f64* b_lds,
                                                              lds_f64* x_lds, int lane) {
  double L[N][N], inv[N], y[N], x[N];
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) L[i][j] = (i - j < W) ? A_lds[band_index(i, j, W)] : 0.0;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double d = L[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
    ok = ok && (d > 0.0) && isfinite(d);
    const double l = sqrt(d);
    L[j][j] = l;
    inv[j] = 1.0 / l;
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      double s = L[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
      L[i][j] = s * inv[j];
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double s = b_lds[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
    y[i] = s * inv[i];
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    double s = y[i];
#pragma unroll
    for (int k = i + 1; k < N; ++k) s -= L[k][i] * x[k];
    x[i] = s * inv[i];
  }
#pragma unroll
  for (int i = 0; i < N; ++i) ok = ok && isfinite(x[i]);
  wave_sync();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) x_lds[i] = x[i];
  }
  wave_sync();
  return ok;
}

__device__ inline void finish(LmHead& h, int type, int reason) {
  h.done = 1;
  h.termination_type = type;
  h.termination_reason = reason;
}

// |x - Plus(x, -g)|_inf in the ambient space (TrustRegionMinimizer::EvaluateGradientAndJacobian):
// thread p takes control point p, every thread returns the maximum. Workgroup-wide.
__device__ __forceinline__ double gradient_max_norm(LmShared& S) {
  const LmHead& h = S.h;
  if (static_cast<int>(threadIdx.x) < h.num_poses) {
    const int p = threadIdx.x;
    double m = 0.0;
    if (!h.constant[p]) {
      const double* g = h.g + h.col[p];
      const double neg[6] = {-g[0], -g[1], -g[2], -g[3], -g[4], -g[5]};
      for (int k = 0; k < 3; ++k) m = fmax(m, fabs(h.x[p][k] - (h.x[p][k] + neg[k])));
      double q[4];
      quaternion_plus(h.x[p] + 3, neg + 3, q);
      for (int k = 0; k < 4; ++k) m = fmax(m, fabs(h.x[p][3 + k] - q[k]));
    }
    if (h.vfree[p]) {
      const double* g = h.g + h.vcol[p];
      for (int k = 0; k < 3; ++k) m = fmax(m, fabs(h.x[p][7 + k] - (h.x[p][7 + k] - g[k])));
    }
    S.red[p] = m;
  }
  // the maximum over the control points: one butterfly in wavefront 0 per 64 of them (a maximum does not depend on the
  // order), one barrier instead of two and a loop over the poses in every thread
  if (threadIdx.x < kWave) {
    wave_sync();  // (control point p's thread is lane p of this wavefront when p < 64)
    double m = 0.0;
    for (int p = threadIdx.x; p < h.num_poses; p += kWave) m = fmax(m, S.red[p]);
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    if (threadIdx.x == 0) S.red[kRedScalar + 1] = m;
  }
  __syncthreads();
  return S.red[kRedScalar + 1];
}

// LevenbergMarquardtStrategy::ComputeStep + TrustRegionMinimizer::ComputeTrustRegionStep,
// looping over invalid steps (each one is an iteration). Leaves the next candidate in h.cand
// or terminates. Workgroup-wide: every thread evaluates the (uniform) predicates from the head in
// LDS, thread 0 updates its scalars between barriers, the loops over the matrix are spread over all
// threads; the factorisation itself runs in wavefront 0.
// (Round 5: in three pieces. The factorisation is the only part of a step that needs the whole register file, and
// inside one function with the rest of the step either its state or the step's loop invariants went to scratch memory
// -- or, as a function of its own, its 31 callee-saved registers did, whose reload alone cost 1.9 us per step. Now the
// kernel body holds nothing but attempt_solve; what comes before and behind it are the functions lm_front / lm_back.)
// attempt_begin: FinalizeIterationAndCheckIfMinimizerCanContinue, the LM diagonal, the scaled system. False: the
// solve has terminated. attempt_solve: the factorisation. attempt_end: model cost change, step validity, the candidate;
// true: the step was invalid, the radius has been reduced -- begin again.
__device__ __forceinline__ bool attempt_begin(LmShared& S) {
  LmHead& h = S.h;
  const int n = h.ncols, W = h.bw + 1;
  const int tid = threadIdx.x, nthreads = blockDim.x;
  {
    __syncthreads();
    // FinalizeIterationAndCheckIfMinimizerCanContinue (all threads evaluate the same predicates)
    const bool stop_iter = h.iteration >= h.opt.max_num_iterations;
    const bool stop_grad = h.step_is_successful && h.gradient_max_norm <= h.opt.gradient_tolerance;
    const bool stop_rad = h.radius <= h.opt.min_trust_region_radius;
    const bool reuse = h.reuse_diagonal != 0;
    __syncthreads();
    if (tid == 0) {
      if (h.step_is_successful) ++h.num_successful; else ++h.num_unsuccessful;
      if (stop_iter) finish(h, 1, 4);
      else if (stop_grad) finish(h, 0, 1);
      else if (stop_rad) finish(h, 0, 5);
      else {
        ++h.iteration;
        ++h.num_iterations;
        h.step_is_successful = 0;
      }
    }
    if (stop_iter || stop_grad || stop_rad) {
      __syncthreads();
      return false;
    }
    {
      // A = scaled H + LM diagonal; rows walked without an integer division per entry. The thread of a diagonal entry
      // also renews the LM diagonal of its column when it is not reused (a pass and a barrier of its own before)
      const double radius = h.radius;
      const double dmin = h.opt.min_lm_diagonal, dmax = h.opt.max_lm_diagonal;
      int a = tid / W, c = tid - a * W;
      const int da = nthreads / W, dc = nthreads - da * W;
      for (int idx = tid; idx < n * W; idx += nthreads) {
        const int b = a - (W - 1) + c;
        double v = 0.0;
        if (b >= 0) {
          v = S.H[idx] * h.scale[a] * h.scale[b];
          if (a == b) {
            double dg = h.diagonal[a];
            if (!reuse) {
              dg = fmin(fmax(v, dmin), dmax);  // (v = H_aa scale_a scale_a: the same product)
              h.diagonal[a] = dg;
            }
            const double lm = sqrt(dg / radius);
            v += lm * lm;
          }
        }
        S.A[idx] = v;
        a += da;
        c += dc;
        if (c >= W) { c -= W; ++a; }
      }
    }
    for (int a = tid; a < n; a += nthreads) S.rhs[a] = h.g[a] * h.scale[a];
    __syncthreads();
    HG_STAMP(S, 4);
  }
  return true;
}

__device__ __forceinline__ void attempt_solve(LmShared& S) {
  LmHead& h = S.h;
  const int n = h.ncols, W = h.bw + 1;
  const int tid = threadIdx.x;
  {
    // uniform 6 / 9-column groups, three or more of them: cyclic reduction over all wavefronts
    const bool use_cr = h.btd_groups >= 3 && (h.btd_uniform == 9 || h.btd_uniform == 6) && h.btd_cr != 0 &&
                        (h.btd_groups + 1) / 2 <= static_cast<int>(blockDim.x) / kWave;
    const bool use_tw = h.btd_groups >= 2 && h.btd_groups <= 9 && (h.btd_uniform == 9 || h.btd_uniform == 6) &&
                        h.btd_cr == 2 && static_cast<int>(blockDim.x) >= 8 * kWave;
    if (use_tw) {
#ifdef HG_BIG
      double* ws = S.A + (kHCap + 2);  // LmState::cr_ws follows A
#else
      double* ws = &S.loc[0][0];  // (the local systems were consumed by the assembly)
#endif
      if (h.btd_uniform == 9)
        cholesky_solve_twisted<9>(W, (lds_f64*)S.A, (lds_f64*)S.rhs, (lds_f64*)h.step, (lds_f64*)ws, h.btd_groups,
                                  (lds_i32*)&S.solve_ok);
      else
        cholesky_solve_twisted<6>(W, (lds_f64*)S.A, (lds_f64*)S.rhs, (lds_f64*)h.step, (lds_f64*)ws, h.btd_groups,
                                  (lds_i32*)&S.solve_ok);
    } else if (use_cr) {
#ifdef HG_BIG
      double* ws = S.small ? nullptr : nullptr;
      ws = S.A + (kHCap + 2);  // LmState::cr_ws follows A
      static_assert(offsetof(LmState, cr_ws) == offsetof(LmState, A) + sizeof(double) * (kHCap + 2), "cr_ws behind A");
#else
      double* ws = &S.loc[0][0];  // (the local systems were consumed by the assembly)
      static_assert(sizeof(S.loc) + sizeof(S.small) >= sizeof(double) * kCrStride * kMaxPoses, "cyclic-reduction workspace");
#endif
      if (h.btd_uniform == 9) cholesky_solve_cr<9>(W, S.A, S.rhs, h.step, ws, h.btd_groups, &S.solve_ok);
      else cholesky_solve_cr<6>(W, S.A, S.rhs, h.step, ws, h.btd_groups, &S.solve_ok);
    } else if (tid < kLmThreads) {
      const int lane = tid;
      bool valid = (n == 6)    ? cholesky_solve_regs<6>((const lds_f64*)S.A, W, (const lds_f64*)S.rhs, (lds_f64*)h.step, lane)
                   : (n == 12) ? cholesky_solve_regs<12>((const lds_f64*)S.A, W, (const lds_f64*)S.rhs, (lds_f64*)h.step, lane)
                   : (h.btd_groups >= 2 && h.btd_uniform == 9)
                       ? cholesky_solve_btd_full<9>(W, (lds_f64*)S.A, (const lds_f64*)S.rhs, (lds_f64*)h.step,
                                                    (lds_f64*)S.y, (lds_f64*)S.invd, h.btd_groups, lane)
                   : (h.btd_groups >= 2 && h.btd_uniform == 6)
                       ? cholesky_solve_btd_full<6>(W, (lds_f64*)S.A, (const lds_f64*)S.rhs, (lds_f64*)h.step,
                                                    (lds_f64*)S.y, (lds_f64*)S.invd, h.btd_groups, lane)
                   : (h.btd_groups >= 2)
                       ? cholesky_solve_btd(W, (lds_f64*)S.A, (const lds_f64*)S.rhs, (lds_f64*)h.step, (lds_f64*)S.y,
                                            (lds_f64*)S.invd, (const lds_i32*)h.btd_start, (const lds_i32*)h.btd_size,
                                            h.btd_groups, lane)
#ifdef HG_LM_STAMPS
                               : cholesky_solve_wave(n, W, (lds_f64*)S.A, (const lds_f64*)S.rhs, (lds_f64*)h.step,
                                                     (lds_f64*)S.y, (lds_f64*)S.invd, lane,
                                                     h.stamps[15] ? h.stamps : nullptr);
#else
                               : cholesky_solve_wave(n, W, (lds_f64*)S.A, (const lds_f64*)S.rhs, (lds_f64*)h.step,
                                                     (lds_f64*)S.y, (lds_f64*)S.invd, lane);
#endif
      if (lane == 0) S.solve_ok = valid ? 1 : 0;
    }
    HG_STAMP(S, 5);
    __syncthreads();
  }
}

__device__ __forceinline__ bool attempt_end(LmShared& S) {
  LmHead& h = S.h;
  const int n = h.ncols;
  const int tid = threadIdx.x;
  {
    bool valid = S.solve_ok != 0;
    double mcc = 0.0;
    if (valid) {
      // model_cost_change = -(step . g_s + step . H_s step / 2) on the scaled system, step = -y. y solves
      // (H_s + D) y = g_s with D the LM diagonal the system was built with, so H_s step = -g_s - D step and the value is
      // sum_a step_a (D_a step_a - g_s,a) / 2: one pass over the columns in wavefront 0 (lane-strided partial sums,
      // then a butterfly) where the band matrix-vector product took four barriers and 35 dependent LDS round trips per
      // row (mcc + candidate 8.4k cycles -> DESIGN.md 3.3). Equal to the product form up to the residual of the solve.
      if (tid < kLmThreads) {
        const double radius = h.radius;
        double part = 0.0;
        for (int a = tid; a < n; a += kLmThreads) {
          const double st = -h.step[a];
          h.step[a] = st;
          const double lm = sqrt(h.diagonal[a] / radius);
          part += 0.5 * (st * (lm * lm * st - h.g[a] * h.scale[a]));
        }
        const double m = wave_sum(part);
        if (tid == 0) S.red[kRedScalar] = m;
      }
      __syncthreads();
      mcc = S.red[kRedScalar];
      valid = mcc > 0.0;
    }
    __syncthreads();
    if (!valid) {
      const bool fail = (h.invalid_steps + 1) >= 5;  // max_num_consecutive_invalid_steps
      __syncthreads();
      if (tid == 0) {
        ++h.invalid_steps;
        h.reuse_diagonal = 1;
        if (fail) finish(h, 2, 6);
        else {
          h.radius = h.radius / h.decrease_factor;
          h.decrease_factor *= 2.0;
        }
      }
      __syncthreads();
      return !fail;
    }
    if (tid == 0) {
      h.reuse_diagonal = 1;
      h.invalid_steps = 0;
      h.model_cost_change = mcc;
    }
    // delta = step * scale: control point p's thread writes the columns it is about to read in pose_plus (no pass and
    // barrier of their own)
    if (tid < h.num_poses) {
      const int p = tid;
      if (!h.constant[p])
        for (int k = 0; k < 6; ++k) h.delta[h.col[p] + k] = h.step[h.col[p] + k] * h.scale[h.col[p] + k];
      if (h.vfree[p])
        for (int k = 0; k < 3; ++k) h.delta[h.vcol[p] + k] = h.step[h.vcol[p] + k] * h.scale[h.vcol[p] + k];
    }
    pose_plus(h, h.x, h.delta, h.cand);
    return false;
  }
}

// Local column of global column `col` in a block over control points (pa, pb), or -1. TSDF blocks:
// [pose_a 6 | pose_b 6]; odometry / IMU blocks: [pose_a 6 | vel_a 3 | pose_b 6 | vel_b 3].
__device__ inline int local_col_tsdf(int cp, int pa, int pb) {
  const int p = cp >> 4, sl = cp & 15;
  const int l = p == pa ? sl : (p == pb ? 6 + sl : -1);
  return sl >= 6 ? -1 : l;
}
__device__ inline int local_col_small(int cp, int pa, int pb) {
  const int p = cp >> 4, sl = cp & 15;
  return p == pa ? sl : (p == pb ? 9 + sl : -1);
}

// Normal equations at the candidate from the blocks' local systems: every thread GATHERS its band
// entries (for each entry the blocks that cover both of its columns, in block order), so no entry is
// written twice and there is nothing to synchronise. The loops over the blocks are branch-free (a
// block that does not cover the entry adds 0.0 read from a valid address), so their LDS reads are
// independent of each other and pipeline. Workgroup-wide.
__device__ __forceinline__ void assemble(LmShared& S) {
  LmHead& h = S.h;
  const int n = h.ncols, W = h.bw + 1;
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int nb = h.num_blocks, ns = h.num_small;
  int a = tid / W, c = tid - a * W;
  const int da = nthreads / W, dc = nthreads - da * W;
  const bool by_pairs = LM_T(S).pair_overflow == 0;
  for (int idx = tid; idx < n * W; idx += nthreads) {
    const int b = a - (W - 1) + c;  // entry (a, b), b <= a
    double v = 0.0;
    if (b >= 0 && by_pairs) {
      // the blocks that cover both columns' control points, from the pair's list
      const int cpa = LM_T(S).colcp[a], cpb = LM_T(S).colcp[b];
      const int p = cpa >> 4, q = cpb >> 4, sa = cpa & 15, sb = cpb & 15;
      const bool sw = p < q;  // the list is kept for (larger, smaller) control point
      const int hi_p = sw ? q : p, lo_p = sw ? p : q;
      const int* list = LM_T(S).pair_list[hi_p][lo_p];
      const int cnt = LM_T(S).pair_count[hi_p][lo_p];
      for (int k = 0; k < cnt; ++k) {
        const int e = list[k];
        const int blk = e & 255, small = (e >> 8) & 1;
        const int o_hi = (e >> 12) & 255, o_lo = (e >> 20) & 255;
        const int l1 = sa + (sw ? o_lo : o_hi), l2 = sb + (sw ? o_hi : o_lo);
        const int lo = l1 < l2 ? l1 : l2, hi = l1 < l2 ? l2 : l1;
        const bool on = small || (sa < 6 && sb < 6);
        const double* src = small ? S.small[blk] : S.loc[blk];
        const int at = small ? hi * (hi + 1) / 2 + lo : lo * 12 - (lo * (lo - 1)) / 2 + (hi - lo);
        const double t = src[on ? at : 0];
        v += on ? t : 0.0;
      }
    } else if (b >= 0) {
      const int cpa = LM_T(S).colcp[a], cpb = LM_T(S).colcp[b];
#pragma unroll 4
      for (int k = 0; k < nb; ++k) {
        const int d = LM_T(S).desc[k];
        const int pa = d & 255, pb = (d >> 8) & 255;
        const int l1 = local_col_tsdf(cpa, pa, pb), l2 = local_col_tsdf(cpb, pa, pb);
        const bool on = (d >> 16) && l1 >= 0 && l2 >= 0;
        const int lo = l1 < l2 ? l1 : l2, hi = l1 < l2 ? l2 : l1;
        const double t = S.loc[k][on ? lo * 12 - (lo * (lo - 1)) / 2 + (hi - lo) : 0];
        v += on ? t : 0.0;
      }
#pragma unroll 4
      for (int k = 0; k < ns; ++k) {
        const int d = LM_T(S).desc[kMaxBlocks + k];
        const int pa = d & 255, pb = (d >> 8) & 255;
        const int l1 = local_col_small(cpa, pa, pb), l2 = local_col_small(cpb, pa, pb);
        const bool on = (d >> 16) && l1 >= 0 && l2 >= 0;
        const int lo = l1 < l2 ? l1 : l2, hi = l1 < l2 ? l2 : l1;
        const double t = S.small[k][on ? hi * (hi + 1) / 2 + lo : 0];
        v += on ? t : 0.0;
      }
    }
    S.Hc[idx] = v;
    a += da;
    c += dc;
    if (c >= W) { c -= W; ++a; }
  }
  for (int i = tid; i < n; i += nthreads) {
    const int cp = LM_T(S).colcp[i];
    double v = 0.0;
#pragma unroll 4
    for (int k = 0; k < nb; ++k) {
      const int d = LM_T(S).desc[k];
      const int l = local_col_tsdf(cp, d & 255, (d >> 8) & 255);
      const bool on = (d >> 16) && l >= 0;
      const double t = S.loc[k][on ? 78 + l : 0];
      v += on ? t : 0.0;
    }
#pragma unroll 4
    for (int k = 0; k < ns; ++k) {
      const int d = LM_T(S).desc[kMaxBlocks + k];
      const int l = local_col_small(cp, d & 255, (d >> 8) & 255);
      const bool on = (d >> 16) && l >= 0;
      const double t = S.small[k][on ? kSmallTri + l : 0];
      v += on ? t : 0.0;
    }
    h.gc[i] = v;
  }
  if (tid == 0) {
    double cost = 0.0;
    for (int k = 0; k < nb; ++k)
      if (h.blocks[k].active) cost += S.loc[k][90];
    for (int k = 0; k < ns; ++k)
      if (h.small[k].active) cost += S.small[k][kSmallTri + 18];
    h.cand_cost = 0.5 * cost;
  }
  __syncthreads();
}

// The same from the static gather lists (LmState::gather_*): every contribution is an independent load from
// device memory (the local systems were written by the previous launch), summed in list order = block order.
__device__ __forceinline__ void assemble_gathered(LmShared& S, const LmState* G, const double* loc_sums,
                                                  const SmallOut* small_out) {
  LmHead& h = S.h;
  const int n = h.ncols, W = h.bw + 1;
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const double* small = reinterpret_cast<const double*>(small_out);
  auto fetch = [&](unsigned short at) {
    const bool on = at != 0xFFFFu;
    const bool sm = at >= kGatherSmallBase;
    const double* src = sm ? small : loc_sums;
    const unsigned o = on ? (sm ? at - kGatherSmallBase : at) : 0u;
    const double v = src[o];
    return on ? v : 0.0;
  };
  for (int idx = tid; idx < n * W; idx += nthreads) {
    typedef unsigned short us8 __attribute__((ext_vector_type(kGatherMax)));
    const us8 at = *reinterpret_cast<const us8*>(&G->gather_h[idx][0]);
    double t[kGatherMax];
#pragma unroll
    for (int k = 0; k < kGatherMax; ++k) t[k] = fetch(at[k]);
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < kGatherMax; ++k) v += t[k];
    S.Hc[idx] = v;
  }
  for (int i = tid; i < n; i += nthreads) {
    typedef unsigned short us8 __attribute__((ext_vector_type(kGatherMax)));
    const us8 at = *reinterpret_cast<const us8*>(&G->gather_g[i][0]);
    double t[kGatherMax];
#pragma unroll
    for (int k = 0; k < kGatherMax; ++k) t[k] = fetch(at[k]);
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < kGatherMax; ++k) v += t[k];
    h.gc[i] = v;
  }
  // cost: the blocks' r^T r, added in block order by thread 0 (loads by one thread each)
  const int nb = h.num_blocks, ns = h.num_small;
  for (int k = tid; k < nb; k += nthreads) S.A[k] = h.blocks[k].active ? loc_sums[k * kLoc + 90] : 0.0;  // (A is free until the next solve)
  for (int k = tid; k < ns; k += nthreads)
    S.A[kMaxBlocks + k] = h.small[k].active ? small[k * kSmallLoc + kSmallTri + 18] : 0.0;
  __syncthreads();
  if (tid == 0) {
    double cost = 0.0;
    for (int k = 0; k < nb; ++k) cost += S.A[k];
    for (int k = 0; k < ns; ++k) cost += S.A[kMaxBlocks + k];
    h.cand_cost = 0.5 * cost;
  }
  __syncthreads();
}

#ifndef HG_BIG
// What a step's thread asks of device memory before it knows the problem's shape (round 5): its words of the head, its
// entries of H, the gather lists of its entries of the candidate's normal equations and its share of the blocks'
// local systems (written by the residual launch in front; how many blocks there are comes as a kernel argument) --
// ONE round trip at kernel entry. The local systems are staged in LDS (coalesced 16-byte loads) and the entries
// gathered from there: gathered straight from device memory, 20k scattered 8-byte loads went through the CU's one
// address unit at ~4 lanes a cycle (8.3k cycles behind the head: in-kernel stamps); the step used to walk through
// five dependent round trips (head, H, lists, values, cost terms; head + H + assembly 12.9k cycles).
// Covers the first kPre * blockDim entries (2048: the LDS-resident build holds 3240; the rest take the plain loops).
constexpr int kPre = 4;
constexpr int kStageLoc = (kMaxBlocks * kLoc / 2 + kLmBlock - 1) / kLmBlock;        // 16-byte loads per thread: every block
constexpr int kStageSmall = (kMaxSmall * kSmallLoc / 2 + kLmBlock - 1) / kLmBlock;
static_assert(kLoc % 2 == 0 && kSmallLoc % 2 == 0, "local systems are staged in 16-byte pieces");
typedef unsigned short hg_us8 __attribute__((ext_vector_type(kGatherMax)));
typedef double hg_d2 __attribute__((ext_vector_type(2)));
struct LmPrefetch {
  unsigned long long head[(sizeof(LmHead) / 8 + kLmBlock - 1) / kLmBlock];
  double H[kPre];
  hg_us8 at_h[kPre];
  hg_us8 at_g;
  hg_d2 loc[kStageLoc], small[kStageSmall];
};
__device__ __forceinline__ void lm_prefetch(LmPrefetch& P, const LmState* G, const double* loc_sums, const SmallOut* small_out,
                                            unsigned nb, unsigned ns, unsigned nw) {
  const unsigned tid = threadIdx.x;
  constexpr unsigned kWords = sizeof(LmHead) / 8;
  const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&G->h);
#pragma unroll
  for (unsigned u = 0; u < sizeof(P.head) / 8; ++u) {
    const unsigned i = tid + u * kLmBlock;
    P.head[u] = src[i < kWords ? i : 0u];
  }
#pragma unroll
  for (int u = 0; u < kPre; ++u) {
    const unsigned idx = tid + u * kLmBlock;  // < kPre * kLmBlock <= kHCap: always inside the arrays
    if (u * kLmBlock < nw) {  // (uniform: a round of entries the band does not have is not asked for)
      P.H[u] = G->H[idx];
      P.at_h[u] = *reinterpret_cast<const hg_us8*>(&G->gather_h[idx][0]);
    } else {
      P.H[u] = 0.0;
      P.at_h[u] = static_cast<unsigned short>(0xFFFFu);
    }
  }
  P.at_g = *reinterpret_cast<const hg_us8*>(&G->gather_g[tid < kMaxCols ? tid : 0][0]);
  const hg_d2* l2 = reinterpret_cast<const hg_d2*>(loc_sums);
  const hg_d2* s2 = reinterpret_cast<const hg_d2*>(small_out);
  const unsigned nl = nb * (kLoc / 2), nsm = ns * (kSmallLoc / 2);
#pragma unroll
  for (int u = 0; u < kStageLoc; ++u) {
    const unsigned i = tid + u * kLmBlock;
    if (i < nl) P.loc[u] = l2[i];
  }
#pragma unroll
  for (int u = 0; u < kStageSmall; ++u) {
    const unsigned i = tid + u * kLmBlock;
    if (i < nsm) P.small[u] = s2[i];
  }
}
static_assert(kPre * kLmBlock <= kHCap, "prefetched entries lie inside the band storage");

// the staged local systems to LDS (before the barrier behind the head)
__device__ __forceinline__ void lm_stage(LmShared& S, const LmPrefetch& P, unsigned nb, unsigned ns) {
  const unsigned tid = threadIdx.x;
  hg_d2* l2 = reinterpret_cast<hg_d2*>(&S.loc[0][0]);
  hg_d2* s2 = reinterpret_cast<hg_d2*>(&S.small[0][0]);
  const unsigned nl = nb * (kLoc / 2), nsm = ns * (kSmallLoc / 2);
#pragma unroll
  for (int u = 0; u < kStageLoc; ++u) {
    const unsigned i = tid + u * kLmBlock;
    if (i < nl) l2[i] = P.loc[u];
  }
#pragma unroll
  for (int u = 0; u < kStageSmall; ++u) {
    const unsigned i = tid + u * kLmBlock;
    if (i < nsm) s2[i] = P.small[u];
  }
}

// assemble_gathered with the lists in registers and the local systems in LDS (same order of the sums per entry)
__device__ __forceinline__ void assemble_prefetched(LmShared& S, const LmPrefetch& P, const LmState* G) {
  LmHead& h = S.h;
  const int n = h.ncols, W = h.bw + 1, nW = n * W;
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const double* locf = &S.loc[0][0];
  const double* smallf = &S.small[0][0];
  auto fetch = [&](unsigned short at, bool live) {
    const bool on = live && at != 0xFFFFu;
    const bool sm = at >= kGatherSmallBase;
    const double* src = sm ? smallf : locf;
    const unsigned o = on ? (sm ? at - kGatherSmallBase : at) : 0u;
    const double v = src[o];
    return on ? v : 0.0;
  };
  // contributions k0 .. k1 - 1 of every entry of this thread, all reads in flight; the lists are packed from the front,
  // so the later groups are skipped when no thread of the wavefront has that many
  double v[kPre + 1];
#pragma unroll
  for (int u = 0; u <= kPre; ++u) v[u] = 0.0;
  bool live[kPre + 1];
#pragma unroll
  for (int u = 0; u < kPre; ++u) live[u] = tid + u * kLmBlock < nW;
  live[kPre] = tid < n;
  auto group = [&](int k0, int k1) {
    double t[kPre + 1][kGatherMax];
#pragma unroll
    for (int u = 0; u <= kPre; ++u) {
      if (u < kPre && u * kLmBlock >= nW) continue;  // (uniform: a round of entries the band does not have)
#pragma unroll
      for (int k = k0; k < k1; ++k) t[u][k] = fetch(u < kPre ? P.at_h[u][k] : P.at_g[k], live[u]);
    }
#pragma unroll
    for (int u = 0; u <= kPre; ++u) {
      if (u < kPre && u * kLmBlock >= nW) continue;
#pragma unroll
      for (int k = k0; k < k1; ++k) v[u] += t[u][k];
    }
  };
  auto any_at = [&](int k) {
    bool a = live[kPre] && P.at_g[k] != 0xFFFFu;
#pragma unroll
    for (int u = 0; u < kPre; ++u) a = a || (live[u] && P.at_h[u][k] != 0xFFFFu);
    return __ballot(a) != 0ull;
  };
  constexpr int kFirst = kGatherMax < 4 ? kGatherMax : 4;
  group(0, kFirst);
#pragma unroll
  for (int k0 = kFirst; k0 < kGatherMax; k0 += 2) {
    if (!any_at(k0)) break;  // (wave-uniform)
    group(k0, k0 + 2 < kGatherMax ? k0 + 2 : kGatherMax);
  }
  // the blocks' r^T r: one term per lane of wavefront 0 (lane-strided beyond 64 blocks), added by a butterfly
  const int nb = h.num_blocks, ns = h.num_small;
  if (tid < kWave) {
    double term = 0.0;
    for (int k = tid; k < nb + ns; k += kWave) {
      const bool tsdf = k < nb;
      const bool on = tsdf ? h.blocks[k].active != 0 : h.small[k - nb].active != 0;
      const double c = tsdf ? locf[k * kLoc + 90] : smallf[(k - nb) * kSmallLoc + kSmallTri + 18];
      term += on ? c : 0.0;
    }
    const double cost = wave_sum(term);
    if (tid == 0) h.cand_cost = 0.5 * cost;
  }
#pragma unroll
  for (int u = 0; u < kPre; ++u)
    if (live[u]) S.Hc[tid + u * kLmBlock] = v[u];
  if (live[kPre]) h.gc[tid] = v[kPre];
  // (entries beyond the prefetched ones: lists from device memory, contributions from LDS)
  for (int idx = tid + kPre * kLmBlock; idx < nW; idx += nthreads) {
    const hg_us8 at = *reinterpret_cast<const hg_us8*>(&G->gather_h[idx][0]);
    double w = 0.0;
#pragma unroll
    for (int k = 0; k < kGatherMax; ++k) w += fetch(at[k], true);
    S.Hc[idx] = w;
  }
  for (int i = tid + kLmBlock; i < n; i += nthreads) {
    const hg_us8 at = *reinterpret_cast<const hg_us8*>(&G->gather_g[i][0]);
    double w = 0.0;
#pragma unroll
    for (int k = 0; k < kGatherMax; ++k) w += fetch(at[k], true);
    h.gc[i] = w;
  }
  __syncthreads();
}
#endif

// Transforms of every block at the candidate (thread b: block b), written straight to device memory
// for the next residual launch. Only the structural non-zeros of M are filled by prepare_block.
__device__ __forceinline__ void prepare_all(const LmHead& h, BlockXform* xf) {
  if (static_cast<int>(threadIdx.x) < h.num_blocks) {
    BlockXform* dst = xf + threadIdx.x;
    double* d = reinterpret_cast<double*>(dst);
    for (int i = 0; i < static_cast<int>(sizeof(BlockXform) / sizeof(double)); ++i) d[i] = 0.0;
    prepare_block(h.blocks[threadIdx.x], h.cand, dst);
  }
}

// Builds the index tables of the problem in LDS (head already there) and stores them for the LM steps.
// Workgroup-wide.
__device__ __forceinline__ void build_tables(LmShared& S, LmState* G) {
  const LmHead& h = S.h;
  const int tid = threadIdx.x, nthreads = blockDim.x;
  if (tid == 0) LM_T(S).pair_overflow = 0;
  __syncthreads();
  if (tid < h.num_poses) {
    const int p = tid;
    if (!h.constant[p])
      for (int k = 0; k < 6; ++k) LM_T(S).colcp[h.col[p] + k] = p * 16 + k;
    if (h.vfree[p])
      for (int k = 0; k < 3; ++k) LM_T(S).colcp[h.vcol[p] + k] = p * 16 + 6 + k;
  }
  for (int b = tid; b < h.num_blocks; b += nthreads) {
    const BlockInfo& bi = h.blocks[b];
    LM_T(S).desc[b] = (bi.active ? 1 << 16 : 0) | ((bi.pose_b & 255) << 8) | (bi.pose_a & 255);
  }
  for (int b = tid; b < h.num_small; b += nthreads) {
    const SmallBlockDev& sb = h.small[b];
    LM_T(S).desc[kMaxBlocks + b] = (sb.active ? 1 << 16 : 0) | ((sb.b & 255) << 8) | (sb.a & 255);
  }
  for (int pq = tid; pq < kMaxPoses * kMaxPoses; pq += nthreads) {
    // (p, q), q <= p: the blocks covering both control points, in block order
    const int p = pq / kMaxPoses, q = pq % kMaxPoses;
    if (q <= p && p < h.num_poses) {
      int cnt = 0;
      bool over = false;
      for (int k = 0; k < h.num_blocks; ++k) {
        const BlockInfo& bi = h.blocks[k];
        const bool hp = bi.pose_a == p || bi.pose_b == p, hq = bi.pose_a == q || bi.pose_b == q;
        if (!bi.active || !hp || !hq) continue;
        if (cnt < kPairMax)
          LM_T(S).pair_list[p][q][cnt] = k | ((bi.pose_a == p ? 0 : 6) << 12) | ((bi.pose_a == q ? 0 : 6) << 20);
        else over = true;
        ++cnt;
      }
      for (int k = 0; k < h.num_small; ++k) {
        const SmallBlockDev& sb = h.small[k];
        const bool hp = sb.a == p || sb.b == p, hq = sb.a == q || sb.b == q;
        if (!sb.active || !hp || !hq) continue;
        if (cnt < kPairMax)
          LM_T(S).pair_list[p][q][cnt] = k | (1 << 8) | ((sb.a == p ? 0 : 9) << 12) | ((sb.a == q ? 0 : 9) << 20);
        else over = true;
        ++cnt;
      }
      LM_T(S).pair_count[p][q] = cnt < kPairMax ? cnt : kPairMax;
      if (over) LM_T(S).pair_overflow = 1;
    }
  }
  __syncthreads();
#ifndef HG_BIG
  {
    const int* src = reinterpret_cast<const int*>(&LM_T(S));
    int* td = reinterpret_cast<int*>(&G->T);
    for (int i = tid; i < static_cast<int>(sizeof(LmTables) / sizeof(int)); i += nthreads) td[i] = src[i];
  }
#endif
  if (LM_T(S).pair_overflow) return;  // uniform: the step scans the blocks per entry instead
  // gather lists from the pair lists (same contributions, same order as assemble's by-pairs loop)
  const int n = h.ncols, W = h.bw + 1;
  for (int idx = tid; idx < n * W; idx += nthreads) {
    const int a = idx / W, b = a - (W - 1) + (idx - a * W);
    int k_out = 0;
    if (b >= 0) {
      const int cpa = LM_T(S).colcp[a], cpb = LM_T(S).colcp[b];
      const int p = cpa >> 4, q = cpb >> 4, sa = cpa & 15, sb = cpb & 15;
      const bool sw = p < q;
      const int hi_p = sw ? q : p, lo_p = sw ? p : q;
      const int cnt = LM_T(S).pair_count[hi_p][lo_p];
      for (int k = 0; k < cnt; ++k) {
        const int e = LM_T(S).pair_list[hi_p][lo_p][k];
        const int blk = e & 255, small = (e >> 8) & 1;
        const int o_hi = (e >> 12) & 255, o_lo = (e >> 20) & 255;
        const int l1 = sa + (sw ? o_lo : o_hi), l2 = sb + (sw ? o_hi : o_lo);
        const int lo = l1 < l2 ? l1 : l2, hi = l1 < l2 ? l2 : l1;
        if (!(small || (sa < 6 && sb < 6))) continue;
        const unsigned at = small ? kGatherSmallBase + blk * kSmallLoc + hi * (hi + 1) / 2 + lo
                                  : blk * kLoc + lo * 12 - (lo * (lo - 1)) / 2 + (hi - lo);
        G->gather_h[idx][k_out++] = static_cast<unsigned short>(at);
      }
    }
    for (; k_out < kGatherMax; ++k_out) G->gather_h[idx][k_out] = 0xFFFFu;
  }
  for (int i = tid; i < n; i += nthreads) {
    const int cp = LM_T(S).colcp[i];
    const int p = cp >> 4, sl = cp & 15;
    const int cnt = LM_T(S).pair_count[p][p];
    int k_out = 0;
    for (int k = 0; k < cnt; ++k) {
      const int e = LM_T(S).pair_list[p][p][k];
      const int blk = e & 255, small = (e >> 8) & 1, off = (e >> 12) & 255;
      if (!small && sl >= 6) continue;
      const unsigned at = small ? kGatherSmallBase + blk * kSmallLoc + kSmallTri + off + sl : blk * kLoc + 78 + off + sl;
      G->gather_g[i][k_out++] = static_cast<unsigned short>(at);
    }
    for (; k_out < kGatherMax; ++k_out) G->gather_g[i][k_out] = 0xFFFFu;
  }
}

// One LM iteration by the calling workgroup: loads the solver head, the blocks' local systems
// (written by the tails of k_window_residuals and by the odometry / IMU wavefronts) and advances the
// state machine.
// The end of a step: transforms of the candidate for the next residual launch, the head (and H, if it changed) back to
// device memory, the mailbox when the solve has terminated. Workgroup-wide.
__device__ __forceinline__ void lm_finish(LmShared& S, LmState* G, BlockXform* xf) {
  const int tid = threadIdx.x, nthreads = blockDim.x;
  LmHead& h = S.h;
  __syncthreads();
  HG_STAMP(S, 6);
#ifdef HG_LM_STAMPS
  if (threadIdx.x == 0 && S.h.stamps[15]) S.h.stamps[12] = __builtin_amdgcn_s_memrealtime();
#endif
  if (!h.done) prepare_all(h, xf);
  HG_STAMP(S, 7);
#ifdef HG_LM_STAMPS
  __syncthreads();  // (the stamp is in the head the threads store below)
#endif
  // store the head and (if it changed) H
  {
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&S.h);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(&G->h);
    for (unsigned i = tid; i < sizeof(LmHead) / 8; i += nthreads) dst[i] = src[i];
  }
#ifndef HG_BIG
  if (S.h_changed) {
    const int nW = h.ncols * (h.bw + 1);
    for (int i = tid; i < nW; i += nthreads) G->H[i] = S.H[i];
  }
#endif
  if (h.done && h.box) {
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&S.h);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(&h.box->h);
    for (unsigned i = tid; i < sizeof(LmHead) / 8; i += nthreads) dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
      *reinterpret_cast<volatile unsigned long long*>(&h.box->flag) = h.seq;
      __threadfence_system();
    }
  }
}

// A step up to its first factorisation (or all of MODE_PREPARE / MODE_ASSEMBLE). True: the scaled system stands in
// S.A / S.rhs -- attempt_solve, then lm_back. A function of its own (see attempt_begin); its state lives in g_lm_shared.
// (The pointers are typed as device memory: a function cannot infer that from its callers as a kernel does, and
// generic pointers made every access a flat operation, which counts against the LDS counter as well.)
#define HG_GLOBAL __attribute__((address_space(1)))
__device__ __attribute__((noinline)) bool lm_front(HG_GLOBAL LmState* G1, HG_GLOBAL BlockXform* xf1, const HG_GLOBAL double* loc1,
                        const HG_GLOBAL SmallOut* small1, int mode, const HG_GLOBAL PinBox* up1,
                        unsigned up_words, unsigned stage /* lm_stage_counts(), or 0: unknown */) {
  LmState* G = (LmState*)G1;
  BlockXform* xf = (BlockXform*)xf1;
  const double* loc_sums = (const double*)loc1;
  const SmallOut* small_out = (const SmallOut*)small1;
  const PinBox* host_up = (const PinBox*)up1;
  LmShared& S = g_lm_shared;
  const int tid = threadIdx.x, nthreads = blockDim.x;
#ifdef HG_LM_STAMPS
  const long long t_entry = __builtin_amdgcn_s_memtime();
#endif
#ifdef HG_BIG
  // the big build works on the matrices, local systems and tables where they are (device memory)
  if (threadIdx.x == 0) {
    S.H = G->H;
    S.Hc = G->Hc;
    S.A = G->A;
    S.loc = reinterpret_cast<double (*)[kLoc]>(const_cast<double*>(loc_sums));
    S.small = reinterpret_cast<double (*)[kSmallLoc]>(const_cast<SmallOut*>(small_out));
    S.Tp = &G->T;
  }
  __syncthreads();
#endif
#ifndef HG_BIG
  LmPrefetch pre;
  const unsigned st_nb = stage & 0xFFu, st_ns = (stage >> 8) & 0xFFu, st_nw = stage >> 16;
  const bool prefetched = mode != MODE_PREPARE && blockDim.x == kLmBlock && stage != 0u && st_nb <= kMaxBlocks && st_ns <= kMaxSmall;
  if (prefetched) lm_prefetch(pre, G, loc_sums, small_out, st_nb, st_ns, st_nw);
#endif
  if (mode == MODE_PREPARE && host_up) {
    // zero-copy upload: scatter the non-zero words the host left in the mailbox
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(&S.h);
    constexpr int kUp = static_cast<int>((sizeof(LmHead) / 8 + kLmBlock - 1) / kLmBlock);  // words per thread
    unsigned long long val[kUp];
    unsigned idx[kUp];
#pragma unroll
    for (int u = 0; u < kUp; ++u) {  // all host reads in flight together
      const unsigned w = threadIdx.x + u * blockDim.x;
      idx[u] = w < up_words ? host_up->up_idx[w] : 0xFFFFFFFFu;
      val[u] = w < up_words ? host_up->up_val[w] : 0ull;
    }
    for (unsigned i = threadIdx.x; i < sizeof(LmHead) / 8; i += blockDim.x) dst[i] = 0ull;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kUp; ++u)
      if (idx[u] < sizeof(LmHead) / 8) dst[idx[u]] = val[u];
  } else {
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(&S.h);
#ifndef HG_BIG
    if (prefetched) {
#pragma unroll
      for (unsigned u = 0; u < sizeof(pre.head) / 8; ++u) {
        const unsigned i = threadIdx.x + u * kLmBlock;
        if (i < sizeof(LmHead) / 8) dst[i] = pre.head[u];
      }
      lm_stage(S, pre, st_nb, st_ns);
    } else
#endif
    {
      const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&G->h);
      for (unsigned i = threadIdx.x; i < sizeof(LmHead) / 8; i += blockDim.x) dst[i] = src[i];
    }
  }
  __syncthreads();
  LmHead& h = S.h;
  if (mode == MODE_STEP && h.done) return false;  // (uniform; the kernels look before the call when they cannot prefetch)
  const int n = h.ncols, nW = n * (h.bw + 1);
#ifdef HG_LM_STAMPS
  if (threadIdx.x == 0) {
    h.stamps[15] = (h.iteration == 3) ? 1 : 0;
    if (h.stamps[15]) h.stamps[8] = t_entry;
  }
  __syncthreads();
#endif
  if (mode == MODE_PREPARE) {
    if (host_up) {
      const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&S.h);
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(&G->h);
      for (unsigned i = threadIdx.x; i < sizeof(LmHead) / 8; i += blockDim.x) dst[i] = src[i];
    }
    prepare_all(h, xf);
    build_tables(S, G);
    return false;
  }
  HG_STAMP(S, 0);
#ifdef HG_LM_STAMPS
  if (threadIdx.x == 0 && S.h.stamps[15]) S.h.stamps[11] = __builtin_amdgcn_s_memrealtime();
#endif
  // H at x; the assembly gathers from device memory through the static lists, or -- more than kPairMax
  // blocks on a pair of control points -- stages the local systems and the tables and searches per entry
  const bool gathered = G->T.pair_overflow == 0;  // uniform
#ifndef HG_BIG
  if (h.phase != PHASE_INIT) {
    if (prefetched) {
#pragma unroll
      for (int u = 0; u < kPre; ++u)
        if (tid + u * kLmBlock < nW) S.H[tid + u * kLmBlock] = pre.H[u];
      for (int i = tid + kPre * kLmBlock; i < nW; i += nthreads) S.H[i] = G->H[i];
    } else {
      for (int i = tid; i < nW; i += nthreads) S.H[i] = G->H[i];
    }
  }
#endif
  if (gathered) {
    HG_STAMP(S, 1);
#ifndef HG_BIG
    if (prefetched) assemble_prefetched(S, pre, G);
    else
#endif
      assemble_gathered(S, G, loc_sums, small_out);
  } else {
#ifndef HG_BIG
    double* dst = &S.loc[0][0];
    for (int i = tid; i < h.num_blocks * kLoc; i += nthreads) dst[i] = loc_sums[i];
    {
      // (inactive blocks keep whatever an earlier solve left there: the descriptors mask them out)
      const double* src = reinterpret_cast<const double*>(small_out);
      double* sd = &S.small[0][0];
      for (int i = tid; i < h.num_small * kSmallLoc; i += nthreads) sd[i] = src[i];
    }
    {
      const int* src = reinterpret_cast<const int*>(&G->T);
      int* td = reinterpret_cast<int*>(&LM_T(S));
      for (int i = tid; i < static_cast<int>(sizeof(LmTables) / sizeof(int)); i += nthreads) td[i] = src[i];
    }
    __syncthreads();
#endif
    HG_STAMP(S, 1);
    assemble(S);
  }
  HG_STAMP(S, 2);
  if (mode == MODE_ASSEMBLE) {
#ifndef HG_BIG
    for (int i = tid; i < nW; i += nthreads) G->Hc[i] = S.Hc[i];
#endif
    for (int i = tid; i < n; i += nthreads) G->h.gc[i] = h.gc[i];
    if (tid == 0) G->h.cand_cost = h.cand_cost;
    return false;
  }
  bool h_changed = false;
  bool solve = false;
  if (h.phase == PHASE_INIT) {
    // IterationZero: EvaluateGradientAndJacobian at x (= cand)
    for (int i = tid; i < nW; i += nthreads) S.H[i] = S.Hc[i];
    for (int i = tid; i < n; i += nthreads) h.g[i] = h.gc[i];
    __syncthreads();
    for (int k = tid; k < n; k += nthreads)
      h.scale[k] = h.opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(S.H[band_index(k, k, h.bw + 1)])) : 1.0;
    const double gmn = gradient_max_norm(S);  // barriers inside
    if (tid == 0) {
      ++h.num_cost_evals;
      ++h.num_jac_evals;
      h.x_cost = h.cand_cost;
      h.initial_cost = h.cand_cost;
      h.gradient_max_norm = gmn;
      h.step_is_successful = 1;
      h.num_iterations = 1;
      h.phase = PHASE_CANDIDATE;
    }
    h_changed = true;
    solve = attempt_begin(S);
  } else {
    // candidate evaluated: ParameterToleranceReached / FunctionToleranceReached / IsStepSuccessful
    // (every thread evaluates the same scalars from the head in LDS)
    if (tid < h.num_poses) {  // thread p: the squared norms of control point p's free parameters
      const int p = tid;
      const int k0 = h.constant[p] ? 7 : 0, k1 = h.vfree[p] ? kState : 7;
      double dn = 0.0, pn = 0.0;
#pragma unroll
      for (int k = 0; k < kState; ++k) {
        const bool on = k >= k0 && k < k1;
        const double xv = h.x[p][k], d = xv - h.cand[p][k];
        dn += on ? d * d : 0.0;
        pn += on ? xv * xv : 0.0;
      }
      S.red[p] = dn;
      S.A[p] = pn;  // (A is free until the next solve)
    }
    // the two sums over the control points, in control-point order as before, by lane 0 of wavefront 0 (whose lanes
    // are the threads above: kMaxPoses <= 64) instead of by every thread behind a barrier of its own
    if (tid == 0) {
      wave_sync();
      double sn0 = 0.0, xn0 = 0.0;
      for (int p = 0; p < h.num_poses; ++p) {
        sn0 += S.red[p];
        xn0 += S.A[p];
      }
      S.red[kRedScalar + 2] = sn0;
      S.A[kMaxPoses] = xn0;
    }
    static_assert(kMaxPoses <= kWave, "the per-pose threads of a step sit in one wavefront");
    __syncthreads();
    const double sn = sqrt(S.red[kRedScalar + 2]);
    const double xn = sqrt(S.A[kMaxPoses]);
    const double cost_change = h.x_cost - h.cand_cost;
    const bool ptol = sn <= h.opt.parameter_tolerance * (xn + h.opt.parameter_tolerance);
    const bool ftol = fabs(cost_change) <= h.opt.function_tolerance * h.x_cost;
    const double relative_decrease = cost_change / h.model_cost_change;
    const bool accept = relative_decrease > h.opt.min_relative_decrease;
    __syncthreads();
    HG_STAMP(S, 3);
    if (tid == 0) ++h.num_cost_evals;
    if (ptol) {
      if (tid == 0) finish(h, 0, 2);
    } else if (ftol) {
      if (tid == 0) finish(h, 0, 3);
    } else {
      if (accept) {
        // HandleSuccessfulStep: the candidate's normal equations become x's
        for (int i = tid; i < nW; i += nthreads) S.H[i] = S.Hc[i];
        for (int i = tid; i < n; i += nthreads) h.g[i] = h.gc[i];
        if (tid < h.num_poses)
          for (int k = 0; k < kState; ++k) h.x[tid][k] = h.cand[tid][k];
        h_changed = true;
        __syncthreads();
        const double gmn = gradient_max_norm(S);
        if (tid == 0) {
          h.x_cost = h.cand_cost;
          ++h.num_jac_evals;
          h.gradient_max_norm = gmn;
          h.step_is_successful = 1;
          {
            const double t = 2.0 * relative_decrease - 1.0;
            h.radius = h.radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
          }
          h.radius = fmin(h.opt.max_trust_region_radius, h.radius);
          h.decrease_factor = 2.0;
          h.reuse_diagonal = 0;
        }
      } else {
        if (tid == 0) {
          h.radius = h.radius / h.decrease_factor;
          h.decrease_factor *= 2.0;
          h.reuse_diagonal = 1;
        }
      }
      solve = attempt_begin(S);
    }
  }
  if (tid == 0) {  // (uniform; read behind barriers)
    S.h_changed = h_changed ? 1 : 0;
    S.step_G = G;
    S.step_xf = xf;
  }
  if (!solve) lm_finish(S, G, xf);
  return solve;
}

// A step behind a factorisation. True: the step was invalid and the system has been rebuilt with a smaller radius --
// attempt_solve again.
__device__ __attribute__((noinline)) bool lm_back() {
  LmShared& S = g_lm_shared;
  LmState* G = (LmState*)(HG_GLOBAL LmState*)S.step_G;
  BlockXform* xf = (BlockXform*)(HG_GLOBAL BlockXform*)S.step_xf;
  bool again = attempt_end(S);
  if (again) again = attempt_begin(S);
  if (!again) lm_finish(S, G, xf);
  return again;
}

// Invalid steps (the rare path: the first factorisation of a step is in the kernel body, straight-line -- in a loop
// there its addresses and lane indices were hoisted in front of the loop and spilled): factorise again until a step is
// valid or the solve gives up.
__device__ __attribute__((noinline)) void lm_retry() {
  bool again = true;
  while (again) {  // (uniform)
    attempt_solve(g_lm_shared);
    again = lm_back();
  }
}

// MODE_PREPARE / MODE_ASSEMBLE / one LM step, for the kernels
__device__ __forceinline__ void lm_step(LmState* G, BlockXform* xf, const double* loc_sums, const SmallOut* small_out, int mode,
                                        const PinBox* host_up = nullptr, unsigned up_words = 0, unsigned stage = 0) {
  bool solve = lm_front((HG_GLOBAL LmState*)G, (HG_GLOBAL BlockXform*)xf, (const HG_GLOBAL double*)loc_sums,
                        (const HG_GLOBAL SmallOut*)small_out, mode, (const HG_GLOBAL PinBox*)host_up, up_words, stage);
  if (solve) {  // (uniform)
    attempt_solve(g_lm_shared);
    if (lm_back()) lm_retry();
  }
}

// ------------------------------------------------------------------------------------------
// Single free pose, one per-scan block, no velocity / odometry / IMU blocks (CeresScanMatcher3D and
// the per-scan registration step): the same state machine as lm_step with n = 6, but every lane of
// wavefront 0 keeps the whole solver state in registers and advances it redundantly — no LDS
// round trips or barriers after the partial reduction. The general path's LDS/loop overhead is
// ~40k cycles per iteration; this is the launch-latency-bound inner loop of the registration.
// `scratch` needs (blockDim / 36) * 36 + 36 doubles of LDS.
// ------------------------------------------------------------------------------------------
#ifdef HG_EVAL_STAMPS
__device__ unsigned long long g_tail_stamps[8];
#define TAIL_STAMP(i) do { if (threadIdx.x == 0 && g_tail_stamps[7] == 3) g_tail_stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TAIL_STAMP(i) do {} while (0)
#endif
__device__ inline int b6(int i, int j) { return i * 6 + 5 - (i - j); }  // band_index(i, j, W = 6), j <= i

template <bool FIRST_T = false, bool PERSIST = false /* called from the persistent solve */>
__device__ __forceinline__ void lm_step_single(double* scratch, LmState* G, BlockXform* xf, const double* partials,
                               unsigned num_wg, const PinBox* host_up = nullptr, unsigned up_words = 0,
                               unsigned epoch = 0 /* != 0: `partials` holds tagged granules of this epoch */,
                               double* persist_out = nullptr /* LDS, 8 doubles: the next candidate and the done flag (persistent solve) */,
                               int* timeout_flag = nullptr /* LDS: set when a granule never arrived; the step then ends the solve as failed */,
                               unsigned long long* persist_bcast = nullptr /* persistent solve: kBcastReplicas x 8 granule pairs, the hand-back to the other workgroups */,
                               bool first_rt = false /* PERSIST: this is the solve's first step (run time) */) {
  // (Round 5: multiply-adds of this function are fused -- the file is compiled -ffp-contract=off for the voxel lookups,
  // whose discrete decisions need the reference's roundings; nothing in the LM step takes one, and the tail is a chain
  // of dependent fp64 operations: 14.7 -> 14.0 us per launch together with the right-looking factorisation below.)
#pragma clang fp contract(fast)
  // The first step of a solve differs in its upload. A launch per evaluation has a kernel of its own for it (FIRST_T);
  // the persistent solve decides at run time and runs ONE copy of the step for all its evaluations: two inlined copies
  // meant that the second evaluation found none of its instructions in the cache the first one had just filled.
  constexpr bool kMaybeFirst = FIRST_T || PERSIST;
  const bool FIRST = PERSIST ? first_rt : FIRST_T;
  int t_opaque = threadIdx.x;
  if (PERSIST) asm volatile("" : "+v"(t_opaque));  // (the persistent solve calls the step inside its evaluation loop: addresses formed
                                                   // from the thread index were hoisted in front of the loop and spilled there)
  const int t = t_opaque;
  LmHead& gh = G->h;
  TAIL_STAMP(0);
  // the solver head and H are staged through LDS by all threads while the partial loads are in
  // flight (one memory round trip for both); layout: [stripes*36 | 36 sums | head | 36 H entries]
  const int stripes0 = static_cast<int>(blockDim.x) / kAcc;
  LmHead& sh = *reinterpret_cast<LmHead*>(scratch + (stripes0 + 1) * kAcc);
  double* sH = reinterpret_cast<double*>(&sh + 1);
  constexpr int kUp = static_cast<int>((sizeof(LmHead) / 8 + kEvalThreads - 1) / kEvalThreads);  // mailbox words per thread: 3
  unsigned long long up_val[kMaybeFirst ? kUp : 1];
  unsigned up_idx[kMaybeFirst ? kUp : 1];
  if (FIRST) {
    // the head arrives as its non-zero words in the host's mailbox: reads in flight now, scattered
    // into the zeroed LDS head behind the first barrier below
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(&sh);
#pragma unroll
    for (int u = 0; u < kUp; ++u) {
      const unsigned w = t + u * blockDim.x;
      up_idx[u] = w < up_words ? host_up->up_idx[w] : 0xFFFFFFFFu;
      up_val[u] = w < up_words ? host_up->up_val[w] : 0ull;
    }
    for (unsigned i = t; i < sizeof(LmHead) / 8; i += blockDim.x) dst[i] = 0ull;
    if (t < 36) sH[t] = 0.0;
  } else {
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&gh);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(&sh);
    for (unsigned i = t; i < sizeof(LmHead) / 8; i += blockDim.x) dst[i] = src[i];
    if (t < 36) sH[t] = G->H[t];
  }
  // --- partial sums: thread (stripe j, column k), <= 16 loads in flight, fixed-order stripe sum ---
  {
    const int stripes = static_cast<int>(blockDim.x) / kAcc;
    const int j = t / kAcc, k = t % kAcc;
    if (j < stripes) {
      double acc = 0.0;
      const double* p = partials + k;
      if (epoch != 0u) {  // (uniform) tagged granules: re-read what has not arrived yet
        const unsigned long long* gp = reinterpret_cast<const unsigned long long*>(partials) + 2 * k;
        for (unsigned w = j; w < num_wg; w += kSweep * stripes) {
          hg_u64x2 gr[kSweep];
          const unsigned long long* ga[kSweep];
          const unsigned long long none = static_cast<unsigned long long>(epoch) << 32;  // (reads as 0.0)
#pragma unroll
          for (int u = 0; u < kSweep; ++u) {
            const unsigned idx = w + u * stripes;
            ga[u] = gp + static_cast<size_t>(idx < num_wg ? idx : 0u) * (2 * kAcc);
          }
          load_granule_pairs(gr, ga);  // (as 8-byte loads: 3927 against 4065 scans/s)
#pragma unroll
          for (int u = 0; u < kSweep; ++u)
            if (w + u * stripes >= num_wg) gr[u] = hg_u64x2{none, none};
          bool pending = true;
          unsigned long long t_first = 0;
          for (unsigned spins = 0; pending; ++spins) {
            pending = false;
#pragma unroll
            for (int u = 0; u < kSweep; ++u) {
              if (static_cast<unsigned>(gr[u].x >> 32) != epoch || static_cast<unsigned>(gr[u].y >> 32) != epoch) {
                gr[u] = load_granule_pair(ga[u]);
                pending = true;
              }
            }
            if (pending && spins == 0u) t_first = __builtin_amdgcn_s_memrealtime();
            // (back off between re-reads: the other workgroups of the launch, and other processes on the device,
            // share the L2 this loop polls through -- ADVICE r5. Each 8-byte half of a 16-byte sc1 load is taken
            // as single-copy atomic: observed untorn on gfx950 / ROCm 7.2, MI355X_MICROARCH.md "R2's granule"; a
            // torn half would carry a stale tag and simply be read again)
            if (pending) __builtin_amdgcn_s_sleep(1);
            if (pending && (spins & 63u) == 63u && __builtin_amdgcn_s_memrealtime() - t_first > 200000000ull) {
              // (two seconds of polling -- s_memrealtime counts 100 MHz -- cannot happen: every workgroup of the launch
              // runs, the others never wait for this one) the sum becomes NaN and the step is reported as failed
              // instead of spinning for ever
              gr[0] = hg_u64x2{0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull};
#pragma unroll
              for (int u = 1; u < kSweep; ++u) gr[u] = hg_u64x2{none, none};
              if (timeout_flag) *timeout_flag = 1;
              break;
            }
          }
#pragma unroll
          for (int u = 0; u < kSweep; ++u)
            acc += __longlong_as_double(static_cast<long long>((gr[u].y << 32) | (gr[u].x & 0xFFFFFFFFull)));
        }
      } else {
        for (unsigned w = j; w < num_wg; w += 16 * stripes) {
          double v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const unsigned idx = w + u * stripes;
            v[u] = idx < num_wg ? load_partial(&p[static_cast<size_t>(idx) * kAcc]) : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 16; ++u) acc += v[u];
        }
      }
      scratch[j * kAcc + k] = acc;
    }
    __syncthreads();
    if (FIRST) {
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(&sh);
#pragma unroll
      for (int u = 0; u < kUp; ++u)
        if (up_idx[u] < sizeof(LmHead) / 8) dst[up_idx[u]] = up_val[u];
    }
    if (t < kAcc) {
      double sum = 0.0;
      for (int jj = 0; jj < stripes; ++jj) sum += scratch[jj * kAcc + t];
      scratch[stripes * kAcc + t] = sum;
    }
    __syncthreads();
    if (t >= kLmThreads) return;
    if (FIRST) {
      // the whole head goes to device memory for the following launches; the fields this step changes
      // are stored again at the end by the same wavefront (stores of one wavefront stay in order)
      const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&sh);
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(&gh);
      for (unsigned i = t; i < sizeof(LmHead) / 8; i += kLmThreads) dst[i] = src[i];
    }
    scratch += stripes * kAcc;
  }
  TAIL_STAMP(1);
  // --- solver state (uniform loads) ---
  const hg_solver_opts& opt = sh.opt;  // read from LDS where used: the step is short of registers
  PinBox* const box = sh.box;
  const unsigned long long seq = sh.seq;
  int iteration = sh.iteration, phase = sh.phase, step_is_successful = sh.step_is_successful;
  int reuse_diagonal = sh.reuse_diagonal, invalid_steps = sh.invalid_steps;
  int num_iterations = sh.num_iterations, num_successful = sh.num_successful;
  int num_unsuccessful = sh.num_unsuccessful, num_cost_evals = sh.num_cost_evals;
  int num_jac_evals = sh.num_jac_evals;
  int done = 0, termination_type = sh.termination_type, termination_reason = sh.termination_reason;
  double radius = sh.radius, decrease_factor = sh.decrease_factor, x_cost = sh.x_cost;
  double model_cost_change = sh.model_cost_change, gradient_max = sh.gradient_max_norm;
  double initial_cost = sh.initial_cost;
  double x[7], cand[7], scale[6], diagonal[6], g[6];
  // H (J^T J at x) stays in LDS (sH, band layout b6): 21 doubles fewer in registers -- the step kept
  // more state than the 256 VGPRs hold and spilled 50 of them to scratch inside this serial path
#define HX(i, j) sH[b6(i, j)]
#pragma unroll
  for (int k = 0; k < 7; ++k) { x[k] = sh.x[0][k]; cand[k] = sh.cand[0][k]; }
#pragma unroll
  for (int k = 0; k < 6; ++k) { scale[k] = sh.scale[k]; diagonal[k] = sh.diagonal[k]; g[k] = sh.g[k]; }

  TAIL_STAMP(2);
  // A7 (symmetric 7 x 7), b7 = J^T r, c = r^T r from the 36 sums (constant indices after unrolling)
  double A7m[7][7], b7[7];
  {
    int idx = 0;
#pragma unroll
    for (int a = 0; a < 7; ++a)
#pragma unroll
      for (int b = a; b < 7; ++b) {
        const double v = scratch[idx++];
        A7m[a][b] = v;
        A7m[b][a] = v;
      }
#pragma unroll
    for (int a = 0; a < 7; ++a) b7[a] = scratch[28 + a];
  }
  const double rtr = scratch[35];
#define A7(a, b) A7m[a][b]

  // --- assemble at the candidate: Hc = M^T A7 M, gc = M^T b7, M = diag(I3, dq/dlocal(cand q)) ---
  double P[12];
  quaternion_plus_jacobian(cand + 3, P);
  double Hc[21], gc[6];
  double AP[7][3];  // A7[:, 3:7] * P
#pragma unroll
  for (int a = 0; a < 7; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double v = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) v += A7(a, 3 + r) * P[r * 3 + c];
      AP[a][c] = v;
    }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) Hc[i * (i + 1) / 2 + j] = A7(i, j);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int j = 0; j < 3; ++j) Hc[(3 + c) * (4 + c) / 2 + j] = AP[j][c];
#pragma unroll
    for (int c2 = 0; c2 <= c; ++c2) {
      double v = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) v += P[r * 3 + c] * AP[3 + r][c2];
      Hc[(3 + c) * (4 + c) / 2 + 3 + c2] = v;
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) gc[i] = b7[i];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    double v = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) v += P[r * 3 + c] * b7[3 + r];
    gc[3 + c] = v;
  }
  const double cand_cost = 0.5 * rtr;
#undef A7

  auto finish_ = [&](int type, int reason) { done = 1; termination_type = type; termination_reason = reason; };
  // |x - Plus(x, -g)|_inf (TrustRegionMinimizer::EvaluateGradientAndJacobian)
  auto gradient_max_norm_ = [&]() {
    double m = 0.0;
    const double neg[6] = {-g[0], -g[1], -g[2], -g[3], -g[4], -g[5]};
#pragma unroll
    for (int k = 0; k < 3; ++k) m = fmax(m, fabs(x[k] - (x[k] + neg[k])));
    double q[4];
    quaternion_plus(x + 3, neg + 3, q);
#pragma unroll
    for (int k = 0; k < 4; ++k) m = fmax(m, fabs(x[3 + k] - q[k]));
    return m;
  };
  bool h_changed = false;
  bool want_candidate = false;
  TAIL_STAMP(3);
  // (persistent solve) a workgroup's sums never arrived: the solve ends here as FAILURE, x stays the last accepted point
  const bool timed_out = timeout_flag != nullptr && *timeout_flag != 0;
  if (timed_out) {
    finish_(2, 7);
  } else if (phase == PHASE_INIT) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) HX(i, j) = Hc[i * (i + 1) / 2 + j];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      g[k] = gc[k];
      scale[k] = opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(Hc[k * (k + 1) / 2 + k])) : 1.0;
    }
    ++num_cost_evals;
    ++num_jac_evals;
    x_cost = cand_cost;
    initial_cost = cand_cost;
    gradient_max = gradient_max_norm_();
    step_is_successful = 1;
    num_iterations = 1;
    phase = PHASE_CANDIDATE;
    h_changed = true;
    want_candidate = true;
  } else {
    double sn = 0.0, xn = 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const double d = x[k] - cand[k];
      sn += d * d;
      xn += x[k] * x[k];
    }
    sn = sqrt(sn);
    xn = sqrt(xn);
    const double cost_change = x_cost - cand_cost;
    const bool ptol = sn <= opt.parameter_tolerance * (xn + opt.parameter_tolerance);
    const bool ftol = fabs(cost_change) <= opt.function_tolerance * x_cost;
    const double relative_decrease = cost_change / model_cost_change;
    const bool accept = relative_decrease > opt.min_relative_decrease;
    ++num_cost_evals;
    if (ptol) {
      finish_(0, 2);
    } else if (ftol) {
      finish_(0, 3);
    } else {
      if (accept) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) HX(i, j) = Hc[i * (i + 1) / 2 + j];
#pragma unroll
        for (int k = 0; k < 6; ++k) g[k] = gc[k];
#pragma unroll
        for (int k = 0; k < 7; ++k) x[k] = cand[k];
        h_changed = true;
        x_cost = cand_cost;
        ++num_jac_evals;
        gradient_max = gradient_max_norm_();
        step_is_successful = 1;
        const double tt = 2.0 * relative_decrease - 1.0;
        radius = radius / fmax(1.0 / 3.0, 1.0 - tt * tt * tt);
        radius = fmin(opt.max_trust_region_radius, radius);
        decrease_factor = 2.0;
        reuse_diagonal = 0;
      } else {
        radius = radius / decrease_factor;
        decrease_factor *= 2.0;
        reuse_diagonal = 1;
      }
      want_candidate = true;
    }
  }
  TAIL_STAMP(4);
  // LevenbergMarquardtStrategy::ComputeStep + ComputeTrustRegionStep (see compute_next_candidate)
  while (want_candidate) {
    const bool stop_iter = iteration >= opt.max_num_iterations;
    const bool stop_grad = step_is_successful && gradient_max <= opt.gradient_tolerance;
    const bool stop_rad = radius <= opt.min_trust_region_radius;
    if (step_is_successful) ++num_successful; else ++num_unsuccessful;
    if (stop_iter) { finish_(1, 4); break; }
    if (stop_grad) { finish_(0, 1); break; }
    if (stop_rad) { finish_(0, 5); break; }
    ++iteration;
    ++num_iterations;
    step_is_successful = 0;
    if (!reuse_diagonal) {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const double sd = HX(k, k) * scale[k] * scale[k];
        diagonal[k] = fmin(fmax(sd, opt.min_lm_diagonal), opt.max_lm_diagonal);
      }
    }
    double L[6][6], inv[6], rhs[6], y[6], step[6];
    // lm_a^2 = sqrt(diagonal_a / radius)^2: lane a evaluates entry a, the wavefront reads them back
    double lm2[6];
    {
      const int ln = t & 63;
      double mine = diagonal[0];
#pragma unroll
      for (int a = 1; a < 6; ++a) mine = (ln == a) ? diagonal[a] : mine;
      const double lm = sqrt(mine / radius);
      const double sq = lm * lm;
#pragma unroll
      for (int a = 0; a < 6; ++a) lm2[a] = readlane_f64(sq, a);
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int b = 0; b <= a; ++b) {
        double v = HX(a, b) * scale[a] * scale[b];
        if (a == b) v += lm2[a];
        L[a][b] = v;
      }
      rhs[a] = g[a] * scale[a];
    }
    bool valid = true;
    // (round 5, as in hg_btd.h) right-looking with explicit fused multiply-adds: the updates of a column are
    // independent, the serial chain per column is rsq -> Newton step -> scale -> update; the right-hand side rides
    // through the same loop; 1 / sqrt(d) = hardware estimate + one Newton step. The left-looking form was a
    // chain of ~200 dependent multiplies and adds.
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = rhs[i];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const double d = L[j][j];
      valid = valid && d > 0.0 && d < 1e300;
      const double y0 = __builtin_amdgcn_rsq(d);
      const double r = fma(y0, fma(-(d * y0), 0.5 * y0, 0.5), y0);
      inv[j] = r;
      y[j] *= r;
#pragma unroll
      for (int i = j + 1; i < 6; ++i) L[i][j] *= r;
#pragma unroll
      for (int i = j + 1; i < 6; ++i) {
#pragma unroll
        for (int k = j + 1; k <= i; ++k) L[i][k] = fma(-L[i][j], L[k][j], L[i][k]);
        y[i] = fma(-L[i][j], y[j], y[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) step[i] = y[i];
#pragma unroll
    for (int k = 5; k >= 0; --k) {
      step[k] *= inv[k];
#pragma unroll
      for (int j = 0; j < k; ++j) step[j] = fma(-L[k][j], step[k], step[j]);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) valid = valid && isfinite(step[i]);
    double mcc = 0.0;
    if (valid) {
#pragma unroll
      for (int k = 0; k < 6; ++k) step[k] = -step[k];
      // model_cost_change = -(step . g_s + step . H_s step / 2) with H_s step = -g_s - D step (the system just solved,
      // D = lm2): sum_a step_a (lm2_a step_a - g_s,a) / 2 -- six terms instead of a 6 x 6 product
      double part = 0.0;
#pragma unroll
      for (int a = 0; a < 6; ++a) part = fma(0.5 * step[a], fma(lm2[a], step[a], -rhs[a]), part);
      mcc = part;
      valid = mcc > 0.0;
    }
    if (!valid) {
      const bool fail = (invalid_steps + 1) >= 5;  // max_num_consecutive_invalid_steps
      ++invalid_steps;
      reuse_diagonal = 1;
      if (fail) { finish_(2, 6); break; }
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      continue;
    }
    reuse_diagonal = 1;
    invalid_steps = 0;
    model_cost_change = mcc;
    double delta[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) delta[k] = step[k] * scale[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) cand[k] = x[k] + delta[k];
    quaternion_plus(x + 3, delta + 3, cand + 3);
    break;
  }
  if (done) {
    // the general path leaves cand = x's last candidate; nothing reads it after termination
  }
  TAIL_STAMP(5);
#ifdef HG_PERSIST_LATE_HANDBACK
  if (false) {
#else
  if (persist_bcast) {
#endif
    // persistent solve: the hand-back leaves FIRST, straight from the registers of this wavefront (every lane holds the
    // state): lane l writes granule pair l % 8 of replicas l / 8, + 8, + 16, + 24. Behind the write-back of the head
    // it waited for ~80 scalar-width stores to drain and a workgroup barrier -- on the critical path of all 196
    // workgroups, which do nothing until they see it.
    const int k = t & 7;
    double v = done ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < 7; ++q) v = (k == q) ? cand[q] : v;
#pragma unroll
    for (unsigned r = 0; r < kBcastReplicas; r += 8u)
      store_partial_tagged(persist_bcast + 16u * (r + (static_cast<unsigned>(t) >> 3)) + 2u * static_cast<unsigned>(k), v, epoch + 1u);
  }
  // --- write back (lane 0; every lane holds the same values) ---
  if (t == 0) {
    auto store_fields = [&](LmHead& d) {
      d.iteration = iteration; d.phase = phase; d.step_is_successful = step_is_successful;
      d.reuse_diagonal = reuse_diagonal; d.invalid_steps = invalid_steps;
      d.num_iterations = num_iterations; d.num_successful = num_successful;
      d.num_unsuccessful = num_unsuccessful; d.num_cost_evals = num_cost_evals;
      d.num_jac_evals = num_jac_evals; d.done = done;
      d.termination_type = termination_type; d.termination_reason = termination_reason;
      d.radius = radius; d.decrease_factor = decrease_factor; d.x_cost = x_cost;
      d.cand_cost = cand_cost; d.model_cost_change = model_cost_change;
      d.gradient_max_norm = gradient_max; d.initial_cost = initial_cost;
#pragma unroll
      for (int k = 0; k < 7; ++k) { d.x[0][k] = x[k]; d.cand[0][k] = cand[k]; }
#pragma unroll
      for (int k = 0; k < 6; ++k) { d.scale[k] = scale[k]; d.diagonal[k] = diagonal[k]; d.g[k] = g[k]; d.gc[k] = gc[k]; }
    };
    store_fields(gh);
    if (persist_out) {
#pragma unroll
      for (int k = 0; k < 7; ++k) persist_out[k] = cand[k];
      persist_out[7] = done ? 1.0 : 0.0;
    }
    if (h_changed) {
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) G->H[b6(i, j)] = HX(i, j);
    }
    if (!done) {
#pragma unroll
      for (int k = 0; k < 3; ++k) xf[0].t[k] = cand[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) xf[0].q[k] = cand[3 + k];
    } else if (box) {
      // the mailbox head still holds the uploaded state: only the fields this path changes go out
      store_fields(box->h);
      __threadfence_system();
      *reinterpret_cast<volatile unsigned long long*>(&box->flag) = seq;
      __threadfence_system();
    }
  }
}

// ------------------------------------------------------------------------------------------
// Odometry / IMU residual blocks: one wavefront per block, lane k carries d/d(parameter k) of
// every intermediate (forward-mode differentiation spread over the lanes; scalar parts are
// computed redundantly). Parameter lanes: t_a 0-2, v_a 3-5, q_a 6-9, t_b 10-12, v_b 13-15,
// q_b 16-19 (Ceres Jet arithmetic, jet.h).
// ------------------------------------------------------------------------------------------
struct LJ {
  double a, v;
};
__device__ inline LJ lj(double a) { return {a, 0.0}; }
__device__ inline LJ operator+(const LJ& f, const LJ& g) { return {f.a + g.a, f.v + g.v}; }
__device__ inline LJ operator-(const LJ& f, const LJ& g) { return {f.a - g.a, f.v - g.v}; }
__device__ inline LJ operator-(const LJ& f) { return {-f.a, -f.v}; }
__device__ inline LJ operator*(const LJ& f, const LJ& g) { return {f.a * g.a, f.a * g.v + f.v * g.a}; }
__device__ inline LJ operator*(double s, const LJ& f) { return {f.a * s, f.v * s}; }
__device__ inline LJ operator/(const LJ& f, const LJ& g) {
  const double gi = 1.0 / g.a, fg = f.a * gi;
  return {fg, (f.v - fg * g.v) * gi};
}
__device__ inline LJ lj_sqrt(const LJ& f) {
  const double t = sqrt(f.a);
  return {t, f.v * (1.0 / (2.0 * t))};
}
__device__ inline LJ lj_atan2(const LJ& g, const LJ& f) {
  const double tmp = 1.0 / (f.a * f.a + g.a * g.a);
  return {atan2(g.a, f.a), tmp * (-g.a * f.v + f.a * g.v)};
}
__device__ inline LJ lj_asin(const LJ& f) {
  const double tmp = 1.0 / sqrt(1.0 - f.a * f.a);
  return {asin(f.a), tmp * f.v};
}
struct LJ3 { LJ x, y, z; };
struct LJ4 { LJ w, x, y, z; };
__device__ inline LJ3 lj_cross(const LJ3& a, const LJ3& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ inline LJ3 lj_rotate(const LJ4& q, const LJ3& v) {  // Eigen _transformVector
  const LJ3 u{q.x, q.y, q.z};
  LJ3 uv = lj_cross(u, v);
  uv = {uv.x + uv.x, uv.y + uv.y, uv.z + uv.z};
  const LJ3 c = lj_cross(u, uv);
  return {v.x + q.w * uv.x + c.x, v.y + q.w * uv.y + c.y, v.z + q.w * uv.z + c.z};
}
__device__ inline LJ4 lj_qmul(const LJ4& a, const LJ4& b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
__device__ inline LJ4 lj_qnormalized(const LJ4& q) {
  const LJ n = lj_sqrt((q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w));
  return {q.w / n, q.x / n, q.y / n, q.z / n};
}
struct LJRigid { LJ3 t; LJ4 q; };
__device__ inline LJRigid lj_inverse(const LJRigid& r) {  // rigid_transform.h:159-163
  const LJ4 rc{r.q.w, -r.q.x, -r.q.y, -r.q.z};
  const LJ3 t = lj_rotate(rc, r.t);
  return {{-t.x, -t.y, -t.z}, rc};
}
__device__ inline LJRigid lj_mul(const LJRigid& a, const LJRigid& b) {  // :184-190
  const LJ3 t = lj_rotate(a.q, b.t);
  return {{t.x + a.t.x, t.y + a.t.y, t.z + a.t.z}, lj_qnormalized(lj_qmul(a.q, b.q))};
}

// One wavefront (lanes 0..63 of the calling workgroup; the arrays are its own).
__device__ __forceinline__ void small_block_eval(const LmState* G, int b, SmallOut* out, double* residuals) {
  const int lane = threadIdx.x;
  const SmallBlockDev sb = G->h.small[b];
  if (!sb.active) return;
  __shared__ double Ja[9][20];   // ambient Jacobian rows
  __shared__ double Jl[9][18];   // local Jacobian rows
  __shared__ double rs[9];
  const double* A = G->h.cand[sb.a];
  const double* B = G->h.cand[sb.b];
  auto var = [&](double a, int k) { return LJ{a, lane == k ? 1.0 : 0.0}; };
  const LJ3 ta{var(A[0], 0), var(A[1], 1), var(A[2], 2)};
  const LJ3 va{var(A[7], 3), var(A[8], 4), var(A[9], 5)};
  const LJ4 qa{var(A[3], 6), var(A[4], 7), var(A[5], 8), var(A[6], 9)};
  const LJ3 tb{var(B[0], 10), var(B[1], 11), var(B[2], 12)};
  const LJ3 vb{var(B[7], 13), var(B[8], 14), var(B[9], 15)};
  const LJ4 qb{var(B[3], 16), var(B[4], 17), var(B[5], 18), var(B[6], 19)};
  LJ r[9];
  int rows;
  if (sb.type == 1) {
    // relative_translation_and_yaw_cost_function.h:41-63
    rows = 6;
    const LJRigid delta = lj_mul(lj_inverse({tb, qb}), {ta, qa});
    const LJRigid dc{{lj(sb.delta[0]), lj(sb.delta[1]), lj(sb.delta[2])},
                     {lj(sb.delta[3]), lj(sb.delta[4]), lj(sb.delta[5]), lj(sb.delta[6])}};
    const LJRigid e = lj_mul(lj_inverse(delta), dc);
    r[0] = sb.w[0] * e.t.x;
    r[1] = sb.w[0] * e.t.y;
    r[2] = sb.w[0] * e.t.z;
    const LJ4& q = e.q;
    r[3] = sb.w[1] * lj_atan2(lj(2.0) * (q.w * q.x + q.y * q.z), lj(1.0) - lj(2.0) * (q.x * q.x + q.y * q.y));
    const LJ sinp = lj(2.0) * (q.w * q.y - q.z * q.x);
    r[4] = sb.w[1] * ((fabs(sinp.a) >= 1.0) ? lj(M_PI / 2) : lj_asin(sinp));
    const LJ3 d = lj_rotate(q, {lj(1.0), lj(0.0), lj(0.0)});
    r[5] = sb.w[1] * lj_atan2(d.y, d.x);
    r[6] = r[7] = r[8] = lj(0.0);
  } else {
    // prediction_imu_preintegration_cost_functor.h:49-101
    rows = 9;
    const LJ dt = lj(sb.dt);
    r[0] = sb.w[0] * (tb.x - ta.x - dt * va.x);
    r[1] = sb.w[0] * (tb.y - ta.y - dt * va.y);
    r[2] = sb.w[0] * (tb.z - ta.z - dt * va.z);
    r[3] = sb.w[1] * (vb.x - va.x);
    r[4] = sb.w[1] * (vb.y - va.y);
    r[5] = sb.w[1] * (vb.z - va.z);
    const LJ4 qbc{qb.w, -qb.x, -qb.y, -qb.z};
    const LJ4 dq{lj(sb.delta[3]), lj(sb.delta[4]), lj(sb.delta[5]), lj(sb.delta[6])};
    const LJ4 re = lj_qmul(lj_qmul(qbc, qa), dq);
    r[6] = sb.w[2] * re.x;
    r[7] = sb.w[2] * re.y;
    r[8] = sb.w[2] * re.z;
  }
  if (lane < 20)
    for (int i = 0; i < 9; ++i) Ja[i][lane] = (i < rows) ? r[i].v : 0.0;
  if (lane == 0)
    for (int i = 0; i < 9; ++i) rs[i] = (i < rows) ? r[i].a : 0.0;
  wave_sync();
  // ambient -> local columns [t_a 3 | rot_a 3 | v_a 3 | t_b 3 | rot_b 3 | v_b 3]
  if (lane < 18) {
    double pj[12];
    const int c = lane;
    for (int i = 0; i < 9; ++i) {
      double v;
      if (c < 3) v = Ja[i][c];
      else if (c < 6) {
        quaternion_plus_jacobian(A + 3, pj);
        v = 0.0;
        for (int j = 0; j < 4; ++j) v += Ja[i][6 + j] * pj[j * 3 + (c - 3)];
      } else if (c < 9) v = Ja[i][3 + (c - 6)];
      else if (c < 12) v = Ja[i][10 + (c - 9)];
      else if (c < 15) {
        quaternion_plus_jacobian(B + 3, pj);
        v = 0.0;
        for (int j = 0; j < 4; ++j) v += Ja[i][16 + j] * pj[j * 3 + (c - 12)];
      } else v = Ja[i][13 + (c - 15)];
      Jl[i][c] = v;
    }
  }
  wave_sync();
  SmallOut& o = out[b];
  {
    int c1 = 0, c2 = lane;  // entry `lane` of the lower triangle; 64 entries further is at most 11 rows down
    while (c2 > c1) { c2 -= c1 + 1; ++c1; }
    for (int idx = lane; idx < kSmallTri; idx += kWave) {
      double s = 0.0;
      for (int i = 0; i < 9; ++i) s += Jl[i][c1] * Jl[i][c2];
      o.v[idx] = s;
      c2 += kWave;
      while (c2 > c1) { c2 -= c1 + 1; ++c1; }
    }
  }
  if (lane < 18) {
    double s = 0.0;
    for (int i = 0; i < 9; ++i) s += Jl[i][lane] * rs[i];
    o.v[kSmallTri + lane] = s;
  }
  if (lane == 0) {
    double s = 0.0;
    for (int i = 0; i < 9; ++i) s += rs[i] * rs[i];
    o.v[kSmallTri + 18] = s;
    if (residuals)
      for (int i = 0; i < rows; ++i) residuals[sb.row_offset + i] = rs[i];
  }
}

__global__ __launch_bounds__(kWave) void k_small_blocks(LmState* G, SmallOut* out, double* residuals) {
  if (G->h.done) return;
  small_block_eval(G, blockIdx.x, out, residuals);
}


#ifdef HG_EVAL_STAMPS
// diagnostic build only: per-workgroup timeline of the last k_tsdf_residuals launch (100 MHz clock)
__device__ unsigned long long g_eval_stamps[1024][4];
#define EVAL_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024 && eval_it == 3) g_eval_stamps[blockIdx.x][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define EVAL_STAMP(i) do {} while (0)
#endif

// The single-pose registration step (one free pose, one per-scan block): the same evaluation with
// the register-resident LM step in the tail, in a kernel of its own so that its LDS footprint is
// the X tiles (36 KB) instead of the general solver's working set (139 KB). (256-thread workgroups
// — 391 of them, on all 256 CUs — were measured slower: twice the partials for the tail to sum.)
// The first launch of a solve carries the initial pose and the mailbox of the solver head as
// kernel arguments (FirstUpload): no separate upload kernel in front of the solve (it cost 5 us plus a
// launch gap per registration). The workgroups take the transform from the arguments; the tail
// builds the head from the mailbox words (one host-link round trip, in flight together with the
// partial loads) and stores it to device memory for the following launches.
struct FirstUpload {
  double pose[7];
  const PinBox* box;
  unsigned up_words;
  unsigned pad;
};

template <int THREADS, bool FIRST = false>
__device__ __forceinline__ void single_eval(const PyramidView& pv, const float* __restrict__ xyz, unsigned n,
                                            unsigned width, double scaling, const BlockXform* __restrict__ xf,
                                            double* __restrict__ partials, LmState* G, unsigned* ticket,
                                            unsigned wg_index, unsigned num_wg, unsigned epoch,
                                            const FirstUpload* up = nullptr) {
  if (!FIRST && G->h.done) return;
#ifdef HG_EVAL_STAMPS
  const int eval_it = G->h.iteration;
  if (threadIdx.x == 0 && blockIdx.x == 0) g_tail_stamps[7] = eval_it;
#endif
  EVAL_STAMP(0);
  constexpr size_t kTiles = (THREADS / kWave) * (kWave * 8 + 64) * sizeof(double);
  constexpr size_t kTail = ((THREADS / kAcc) + 1) * kAcc * sizeof(double) + sizeof(LmHead) + 36 * sizeof(double);
  __shared__ __align__(16) unsigned char smem[kTiles > kTail ? kTiles : kTail];
  tsdf_residuals_body<THREADS>(pv, xyz, n, scaling, xf, partials, nullptr,
                               reinterpret_cast<double (*)[kWave][8]>(smem),
                               reinterpret_cast<double (*)[64]>(smem + (THREADS / kWave) * kWave * 8 * sizeof(double)),
                               xcd_chunk(wg_index, num_wg), FIRST ? up->pose : nullptr, width, 1, 0, epoch);
  EVAL_STAMP(1);
  if (epoch != 0u) {
    // tagged partial sums (store_partial_tagged): workgroup 0 runs the LM step and waits for the granules themselves
    // (the workgroup launched last instead: 3753 against 3790-3830 scans/s)
    if (wg_index != 0u) return;
    __syncthreads();  // (ends this workgroup's use of the tiles in smem)
    EVAL_STAMP(2);
    lm_step_single<FIRST>(reinterpret_cast<double*>(smem), G, const_cast<BlockXform*>(xf), partials, num_wg,
                          FIRST ? up->box : nullptr, FIRST ? up->up_words : 0u, epoch);
    EVAL_STAMP(3);
    return;
  }
  __shared__ int s_last;
  // hand-over of the partials without fences: sc1 stores drained here, one counted arrival per
  // workgroup behind the barrier, sc1 loads in the tail (store_partial / load_partial)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's partial stores have left
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(ticket, 1u);
    s_last = (t == num_wg - 1u) ? 1 : 0;
    if (s_last) *ticket = 0u;  // ready for the next iteration's launch
  }
  __syncthreads();  // the arrival count has returned to wave 0 before any wave loads a partial
  if (!s_last) return;
  EVAL_STAMP(2);
  lm_step_single<FIRST>(reinterpret_cast<double*>(smem), G, const_cast<BlockXform*>(xf), partials, num_wg,
                        FIRST ? up->box : nullptr, FIRST ? up->up_words : 0u);
  EVAL_STAMP(3);
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_tsdf_residuals_single(
    PyramidView pv, const float* __restrict__ xyz, unsigned n, unsigned width, double scaling,
    const BlockXform* __restrict__ xf, double* __restrict__ partials, LmState* G, unsigned* ticket, unsigned epoch) {
  single_eval<THREADS>(pv, xyz, n, width, scaling, xf, partials, G, ticket, blockIdx.x, gridDim.x, epoch);
}
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_tsdf_residuals_single_first(
    PyramidView pv, const float* __restrict__ xyz, unsigned n, unsigned width, double scaling,
    const BlockXform* __restrict__ xf, double* __restrict__ partials, LmState* G, unsigned* ticket,
    FirstUpload up, unsigned epoch) {
  single_eval<THREADS, true>(pv, xyz, n, width, scaling, xf, partials, G, ticket, blockIdx.x, gridDim.x, epoch, &up);
}

#ifndef HG_BIG
// The whole single-pose solve in ONE launch (round 6). The launch-per-evaluation chain pays
// 1.8 us between dependent launches, the ramp of 196 workgroups and one or two empty launches behind convergence,
// per evaluation: ~3 of its ~14 us. Here every workgroup stays and loops over the evaluations:
//   body at the current candidate -> partial sums as granules tagged with the evaluation's epoch (as before)
//   workgroup 0: the LM step on the sums, then the next candidate + done flag as eight granules tagged epoch + 1
//   the others: one wavefront polls those eight granules (sc1 loads, s_sleep between polls), a barrier, next body.
// Same arithmetic in the same order as the launch-per-evaluation chain: bitwise the same poses and iterations.
// Every wait is bounded (two seconds of s_memrealtime); a workgroup that gives up leaves, its granules then never
// arrive, workgroup 0 times out in turn and ends the solve as FAILURE (termination_reason 7) -- the host then stops
// using this form on the context. That can only happen when the 196 workgroups are not resident together, which the
// host rules out before it takes this path (persist_ok: occupancy x CUs >= workgroups, this context the only one of
// the process); bench.py --submaps (several PROCESSES on one GPU) is the case the bound exists for.
#ifdef HG_PERSIST_STAMPS
// diagnostics: s_memrealtime (10 ns) per evaluation: workgroup 0 [e][0] body start, [1] body end, [2] LM step end,
// [3] hand-back stored; workgroup 97 [4] hand-back seen, [5] body end; [6] workgroup 195's body end
__device__ unsigned long long g_persist_stamps[16][8];
#define PSTAMP(w, k) do { if (threadIdx.x == 0 && wg == (w) && e < 16u) g_persist_stamps[e][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PSTAMP(w, k) do {} while (0)
#endif
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_tsdf_residuals_single_persist(
    PyramidView pv, const float* __restrict__ xyz, unsigned n, unsigned width, double scaling,
    const BlockXform* __restrict__ xf, double* __restrict__ granules, LmState* G, FirstUpload up, unsigned epoch0,
    unsigned evals, unsigned long long* __restrict__ bcast) {
  constexpr size_t kTiles = (THREADS / kWave) * (kWave * 8 + 64) * sizeof(double);
  constexpr size_t kTail = ((THREADS / kAcc) + 1) * kAcc * sizeof(double) + sizeof(LmHead) + 36 * sizeof(double);
  __shared__ __align__(16) unsigned char smem[kTiles > kTail ? kTiles : kTail];
  __shared__ double s_out[8];
  __shared__ int s_timeout;
  const unsigned wg = blockIdx.x, num_wg = gridDim.x;
  // The candidate of an evaluation is read from s_out at its start (the first one put there from the kernel argument):
  // carried in registers from the end of one evaluation to the next it lived across workgroup 0's LM step, where it was
  // spilled in the prologue and reloaded behind every exit of the step.
  if (threadIdx.x < 7u) s_out[threadIdx.x] = up.pose[threadIdx.x];
  if (threadIdx.x == 7u) s_out[7] = 0.0;
  if (threadIdx.x == 0) s_timeout = 0;  // (the body's barriers come before anything reads it)
  __syncthreads();
#ifdef HG_EVAL_STAMPS
  const int eval_it = 0;
#endif
  // what does not change from one evaluation to the next is loaded once: the lane's return and the levels' window
  // counters (the map is not written while the solve runs)
#ifndef HG_PERSIST_NO_HOIST
  double pre_v[3];
  {
    const ScanOrder order = make_scan_order(n, width);
    const unsigned first_i0 = xcd_chunk(wg, num_wg) * THREADS + threadIdx.x;
    load_point(xyz, scan_index(order, first_i0 < n ? first_i0 : 0u), pre_v);
  }
  const DirectRaw pre_dp = direct_issue(pv);
#endif
  for (unsigned e = 0; e < evals; ++e) {
    const unsigned epoch = epoch0 + e;
    double pose[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) pose[k] = uniform_f64(s_out[k]);  // (into scalar registers, as a kernel argument would be)
    __syncthreads();  // everyone has the candidate before workgroup 0's step rewrites s_out
    PSTAMP(0u, 0);
    tsdf_residuals_body<THREADS>(pv, xyz, n, scaling, xf, granules, nullptr,
                                 reinterpret_cast<double (*)[kWave][8]>(smem),
                                 reinterpret_cast<double (*)[64]>(smem + (THREADS / kWave) * kWave * 8 * sizeof(double)),
#ifdef HG_PERSIST_NO_HOIST
                                 xcd_chunk(wg, num_wg), pose, width, 1, 0, epoch);
#else
                                 xcd_chunk(wg, num_wg), pose, width, 1, 0, epoch, pre_v, &pre_dp);
#endif
    PSTAMP(0u, 1);
    PSTAMP(97u, 5);
    PSTAMP(195u, 6);
    if (wg == 0u) {
      __syncthreads();  // (ends this workgroup's use of the tiles in smem)
      lm_step_single<false, true>(reinterpret_cast<double*>(smem), G, const_cast<BlockXform*>(xf), granules, num_wg, up.box,
                                  up.up_words, epoch, s_out, &s_timeout, bcast, e == 0u);
      // (the hand-back -- kBcastReplicas copies, a 128-byte line each: 195 workgroups polling ONE line queue at its
      // memory channel -- has left from inside the step, ahead of the head's write-back)
      PSTAMP(0u, 2);
      __syncthreads();
#ifdef HG_PERSIST_LATE_HANDBACK
      if (threadIdx.x < 8u * kBcastReplicas)
        store_partial_tagged(bcast + 16u * (threadIdx.x >> 3) + 2u * (threadIdx.x & 7u), s_out[threadIdx.x & 7u], epoch + 1u);
#endif
      PSTAMP(0u, 3);
    } else {
      if (e + 1u == evals) break;  // (uniform) nothing follows the last evaluation
      if (threadIdx.x < 8u) {
        const unsigned long long* g = bcast + 16u * (wg % kBcastReplicas) + 2u * threadIdx.x;
        unsigned long long t_first = 0;
        for (unsigned spins = 0;; ++spins) {
          const hg_u64x2 v = load_granule_pair16(g);
          if (static_cast<unsigned>(v.x >> 32) == epoch + 1u && static_cast<unsigned>(v.y >> 32) == epoch + 1u) {
            s_out[threadIdx.x] = __longlong_as_double(static_cast<long long>((v.y << 32) | (v.x & 0xFFFFFFFFull)));
            break;
          }
          if (spins == 0u) t_first = __builtin_amdgcn_s_memrealtime();
          if ((spins & 15u) == 15u && __builtin_amdgcn_s_memrealtime() - t_first > 200000000ull) {
            s_timeout = 1;  // (s_memrealtime counts 100 MHz: two seconds) workgroup 0 is not there: leave
            break;
          }
          __builtin_amdgcn_s_sleep(4);
        }
      }
      __syncthreads();
      PSTAMP(97u, 4);
    }
    if (s_timeout != 0) break;     // (uniform per workgroup)
    if (s_out[7] != 0.0) break;    // the solve has terminated
  }
}
#endif  // !HG_BIG

// Several INDEPENDENT single-pose problems per launch (blockIdx.y = problem): each keeps its own
// state, partials, ticket and mailbox and runs exactly the arithmetic of k_tsdf_residuals_single,
// so results are identical to solving them one after the other; a problem that has terminated, or
// whose scan needs fewer workgroups than the widest of the batch, leaves at once. One registration
// chain is latency-bound and fills a fraction of the chip; a batch shares the launches.
struct SingleJob {
  PyramidView pv;
  const float* xyz;
  const BlockXform* xf;
  double* partials;
  LmState* G;
  unsigned* ticket;
  double scaling;
  unsigned n;
  unsigned num_wg;
  const PinBox* box;  // the problem's mailbox (compact upload of its solver head)
  unsigned up_words;
  unsigned width;     // returns per column of the structured scan, or 0 (scan_index)
  unsigned tiles;     // tiles of kBatchThreads returns per workgroup
  unsigned pad;
  const unsigned* fast_n;  // level partition: returns at the front of xyz that stopped at the finest level, or null
  unsigned char* flags;    // level partition, classifying launch: per return (lane order) whether it stopped at the finest level
  unsigned* wave_counts;   //   and per wavefront of 64 returns how many did
};

// Throughput form: the residual pass of all problems in one launch WITHOUT the LM step in its tail
// (that step keeps the whole solver state in registers -- 256 VGPRs, one workgroup per CU -- which
// is right for one latency-bound chain and wrong for a batch that has to hide memory latency with
// occupancy), followed by one launch in which workgroup b runs problem b's LM step. Partials cross a
// kernel boundary here, so no hand-over protocol is needed.
// (compiled for four workgroups per CU: 128 VGPRs instead of 134, 212 against 230 us per launch of 64 matches;
// five -- 102 VGPRs, 75 of them spilled -- 326 us)
#ifndef HG_BATCH_WAVES
#define HG_BATCH_WAVES 4
#endif
template <int THREADS, int MODE /* 0 plain, 1 over a level-partitioned cloud (staged lookup), 2 plain + classifies */>
__global__ __launch_bounds__(THREADS, HG_BATCH_WAVES) void k_tsdf_residuals_single_batch(const SingleJob* __restrict__ jobs) {
  const SingleJob& J = jobs[blockIdx.y];
  if (blockIdx.x >= J.num_wg) return;
  if (J.G->h.done) return;
  const PyramidView& pv = J.pv;
  __shared__ __align__(16) unsigned char smem[(THREADS / kWave) * (kWave * 8 + 64) * sizeof(double)];
  tsdf_residuals_body<THREADS, MODE == 1 ? 2 : MODE == 2 ? 3 : 1>(pv, J.xyz, J.n, J.scaling, J.xf, J.partials, nullptr,
                                     reinterpret_cast<double (*)[kWave][8]>(smem),
                                     reinterpret_cast<double (*)[64]>(smem + (THREADS / kWave) * kWave * 8 * sizeof(double)),
                                     xcd_chunk(blockIdx.x, J.num_wg), nullptr, J.width, J.tiles,
                                     J.fast_n ? *J.fast_n : 0u, 0u, nullptr, nullptr, J.flags, J.wave_counts);
}

// ------------------------------------------------------------------------------------------
// Level partition (round 5). The batched residual pass is bound by vector issue and the L1, and most of both is
// spent on pyramid levels most returns never use: a return takes the first level whose 8 voxels are all known
// (interpolated_multi_resolution_tsdf.h:99-106), and 78 % of the returns of the bench scene find it at the finest
// one. Which returns do is a property of the map and the pose, stable over a solve (the pose moves by centimetres),
// but they are scattered over the wavefronts (37 % of them hold no other lane). Before a batch of solves the returns
// of every problem are therefore SORTED by that property at the initial pose -- three small launches for the whole
// batch: classify (one finest-level lookup per return), scan of the per-workgroup counts, stable scatter of the
// coordinates into a buffer of the problem -- and the residual kernel treats the leading tiles as "expected to stop
// at the finest level" (pred_fast in pyramid_tsd_direct). A permutation of the returns: the normal equations are the
// same sums in another order (results to rounding, as for the structured lane order); deterministic.
// ------------------------------------------------------------------------------------------
struct PartJob {
  PyramidView pv;
  const float* xyz;      // the block's cloud
  float* out;            // the same returns, those that stop at the finest level first (both groups in lane order)
  unsigned char* flags;  // per return (in lane order): stops at the finest level
  unsigned* counts;      // per workgroup of 256 returns: how many do (k_level_scan: their exclusive prefix); [nwg] total
  const BlockXform* xf;  // the transform the returns are classified at: the candidate the solve evaluates next
  const unsigned* wave_counts;  // classified by the residual launch in front (round 6): per wavefront of 64 returns how many do, or null
  const LmState* G;      // the problem's solver state: a problem that has terminated is not partitioned
  unsigned n, nwg, width, pad;
};
__global__ __launch_bounds__(256) void k_level_classify(const PartJob* __restrict__ jobs) {
  const PartJob& J = jobs[blockIdx.y];
  if (blockIdx.x >= J.nwg) return;
  const ScanOrder order = make_scan_order(J.n, J.width);
  const unsigned i0 = blockIdx.x * 256u + threadIdx.x;
  double v[3];
  load_point(J.xyz, scan_index(order, i0 < J.n ? i0 : 0u), v);
  const DirectRaw raw = direct_issue(J.pv);
  const LevelPin lp = pin_level(J.pv.level[0]);
  // directly addressable? (as pyramid_tsd_direct; a pool that is not sends every lookup through the general path,
  // which has no staged form: nothing is expected fast)
  const su8 w = raw.w[0];
  bool lok = w[0] == 0u;
  uint32_t min_b[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lok = lok & ((w[4 + a] - w[1 + a]) < (1u << lp.bits[a]));
    min_b[a] = w[1 + a];
  }
  double tq[7];
  load_transform_uniform(J.xf->t, tq);
  bool fast = false;
  if (lok && i0 < J.n) {
    // the world point as return_row forms it
    const double qw = tq[3];
    const double u[3] = {tq[4], tq[5], tq[6]};
    double uv[3], c2[3];
    cross3(u, v, uv);
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    cross3(u, uv, c2);
    const double x = (v[0] + qw * uv[0] + c2[0]) + tq[0];
    const double y = (v[1] + qw * uv[1] + c2[1]) + tq[1];
    const double z = (v[2] + qw * uv[2] + c2[2]) + tq[2];
    const bool usable = !(fabs(x) >= 1e30 || fabs(y) >= 1e30 || fabs(z) >= 1e30);
    const float p[3] = {static_cast<float>(x), static_cast<float>(y), static_cast<float>(z)};
    const float rr = refined_rcp(lp.res);
    const float q[3] = {cell_quotient_fast(p[0], lp.res, rr), cell_quotient_fast(p[1], lp.res, rr), cell_quotient_fast(p[2], lp.res, rr)};
    DirectFetch f;
    direct_setup(lp, x, y, z, q, min_b, usable, f);
    direct_load(lp, f);
    direct_merge(f);
    fast = f.in & all_weights_valid(f.code);
  }
  if (i0 < J.n) J.flags[i0] = fast ? 1 : 0;
  __shared__ unsigned s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const unsigned long long m = __ballot(fast);
  if ((threadIdx.x & 63u) == 0 && m) atomicAdd(&s_cnt, static_cast<unsigned>(__popcll(m)));
  __syncthreads();
  if (threadIdx.x == 0) J.counts[blockIdx.x] = s_cnt;
}
__global__ __launch_bounds__(1024) void k_level_scan(const PartJob* __restrict__ jobs) {
  const PartJob& J = jobs[blockIdx.x];
  if (J.wave_counts && J.G->h.done) return;  // (uniform; its residual launches return at once, and nothing classified it)
  __shared__ unsigned s_wave[16], s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (unsigned base = 0; base < J.nwg; base += 1024u) {
    const unsigned i = base + threadIdx.x;
    unsigned v = 0u;
    if (i < J.nwg) {
      if (J.wave_counts) {  // (uniform) the four wavefronts of the group
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 w = *reinterpret_cast<const u4*>(J.wave_counts + 4u * i);
        v = w[0] + w[1] + w[2] + w[3];
      } else {
        v = J.counts[i];
      }
    }
    unsigned incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned t = static_cast<unsigned>(__shfl_up(static_cast<int>(incl), off));
      if (lane >= static_cast<unsigned>(off)) incl += t;
    }
    if (lane == 63u) s_wave[wave] = incl;
    __syncthreads();
    unsigned before = s_carry;
    for (unsigned k = 0; k < wave; ++k) before += s_wave[k];
    if (i < J.nwg) J.counts[i] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023u) s_carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) J.counts[J.nwg] = s_carry;
}
__global__ __launch_bounds__(256) void k_level_scatter(const PartJob* __restrict__ jobs) {
  const PartJob& J = jobs[blockIdx.y];
  if (blockIdx.x >= J.nwg) return;
  if (J.wave_counts && J.G->h.done) return;  // (as k_level_scan)
  const ScanOrder order = make_scan_order(J.n, J.width);
  const unsigned i0 = blockIdx.x * 256u + threadIdx.x;
  const bool in = i0 < J.n;
  const bool fast = in && J.flags[i0] != 0;
  __shared__ unsigned s_fast[4], s_slow[4];
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned long long mf = __ballot(fast), ms = __ballot(in && !fast);
  if (lane == 0) { s_fast[wave] = static_cast<unsigned>(__popcll(mf)); s_slow[wave] = static_cast<unsigned>(__popcll(ms)); }
  __syncthreads();
  unsigned bf = 0, bs = 0;
  for (unsigned k = 0; k < wave; ++k) { bf += s_fast[k]; bs += s_slow[k]; }
  const unsigned long long below = (1ull << lane) - 1ull;
  const unsigned fast_before = J.counts[blockIdx.x], total_fast = J.counts[J.nwg];
  if (in) {
    const unsigned pos = fast ? fast_before + bf + static_cast<unsigned>(__popcll(mf & below))
                              : total_fast + (blockIdx.x * 256u - fast_before) + bs + static_cast<unsigned>(__popcll(ms & below));
    typedef float f3 __attribute__((ext_vector_type(3), aligned(4)));
    const f3 pt = *reinterpret_cast<const __attribute__((address_space(1))) f3*>(as_global(J.xyz) + 12ull * scan_index(order, i0));
    *reinterpret_cast<__attribute__((address_space(1))) f3*>(const_cast<__attribute__((address_space(1))) char*>(as_global(J.out)) + 12ull * pos) = pt;
  }
}
// (A kernel over a job table reads its pointers from device memory, where nothing tells the compiler what they point to:
// every access through them became a FLAT operation -- 175 of them in this kernel against 1 in k_window_residuals --
// which counts against the LDS / scalar counter as well as the vector-memory one, so that a wait for an LDS read waits
// for the loads and stores in flight. The pointers pass through the device-memory address space once; the inline asm
// keeps the pair of casts from being folded away. p: uniform.)
template <typename T>
__device__ __forceinline__ T* as_device(T* p) {
  __attribute__((address_space(1))) T* g = (__attribute__((address_space(1))) T*)p;
  asm volatile("" : "+s"(g));
  return (T*)g;
}
// Uploads every problem's solver head from its mailbox and prepares its first transform (k_lm
// MODE_PREPARE for all problems of a batch in one launch).
__global__ __launch_bounds__(kLmBlock) void k_lm_prepare_batch(const SingleJob* __restrict__ jobs) {
  const SingleJob& J = jobs[blockIdx.x];
  lm_step(J.G, const_cast<BlockXform*>(J.xf), J.partials, nullptr, MODE_PREPARE, J.box, J.up_words);
}
__global__ __launch_bounds__(kEvalThreads) void k_lm_single_batch(const SingleJob* __restrict__ jobs) {
  const SingleJob& J = jobs[blockIdx.x];
  if (J.G->h.done) return;
  constexpr size_t kTail = ((kEvalThreads / kAcc) + 1) * kAcc * sizeof(double) + sizeof(LmHead) + 36 * sizeof(double);
  __shared__ __align__(16) unsigned char smem[kTail];
  lm_step_single(reinterpret_cast<double*>(smem), as_device(J.G), as_device(const_cast<BlockXform*>(J.xf)), as_device(J.partials), J.num_wg);
}

// All residual blocks of a problem in ONE launch per LM iteration: workgroup -> (block, local
// workgroup) through the device block table (entries of one kind: per-scan blocks, or blocks with a
// ratio per return; a problem holding both kinds launches the kernel once per kind). Every block's
// first workgroup is a multiple of 8, so that blockIdx % 8 -- the XCD -- is also the local index % 8
// (xcd_chunk). Workgroups behind the last block evaluate the odometry / IMU blocks (one wavefront
// each): their 15 us of serial Jet arithmetic run in the shadow of the TSDF blocks instead of in a
// launch of their own. No solver tail: k_lm follows as its own launch.
#ifndef HG_WINDOW_WAVES
// Workgroups per CU the per-scan body is compiled for (register budget 512 / waves). Measured on the
// window of nine 100k-point scans: 4 (128 VGPRs, 31 of them spilled) 34.9 us per launch and 14.9 MB of
// scratch writes; 3 (no spills) 35.5 us and 0.2 MB of writes -- the spill traffic buys 2 %, not taken.
#define HG_WINDOW_WAVES 3
#endif
template <bool UNWARP>
__device__ __forceinline__ void window_eval(
    const EvalBlock* __restrict__ blocks, int num_eval, unsigned tiles, double* __restrict__ residuals,
    LmState* G, const BlockXform* __restrict__ xf_all, double* __restrict__ partials_all, SmallOut* small_out,
    unsigned tsdf_wg, unsigned num_small, unsigned* tickets, double* __restrict__ loc_all, unsigned bx) {
  if (G->h.done) return;
  if (bx >= tsdf_wg) {
#ifndef HG_NO_SMALL_FOLD
    if (bx - tsdf_wg < num_small && threadIdx.x < kWave)
      small_block_eval(G, static_cast<int>(bx - tsdf_wg), small_out, residuals);
#endif
    return;
  }
  constexpr size_t kTile = (kBatchThreads / kWave) * (kWave * (UNWARP ? 16 : 8) + (UNWARP ? 256 : 64)) * sizeof(double);
  __shared__ __align__(16) unsigned char smem[kTile];
  // block of this workgroup: every lane reads one table entry's first workgroup, one ballot
  int b;
  {
    const int lane = threadIdx.x % kWave;
    int count = 0;
    for (int base = 0; base < num_eval; base += kWave) {  // (one round in the plain build: kMaxBlocks <= 64)
      const bool le = base + lane < num_eval && blocks[base + lane].wg_begin <= bx;
      count += __popcll(__ballot(le));
    }
    b = __builtin_amdgcn_readfirstlane(count - 1);  // wg_begin ascends; entry 0 starts at 0
  }
  const EvalBlock& eb = blocks[b];
  const unsigned wg = bx - eb.wg_begin;
  if (wg >= eb.num_wg) return;  // padding up to the next block's multiple of 8
  double* res = residuals ? residuals + eb.row_offset : nullptr;
  if (UNWARP)
    window_body_unwarp(eb, G->h.cand[eb.pose_a], G->h.cand[eb.pose_b], partials_all + eb.partial_offset, res, wg,
                       tiles, smem);
  else
    window_body_plain(eb, xf_all + eb.index, partials_all + eb.partial_offset, res, wg, tiles, smem);
  // (the barrier at the head of the tail also ends this workgroup's use of the tiles in smem)
  window_block_tail<UNWARP>(eb, partials_all + eb.partial_offset, tickets + 8 + eb.index, xf_all + eb.index,
                            loc_all + static_cast<size_t>(eb.index) * kLoc, reinterpret_cast<double*>(smem));
}

template <bool UNWARP>
__global__ __launch_bounds__(kBatchThreads, UNWARP ? 2 : HG_WINDOW_WAVES) void k_window_residuals(
    const EvalBlock* __restrict__ blocks, int num_eval, unsigned tiles, double* __restrict__ residuals,
    LmState* G, const BlockXform* __restrict__ xf_all, double* __restrict__ partials_all, SmallOut* small_out,
    unsigned tsdf_wg, unsigned* tickets, double* __restrict__ loc_all) {
  window_eval<UNWARP>(blocks, num_eval, tiles, residuals, G, xf_all, partials_all, small_out, tsdf_wg, gridDim.x - tsdf_wg,
                      tickets, loc_all, blockIdx.x);
}

// Several INDEPENDENT general problems (sliding windows of different submaps) per launch: grid row = problem,
// every problem with its own block table, state, partials, tickets and local systems -- the arithmetic of
// k_window_residuals + k_lm per problem, the launches shared (one registration chain of a window leaves most
// of the chip idle, and k_lm is a single workgroup).
struct WindowJob {
  const EvalBlock* blocks;   // per-scan blocks first, then the blocks with a ratio per return
  int num_plain, num_unwarp;
  unsigned wg_plain, wg_unwarp, num_small, tiles;
  LmState* G;
  BlockXform* xf;
  double* partials;
  SmallOut* small_out;
  unsigned* tickets;
  double* loc;
  const PinBox* box;
  unsigned up_words;
  unsigned stage;  // lm_stage_counts(): what lm_step's prefetch needs to know, or 0
};
#ifndef HG_WINDOW_JOBS_WAVES
// (the same body lands at 125 VGPRs in k_window_residuals and at 164 here under the budget of three workgroups per CU:
// with the budget of four -- 128 VGPRs, a few spills -- eight windows per call run 2417 -> 2496 scans/s)
#define HG_WINDOW_JOBS_WAVES 4
#endif
template <bool UNWARP>
__global__ __launch_bounds__(kBatchThreads, UNWARP ? 2 : HG_WINDOW_JOBS_WAVES) void k_window_residuals_jobs(
    const WindowJob* __restrict__ jobs) {
  const WindowJob& J = jobs[blockIdx.y];
  // the odometry / IMU blocks ride on the per-scan launch
  const unsigned tsdf_wg = UNWARP ? J.wg_unwarp : J.wg_plain, small = UNWARP ? 0u : J.num_small;
  if (blockIdx.x >= tsdf_wg + small) return;
  window_eval<UNWARP>(as_device(J.blocks) + (UNWARP ? J.num_plain : 0), UNWARP ? J.num_unwarp : J.num_plain, J.tiles, nullptr,
                      as_device(J.G), as_device(J.xf), as_device(J.partials), as_device(J.small_out), tsdf_wg, small,
                      as_device(J.tickets), as_device(J.loc), blockIdx.x);
}
__global__ __launch_bounds__(kLmBlock) void k_lm_jobs(const WindowJob* __restrict__ jobs, int mode) {
  const WindowJob& J = jobs[blockIdx.x];
  if (mode == MODE_STEP && J.stage == 0u && J.G->h.done) return;  // (with the counts known the step itself looks, behind its prefetch)
  lm_step(J.G, J.xf, J.loc, J.small_out, mode, mode == MODE_PREPARE ? J.box : nullptr, J.up_words, J.stage);
}

__global__ __launch_bounds__(kLmBlock) void k_lm(LmState* G, BlockXform* xf, const double* partials,
                                                 const SmallOut* small_out, int mode,
                                                 const PinBox* host_up, unsigned up_words, unsigned stage) {
  if (mode == MODE_STEP && stage == 0u && G->h.done) return;
  lm_step(G, xf, partials, small_out, mode, host_up, up_words, stage);
}

HG_CAP_NS_CLOSE

using namespace hg;

// what lm_step's prefetch needs to know before the head is there: TSDF blocks | odometry / IMU blocks << 8 | band
// entries << 16 (0 = nothing: the step finds out itself)
static unsigned lm_stage_counts(const LmHead& S) {
#ifdef HG_BIG
  (void)S;
  return 0u;
#else
  if (S.num_blocks < 0 || S.num_blocks > kMaxBlocks || S.num_small < 0 || S.num_small > kMaxSmall) return 0u;
  const unsigned nW = static_cast<unsigned>(S.ncols) * static_cast<unsigned>(S.bw + 1);
  if (S.num_blocks > 255 || S.num_small > 255 || nW > 0xFFFFu) return 0u;
  return static_cast<unsigned>(S.num_blocks) | (static_cast<unsigned>(S.num_small) << 8) | (nW << 16);
#endif
}

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
    const double* d_factor = nullptr;  // per-return interpolation factors (unwarped block)
    unsigned width = 0;  // returns per column when the cloud is a structured scan (hg_problem_set_block_width)
  };
  std::vector<Block> blocks;
  std::vector<std::array<double, 7>> poses;
  std::vector<int> constant;
  std::vector<std::array<double, 3>> velocity;
  std::vector<int> vfree;  // 1: velocity set and not constant
  std::vector<SmallBlockDev> small;
  SmallOut* d_small = nullptr;
  double* d_loc = nullptr;  // per TSDF block: its local normal equations (kLoc doubles), k_window_residuals -> k_lm
  // device state
  LmState* d_state = nullptr;
  unsigned* d_ticket = nullptr;
  hg::DeviceBuffer bcast;      // persistent solve: the candidate + done flag workgroup 0 hands to the others (8 granule pairs)
  bool persist_last = false;   // the solve in flight is a persistent one (its evaluations are one launch)
  hg::DeviceBuffer granules;   // tagged partial sums of the single-pose chain (store_partial_tagged)
  unsigned epoch = 0;          // launches of that chain so far
  BlockXform* d_xf = nullptr;
  DeviceBuffer partials, residuals;
  DeviceBuffer part_xyz, part_flags, part_counts;  // level partition of the block's cloud (batched single-pose solves)
  bool solve_pending = false;
  PyramidView* d_pv = nullptr;  // per block: its pyramid in device memory (PyramidView::self_mem)
  PyramidView* h_pv = nullptr;  // pinned staging = what d_pv holds (re-uploaded only when it changes)
  EvalBlock* d_eval = nullptr;  // block table of the window pass: per-scan blocks first, then the unwarped ones
  EvalBlock* h_eval = nullptr;  // pinned staging
  int num_eval = 0;             // active blocks in the table
  int num_plain = 0, num_unwarp = 0;  // of which per-scan blocks / blocks with a ratio per return
  unsigned wg_plain = 0, wg_unwarp = 0;  // workgroups of the two launches (blocks start at multiples of 8)
  unsigned tiles = 1;           // 256-return tiles per workgroup of the window pass
  unsigned cap_plain = 0, cap_unwarp = 0;  // workgroups of k_window_residuals the device holds at once
  int batch_share = 1;          // problems that share the launches of this solve (hg_problem_solve_batch): each
                                // sizes its window pass for its share of the chip
  PinBox* h_box = nullptr;   // mapped pinned mailbox: upload source and result sink
  PinBox* d_box = nullptr;   // its device address
  unsigned long long seq = 0;
  unsigned up_words = 0;
  bool prof_grouped = false;  // the residual launches of this solve share one event pair
  // the last launches of a general solve, held back until the solve is seen to need them (solve_settle)
  int lazy_left = 0;
  hipEvent_t ev_lazy = nullptr;
  hipEvent_t ev_pv = nullptr;  // behind the last copy out of h_pv (the pinned staging of the blocks' pyramids)
  bool pv_pending = false;
  int single_threads = 0;      // > 0: single-pose registration step with this workgroup size
#ifndef HG_BIG
  // A problem beyond the plain build's limits (kMaxPoses / kMaxBlocks / kMaxSmall, band capacity) is
  // mirrored into a problem of the big build, which then evaluates and solves it (promote()). The vectors
  // above stay the record of what the caller added; poses and velocities are copied back after a solve.
  hg_problem_big* big = nullptr;
  bool promoted = false;
#endif
  LmState h_state;            // host copy
};

namespace {

bool block_active(const hg_problem* p, const hg_problem::Block& b) {
  if (b.n == 0) return false;
  if (!p->constant[b.pose_a]) return true;
  return b.pose_b >= 0 && !p->constant[b.pose_b];
}

// Fills h_state's static part and uploads it. cand = x = current poses.
int upload_state(hg_problem* p, const hg_solver_opts* opts) {
  LmState& ST = p->h_state;
  std::memset(&ST, 0, sizeof(ST));
  LmHead& S = ST.h;
  S.num_poses = static_cast<int>(p->poses.size());
  S.num_blocks = static_cast<int>(p->blocks.size());
  int col = 0;
  for (int i = 0; i < S.num_poses; ++i) {
    for (int k = 0; k < 7; ++k) S.x[i][k] = S.cand[i][k] = p->poses[i][k];
    for (int k = 0; k < 3; ++k) S.x[i][7 + k] = S.cand[i][7 + k] = p->velocity[i][k];
    S.constant[i] = p->constant[i];
    S.col[i] = p->constant[i] ? -1 : col;
    if (!p->constant[i]) col += 6;
    S.vfree[i] = p->vfree[i];
    S.vcol[i] = p->vfree[i] ? col : -1;
    if (p->vfree[i]) col += 3;
  }
  S.ncols = col;
  if (opts) S.opt = *opts; else hg_solver_default_opts(&S.opt);
  S.radius = S.opt.initial_trust_region_radius;
  S.decrease_factor = 2.0;
  S.phase = PHASE_INIT;
  // register-resident LM tail: one free pose without velocity, one per-scan block, nothing else
  // (bw and the activity of the block are checked below, once they are known)
  p->single_threads = 0;
#ifndef HG_BIG  // (a problem of that shape never reaches the big build)
  if (opts && S.num_poses == 1 && !p->constant[0] && !p->vfree[0] && S.num_blocks == 1 && p->small.empty() &&
      !p->blocks[0].d_factor && p->blocks[0].pose_b < 0 && p->blocks[0].n > 0 && p->ctx->opt(OPT_LM_GENERAL) == 0) {
    p->single_threads = kEvalThreads;  // 256-thread workgroups were measured 19 % slower per step
  }
#endif
  // Window pass: a workgroup walks `tiles` tiles of 256 returns, chosen so that all workgroups of an
  // iteration are resident at once (one round on the chip, no tail round).
  p->tiles = 1;
  if (!p->single_threads) {
    if (!p->cap_plain) {
      int cus = 0, occ_p = 0, occ_u = 0;
      HG_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p->ctx->device));
      HG_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_p, k_window_residuals<false>, kBatchThreads, 0));
      HG_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_u, k_window_residuals<true>, kBatchThreads, 0));
      p->cap_plain = static_cast<unsigned>(std::max(1, cus) * std::max(1, occ_p));
      p->cap_unwarp = static_cast<unsigned>(std::max(1, cus) * std::max(1, occ_u));
      if (p->ctx->opt(OPT_WINDOW_CAPACITY) > 0) p->cap_plain = p->cap_unwarp = static_cast<unsigned>(p->ctx->opt(OPT_WINDOW_CAPACITY));
    }
    unsigned long long tiles_plain = 0, tiles_unwarp = 0;
    for (const hg_problem::Block& hb : p->blocks)
      if (block_active(p, hb)) (hb.d_factor ? tiles_unwarp : tiles_plain) += (hb.n + kBatchThreads - 1) / kBatchThreads;
    // every block rounds its workgroups up (and to a multiple of 8 in the grid): leave room for that
    const unsigned slack = 8u * static_cast<unsigned>(p->blocks.size());
    const unsigned share = static_cast<unsigned>(std::max(1, p->batch_share));
    const unsigned cap_p = std::max(1u, p->cap_plain / share), cap_u = std::max(1u, p->cap_unwarp / share);
    const unsigned long long tp = (tiles_plain + std::max(1u, cap_p - std::min(cap_p - 1u, slack)) - 1) /
                                  std::max(1u, cap_p - std::min(cap_p - 1u, slack));
    const unsigned long long tu = (tiles_unwarp + std::max(1u, cap_u - std::min(cap_u - 1u, slack)) - 1) /
                                  std::max(1u, cap_u - std::min(cap_u - 1u, slack));
    p->tiles = static_cast<unsigned>(std::min<unsigned long long>(kMaxTiles, std::max<unsigned long long>(1, std::max(tp, tu))));
    if (p->ctx->opt(OPT_WINDOW_TILES) > 0) p->tiles = static_cast<unsigned>(std::min<long long>(kMaxTiles, p->ctx->opt(OPT_WINDOW_TILES)));
  }
  const unsigned eval_threads = p->single_threads ? static_cast<unsigned>(p->single_threads) : kBatchThreads * p->tiles;
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
    bi.acc = hb.d_factor ? kAccU : kAcc;
    bi.num_wg = bi.active ? (bi.n + eval_threads - 1) / eval_threads : 0;
    bi.partial_offset = wg_off;
    bi.row_offset = row;
    wg_off += bi.num_wg * bi.acc;
    if (bi.active) row += bi.n;
  }
  S.num_small = static_cast<int>(p->small.size());
  for (int b = 0; b < S.num_small; ++b) {
    SmallBlockDev sb = p->small[b];
    const bool free_pose = !p->constant[sb.a] || !p->constant[sb.b];
    const bool free_vel = sb.type == 2 && (p->vfree[sb.a] || p->vfree[sb.b]);
    sb.active = (free_pose || free_vel) ? 1 : 0;
    sb.row_offset = row;
    if (sb.active) row += (sb.type == 1 ? 6u : 9u);
    S.small[b] = sb;
  }
  // half bandwidth: the largest column distance any active block couples
  int bw = 0;
  auto couple = [&](std::initializer_list<int> ids, bool with_velocity) {
    int lo = 1 << 30, hi = -1;
    for (int i : ids) {
      if (i < 0) continue;
      if (!S.constant[i]) { lo = std::min(lo, S.col[i]); hi = std::max(hi, S.col[i] + 5); }
      if (with_velocity && S.vfree[i]) { lo = std::min(lo, S.vcol[i]); hi = std::max(hi, S.vcol[i] + 2); }
    }
    if (hi >= 0) bw = std::max(bw, hi - lo);
  };
  for (int i = 0; i < S.num_poses; ++i) couple({i}, false);  // keeps the diagonal blocks in band
  for (int b = 0; b < S.num_blocks; ++b)
    if (S.blocks[b].active) couple({S.blocks[b].pose_a, S.blocks[b].pose_b}, false);
  for (int b = 0; b < S.num_small; ++b)
    if (S.small[b].active) couple({S.small[b].a, S.small[b].b}, S.small[b].type == 2);
  S.bw = bw;
  {
    // block-tridiagonal partition: one group per control point with free columns (its pose and / or
    // velocity columns are contiguous by construction); usable when every active block couples a
    // group with itself or its neighbour. HG_LM_BAND=1 keeps the band factorisation (diagnostics).
    int group_of[kMaxPoses];
    int groups = 0;
    for (int i = 0; i < S.num_poses; ++i) {
      group_of[i] = -1;
      const int first = !S.constant[i] ? S.col[i] : (S.vfree[i] ? S.vcol[i] : -1);
      if (first < 0) continue;
      group_of[i] = groups;
      S.btd_start[groups] = first;
      S.btd_size[groups] = (!S.constant[i] ? 6 : 0) + (S.vfree[i] ? 3 : 0);
      ++groups;
    }
    bool ok = groups >= 2 && p->ctx->opt(OPT_LM_BAND) == 0;
    auto near = [&](int a, int b) {
      if (a < 0 || b < 0 || group_of[a] < 0 || group_of[b] < 0) return true;  // a constant end couples nothing
      return std::abs(group_of[a] - group_of[b]) <= 1;
    };
    for (int b = 0; b < S.num_blocks && ok; ++b)
      if (S.blocks[b].active) ok = near(S.blocks[b].pose_a, S.blocks[b].pose_b);
    for (int b = 0; b < S.num_small && ok; ++b)
      if (S.small[b].active) ok = near(S.small[b].a, S.small[b].b);
    S.btd_groups = ok ? groups : 0;
    S.btd_uniform = 0;
    if (ok && S.btd_start[0] == 0) {
      const int mb = S.btd_size[0];
      bool uni = (mb == 6 || mb == 9) && bw >= 2 * mb - 1;
      for (int g = 1; g < groups; ++g) uni = uni && S.btd_size[g] == mb;
      if (uni && p->ctx->opt(OPT_LM_BTD_GENERIC) == 0) S.btd_uniform = mb;
      // 2: twisted factorisation from both ends of the chain (up to nine groups: the default shape), 1: cyclic reduction
      // over the workgroup (long chains, or HG_LM_BTD_CR=1), 0: the chain in one wavefront (HG_LM_BTD_CHAIN=1)
      S.btd_cr = p->ctx->opt(OPT_LM_BTD_CHAIN) != 0 ? 0 : (p->ctx->opt(OPT_LM_BTD_CR) != 0 || groups > 9) ? 1 : 2;
    }
    if (!ok) {
      std::memset(S.btd_start, 0, sizeof(S.btd_start));
      std::memset(S.btd_size, 0, sizeof(S.btd_size));
    }
  }
  if (static_cast<long long>(S.ncols) * (bw + 1) > kHCap) {
    set_last_error("normal equations exceed the band capacity (n * (bandwidth + 1) > " +
                   std::to_string(kHCap) + "): blocks must couple nearby control points");
    return HG_ERR_CAPACITY;
  }
  int rc = p->partials.reserve(static_cast<size_t>(std::max(1u, wg_off)) * sizeof(double));
  if (rc != HG_OK) return rc;
  // the pinned buffer may still be the source of the previous (finished) upload: solves are
  // synchronised by their fetch before the next upload
  // block table of the window pass: per-scan blocks first, then the blocks with a ratio per return
  p->num_eval = p->num_plain = p->num_unwarp = 0;
  p->wg_plain = p->wg_unwarp = 0;
  bool pv_copied = false;
  for (int kind = 0; kind < 2; ++kind) {
    unsigned wg_next = 0;
    for (int b = 0; b < S.num_blocks; ++b) {
      const BlockInfo& bi = S.blocks[b];
      const hg_problem::Block& hb = p->blocks[b];
      if (!bi.active || (hb.d_factor != nullptr) != (kind == 1)) continue;
      EvalBlock& eb = p->h_eval[p->num_eval];
      std::memset(&eb, 0, sizeof(eb));
      {
        // the block's pyramid, also kept in device memory (uploaded when it differs from what is there)
        PyramidView pv;
        std::memset(&pv, 0, sizeof(pv));
        pv.levels = static_cast<int>(hb.pyramid.size());
        pv.multi_res = hb.multi_res;
        for (int l = 0; l < pv.levels; ++l) pv.level[l] = hb.pyramid[l]->view;
        pv.self_mem = p->d_pv + b;
        if (std::memcmp(&p->h_pv[b], &pv, sizeof(pv)) != 0) {
          // the staging slot may still feed THIS problem's earlier copy: wait for that copy alone (an event behind it),
          // not for the stream -- a constraint search matches every batch against other submaps, and a stream
          // synchronisation here drained batch k before batch k + 1 could be enqueued: the previous solve of
          // this problem has been fetched, so its copy is long done and the wait returns at once
          if (p->pv_pending) {
            HG_HIP_CHECK(hipEventSynchronize(p->ev_pv));
            p->pv_pending = false;
          }
          p->h_pv[b] = pv;
          HG_HIP_CHECK(hipMemcpyAsync(p->d_pv + b, &p->h_pv[b], sizeof(pv), hipMemcpyHostToDevice, p->ctx->stream));
          pv_copied = true;
        }
        eb.pv = pv;
      }
      eb.xyz = hb.d_xyz;
      eb.factor = hb.d_factor;
      eb.scaling = bi.scaling;
      eb.n = bi.n;
      eb.wg_begin = wg_next;
      eb.num_wg = bi.num_wg;
      wg_next += (bi.num_wg + 7u) & ~7u;
      eb.partial_offset = bi.partial_offset;
      eb.row_offset = bi.row_offset;
      eb.pose_a = bi.pose_a;
      eb.pose_b = bi.pose_b;
      eb.index = b;
      eb.width = hb.width;
      ++p->num_eval;
      ++(kind ? p->num_unwarp : p->num_plain);
    }
    (kind ? p->wg_unwarp : p->wg_plain) = wg_next;
  }
  if (pv_copied) {
    if (!p->ev_pv) HG_HIP_CHECK(hipEventCreateWithFlags(&p->ev_pv, hipEventDisableTiming));
    HG_HIP_CHECK(hipEventRecord(p->ev_pv, p->ctx->stream));
    p->pv_pending = true;
  }
  if (p->num_eval >= 1 && !p->single_threads)
    HG_HIP_CHECK(hipMemcpyAsync(p->d_eval, p->h_eval, sizeof(EvalBlock) * p->num_eval,
                                hipMemcpyHostToDevice, p->ctx->stream));
  S.box = p->d_box;
  S.seq = ++p->seq;
  // zero-copy upload: k_lm MODE_PREPARE reads the head from the mailbox. The previous solve's
  // fetch has seen its result, so the device no longer reads or writes the mailbox.
  std::memcpy(&p->h_box->h, &ST.h, sizeof(LmHead));
  {
    // compact form: the non-zero words and their indices
    const unsigned long long* w = reinterpret_cast<const unsigned long long*>(&ST.h);
    unsigned n = 0;
    for (unsigned i = 0; i < sizeof(LmHead) / 8; ++i)
      if (w[i] != 0ull) {
        p->h_box->up_idx[n] = i;
        p->h_box->up_val[n] = w[i];
        ++n;
      }
    p->up_words = n;
  }
  return HG_OK;
}

#ifndef HG_BIG
// May the single-pose solve of `p` run as ONE persistent launch (k_tsdf_residuals_single_persist)? Its workgroups wait
// for each other inside the launch, so all of them must be resident at once: the kernel's occupancy times the CUs
// covers the grid, and no other context of this process competes for the CUs (another process may: the kernel's
// waits are bounded, and a solve that timed out switches the form off for the context).
bool persist_ok(hg_problem* p, unsigned num_wg) {
  hg_ctx* c = p->ctx;
  if (c->opt(OPT_PERSISTENT_SOLVE) == 0 || c->opt(OPT_TICKET_HANDOVER) != 0 || c->persist_failed) return false;
  if (live_contexts() != 1) return false;
  if (c->persist_blocks_per_cu < 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_tsdf_residuals_single_persist<kEvalThreads>, kEvalThreads, 0) != hipSuccess) nb = 0;
    c->persist_blocks_per_cu = nb;
  }
  return c->persist_blocks_per_cu >= 1 && static_cast<unsigned long long>(c->persist_blocks_per_cu) * static_cast<unsigned>(std::max(0, c->num_cus)) >= num_wg;
}

// The whole solve of the single-pose registration shape as one launch of `evals` evaluations.
int launch_persistent(hg_problem* p, unsigned evals) {
  hipStream_t s = p->ctx->stream;
  const LmHead& S = p->h_state.h;
  const BlockInfo& bi = S.blocks[0];
  const hg_problem::Block& hb = p->blocks[0];
  const PyramidView& pv = p->h_pv[0];
  const size_t need = static_cast<size_t>(bi.num_wg) * kAcc * 2 * sizeof(unsigned long long);
  int rc;
  if (p->granules.bytes < need) {
    if ((rc = p->granules.reserve(need)) != HG_OK) return rc;
    HG_HIP_CHECK(hipMemsetAsync(p->granules.ptr, 0, p->granules.bytes, s));  // (no tag equals an epoch: epochs start at 1)
  }
  if (!p->bcast.ptr) {
    if ((rc = p->bcast.reserve(16 * kBcastReplicas * sizeof(unsigned long long))) != HG_OK) return rc;
    HG_HIP_CHECK(hipMemsetAsync(p->bcast.ptr, 0, p->bcast.bytes, s));
  }
  if (p->epoch > 0xFFFFFFFFu - 2u * (evals + 2u)) p->epoch = 0;  // (tags of four billion evaluations ago match nothing that is left)
  const unsigned epoch0 = p->epoch + 1u;
  p->epoch += evals + 1u;  // epochs epoch0 .. epoch0 + evals - 1 tag the sums, + 1 each the hand-back
  FirstUpload up;
  std::memcpy(up.pose, S.cand[0], sizeof(up.pose));
  up.box = p->d_box;
  up.up_words = p->up_words;
  up.pad = 0;
  hipLaunchKernelGGL(k_tsdf_residuals_single_persist<kEvalThreads>, dim3(bi.num_wg), dim3(kEvalThreads), 0, s, pv, hb.d_xyz,
                     bi.n, hb.width, bi.scaling, p->d_xf, p->granules.as<double>(), p->d_state, up, epoch0, evals,
                     p->bcast.as<unsigned long long>());
  HG_HIP_CHECK(hipGetLastError());
  return HG_OK;
}
#endif

// One evaluation of every residual block at the candidate. with_lm: followed by one LM step (the
// single-pose registration shape runs it in the tail of its residual launch, every other problem as a
// k_lm launch behind the window pass).
int launch_eval(hg_problem* p, double* d_residuals, bool with_lm, bool first = false) {
  hipStream_t s = p->ctx->stream;
  const LmHead& S = p->h_state.h;
  if (with_lm && p->single_threads && S.ncols == 6 && S.bw == 5 && S.blocks[0].active) {
    const BlockInfo& bi = S.blocks[0];
    const hg_problem::Block& hb = p->blocks[0];
    const PyramidView& pv = p->h_pv[0];
    (void)hb;
    ProfScope ps(p->ctx, HG_K_RESIDUALS, bi.n, 1, !p->prof_grouped);
    // partial sums as tagged granules (HG_TICKET_HANDOVER=1: the acknowledged stores + ticket of rounds 1-4)
    const bool tagged = p->ctx->opt(OPT_TICKET_HANDOVER) == 0;
    double* sums = p->partials.as<double>();
    unsigned epoch = 0;
    if (tagged) {
      const size_t need = static_cast<size_t>(bi.num_wg) * kAcc * 2 * sizeof(unsigned long long);
      if (p->granules.bytes < need) {
        const int rc = p->granules.reserve(need);
        if (rc != HG_OK) return rc;
        HG_HIP_CHECK(hipMemsetAsync(p->granules.ptr, 0, p->granules.bytes, s));  // (no tag equals an epoch: epochs start at 1)
      }
      sums = p->granules.as<double>();
      epoch = ++p->epoch;
      if (epoch == 0u) epoch = ++p->epoch;
    }
    if (first) {
      FirstUpload up;
      std::memcpy(up.pose, S.cand[0], sizeof(up.pose));
      up.box = p->d_box;
      up.up_words = p->up_words;
      up.pad = 0;
      hipLaunchKernelGGL(k_tsdf_residuals_single_first<kEvalThreads>, dim3(bi.num_wg), dim3(kEvalThreads), 0, s, pv,
                         hb.d_xyz, bi.n, hb.width, bi.scaling, p->d_xf, sums, p->d_state, p->d_ticket, up, epoch);
    } else {
      hipLaunchKernelGGL(k_tsdf_residuals_single<kEvalThreads>, dim3(bi.num_wg), dim3(kEvalThreads), 0, s, pv,
                         hb.d_xyz, bi.n, hb.width, bi.scaling, p->d_xf, sums, p->d_state, p->d_ticket, epoch);
    }
    HG_HIP_CHECK(hipGetLastError());
    return HG_OK;
  }
  // window pass: all blocks of a kind in one launch; the odometry / IMU blocks ride on the first one
  unsigned small_left = static_cast<unsigned>(S.num_small);
#ifdef HG_NO_SMALL_FOLD
  if (small_left > 0) {
    hipLaunchKernelGGL(k_small_blocks, dim3(small_left), dim3(kWave), 0, s, p->d_state, p->d_small, d_residuals);
    HG_HIP_CHECK(hipGetLastError());
    small_left = 0;
  }
#endif
  unsigned long long units_plain = 0, units_unwarp = 0;
  for (int b = 0; b < S.num_blocks; ++b)
    if (S.blocks[b].active) (p->blocks[b].d_factor ? units_unwarp : units_plain) += S.blocks[b].n;
  if (p->num_plain > 0) {
    ProfScope ps(p->ctx, HG_K_RESIDUALS, units_plain);
    hipLaunchKernelGGL(k_window_residuals<false>, dim3(p->wg_plain + small_left), dim3(kBatchThreads), 0, s,
                       static_cast<const EvalBlock*>(p->d_eval), p->num_plain, p->tiles, d_residuals, p->d_state,
                       static_cast<const BlockXform*>(p->d_xf), p->partials.as<double>(), p->d_small, p->wg_plain,
                       p->d_ticket, p->d_loc);
    HG_HIP_CHECK(hipGetLastError());
    small_left = 0;
  }
  if (p->num_unwarp > 0) {
    ProfScope ps(p->ctx, HG_K_RESIDUALS, units_unwarp);
    hipLaunchKernelGGL(k_window_residuals<true>, dim3(p->wg_unwarp + small_left), dim3(kBatchThreads), 0, s,
                       static_cast<const EvalBlock*>(p->d_eval + p->num_plain), p->num_unwarp, p->tiles, d_residuals,
                       p->d_state, static_cast<const BlockXform*>(p->d_xf), p->partials.as<double>(), p->d_small,
                       p->wg_unwarp, p->d_ticket, p->d_loc);
    HG_HIP_CHECK(hipGetLastError());
    small_left = 0;
  }
  if (small_left > 0) {  // a problem of odometry / IMU blocks only
    hipLaunchKernelGGL(k_small_blocks, dim3(small_left), dim3(kWave), 0, s, p->d_state, p->d_small, d_residuals);
    HG_HIP_CHECK(hipGetLastError());
  }
  if (with_lm) {
    ProfScope ps(p->ctx, HG_K_LM, 1);
    hipLaunchKernelGGL(k_lm, dim3(1), dim3(kLmBlock), 0, s, p->d_state, p->d_xf, static_cast<const double*>(p->d_loc),
                       p->d_small, MODE_STEP, static_cast<const PinBox*>(nullptr), 0u, lm_stage_counts(p->h_state.h));
    HG_HIP_CHECK(hipGetLastError());
  }
  return HG_OK;
}

}  // namespace

#ifdef HG_BIG
void hg::big_orphan(hg_problem* p) { p->ctx = nullptr; }
const double* hg::big_device_poses(hg_problem* p, int* stride) {
  *stride = kState;
  return (p->h_state.h.ncols == 0) ? nullptr : &p->d_state->h.x[0][0];
}
#else
void hg::orphan_problem(hg_problem* p) {
  p->ctx = nullptr;
  if (p->big) big_orphan(p->big);
}

namespace {
// Mirrors the problem as recorded so far into its big twin (created on first use, kept across resets).
// TSDF blocks are handed over as device pointers: the uploads of host-memory blocks stay owned here.
int promote(hg_problem* p) {
  if (p->promoted) return HG_OK;
  if (!p->ctx) return HG_ERR_INVALID;
  int rc;
  if (!p->big && (rc = hg_problem_create_big(p->ctx, &p->big)) != HG_OK) return rc;
  if ((rc = hg_problem_reset_big(p->big)) != HG_OK) return rc;
  for (size_t i = 0; i < p->poses.size(); ++i) {
    if ((rc = hg_problem_add_pose_big(p->big, p->poses[i].data(), p->constant[i])) < 0) return rc;
    if ((rc = hg_problem_set_velocity_big(p->big, static_cast<int>(i), p->velocity[i].data(), p->vfree[i] ? 0 : 1)) != HG_OK) return rc;
  }
  for (const SmallBlockDev& sb : p->small) {
    rc = sb.type == 1 ? hg_problem_add_odometry_block_big(p->big, sb.a, sb.b, sb.w[0], sb.w[1], sb.delta)
                      : hg_problem_add_imu_block_big(p->big, sb.a, sb.b, sb.w[0], sb.w[1], sb.w[2], sb.dt, sb.delta + 3);
    if (rc < 0) return rc;
  }
  for (const hg_problem::Block& b : p->blocks) {
    rc = b.d_factor ? hg_problem_add_unwarped_block_big(p->big, b.d_xyz, b.d_factor, b.n, HG_DEVICE, b.pyramid.data(),
                                                        static_cast<int>(b.pyramid.size()), b.multi_res, b.scaling, b.pose_a, b.pose_b)
                    : hg_problem_add_block_big(p->big, b.d_xyz, b.n, HG_DEVICE, b.pyramid.data(),
                                               static_cast<int>(b.pyramid.size()), b.multi_res, b.scaling, b.pose_a, b.pose_b, b.factor);
    if (rc < 0) return rc;
    if (b.width && (rc = hg_problem_set_block_width_big(p->big, rc, b.width)) != HG_OK) return rc;
  }
  p->promoted = true;
  return HG_OK;
}
// The problem's poses and velocities as the big twin holds them (after its fetch).
void pull_from_big(hg_problem* p) {
  for (size_t i = 0; i < p->poses.size(); ++i) {
    (void)hg_problem_get_pose_big(p->big, static_cast<int>(i), p->poses[i].data());
    (void)hg_problem_get_velocity_big(p->big, static_cast<int>(i), p->velocity[i].data());
  }
}
}  // namespace
#endif

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
#ifndef HG_BIG
  ctx->live_problems.push_back(p);  // (a big twin belongs to its plain problem, which orphans it)
#endif
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&p->d_state), sizeof(LmState));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_xf), sizeof(BlockXform) * kMaxBlocks);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_ticket), sizeof(unsigned) * (64 + kMaxBlocks));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_small), sizeof(SmallOut) * kMaxSmall);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_loc), sizeof(double) * kLoc * kMaxBlocks);
  if (e == hipSuccess) e = hipMemset(p->d_loc, 0, sizeof(double) * kLoc * kMaxBlocks);
  if (e == hipSuccess)
    e = hipHostMalloc(reinterpret_cast<void**>(&p->h_box), sizeof(PinBox), hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) {
    std::memset(p->h_box, 0, sizeof(PinBox));
    e = hipHostGetDevicePointer(reinterpret_cast<void**>(&p->d_box), p->h_box, 0);
  }
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_eval), sizeof(EvalBlock) * kMaxBlocks);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_pv), sizeof(PyramidView) * kMaxBlocks);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&p->h_pv), sizeof(PyramidView) * kMaxBlocks);
  if (e == hipSuccess) std::memset(p->h_pv, 0, sizeof(PyramidView) * kMaxBlocks);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&p->h_eval), sizeof(EvalBlock) * kMaxBlocks);
  if (e == hipSuccess) e = hipMemset(p->d_ticket, 0, sizeof(unsigned) * (64 + kMaxBlocks));
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
  if (p->ctx) {
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
#ifndef HG_BIG
    auto& live = p->ctx->live_problems;
    live.erase(std::remove(live.begin(), live.end(), p), live.end());
#endif
  }
#ifndef HG_BIG
  if (p->big) (void)hg_problem_destroy_big(p->big);
#endif
  for (auto& b : p->blocks)
    if (b.owned) (void)hipFree(b.owned);
  if (p->d_state) (void)hipFree(p->d_state);
  if (p->d_xf) (void)hipFree(p->d_xf);
  if (p->d_ticket) (void)hipFree(p->d_ticket);
  if (p->d_small) (void)hipFree(p->d_small);
  if (p->d_loc) (void)hipFree(p->d_loc);
  if (p->h_box) (void)hipHostFree(p->h_box);
  if (p->ev_lazy) (void)hipEventDestroy(p->ev_lazy);
  if (p->ev_pv) (void)hipEventDestroy(p->ev_pv);
  if (p->d_eval) (void)hipFree(p->d_eval);
  if (p->d_pv) (void)hipFree(p->d_pv);
  if (p->h_pv) (void)hipHostFree(p->h_pv);
  if (p->h_eval) (void)hipHostFree(p->h_eval);
  p->partials.release();
  p->granules.release();
  p->bcast.release();
  p->part_xyz.release();
  p->part_flags.release();
  p->part_counts.release();
  p->residuals.release();
  delete p;
  return HG_OK;
}

int hg_problem_reset(hg_problem* p) {
  HG_REQUIRE_CTX(p);
  bool owned = false;
  for (auto& b : p->blocks) owned = owned || b.owned;
  if (owned) (void)hipStreamSynchronize(p->ctx->stream);
  for (auto& b : p->blocks)
    if (b.owned) (void)hipFree(b.owned);
  p->blocks.clear();
  p->poses.clear();
  p->constant.clear();
  p->velocity.clear();
  p->vfree.clear();
  p->small.clear();
#ifndef HG_BIG
  if (p->promoted) (void)hg_problem_reset_big(p->big);
  p->promoted = false;
#endif
  return HG_OK;
}

int hg_problem_add_pose(hg_problem* p, const double tq[7], int constant) {
  if (!p || !tq) return HG_ERR_INVALID;
#ifdef HG_BIG
  constexpr size_t limit = kMaxPoses;
#else
  constexpr size_t limit = kBigMaxPoses;
#endif
  if (p->poses.size() >= limit) {
    set_last_error("too many pose blocks (limit " + std::to_string(limit) + ")");
    return HG_ERR_CAPACITY;
  }
  std::array<double, 7> a;
  std::memcpy(a.data(), tq, sizeof(double) * 7);
  p->poses.push_back(a);
  p->constant.push_back(constant ? 1 : 0);
  p->velocity.push_back({{0.0, 0.0, 0.0}});
  p->vfree.push_back(0);
#ifndef HG_BIG
  if (p->promoted) {
    const int rc = hg_problem_add_pose_big(p->big, tq, constant);
    if (rc < 0) return rc;
  } else if (p->poses.size() > static_cast<size_t>(kMaxPoses)) {
    const int rc = promote(p);
    if (rc != HG_OK) return rc;
  }
#endif
  return static_cast<int>(p->poses.size()) - 1;
}

int hg_problem_set_pose(hg_problem* p, int index, const double tq[7]) {
  if (!p || !tq || index < 0 || index >= static_cast<int>(p->poses.size())) return HG_ERR_INVALID;
  std::memcpy(p->poses[index].data(), tq, sizeof(double) * 7);
#ifndef HG_BIG
  if (p->promoted) return hg_problem_set_pose_big(p->big, index, tq);
#endif
  return HG_OK;
}

int hg_problem_get_pose(hg_problem* p, int index, double tq[7]) {
  if (!p || !tq || index < 0 || index >= static_cast<int>(p->poses.size())) return HG_ERR_INVALID;
  std::memcpy(tq, p->poses[index].data(), sizeof(double) * 7);
  return HG_OK;
}

int hg_problem_set_velocity(hg_problem* p, int index, const double v[3], int constant) {
  if (!p || !v || index < 0 || index >= static_cast<int>(p->poses.size())) return HG_ERR_INVALID;
  std::memcpy(p->velocity[index].data(), v, sizeof(double) * 3);
  p->vfree[index] = constant ? 0 : 1;
#ifndef HG_BIG
  if (p->promoted) return hg_problem_set_velocity_big(p->big, index, v, constant);
#endif
  return HG_OK;
}

int hg_problem_get_velocity(hg_problem* p, int index, double v[3]) {
  if (!p || !v || index < 0 || index >= static_cast<int>(p->poses.size())) return HG_ERR_INVALID;
  std::memcpy(v, p->velocity[index].data(), sizeof(double) * 3);
  return HG_OK;
}

static int add_small(hg_problem* p, const SmallBlockDev& sb) {
  const int np = static_cast<int>(p->poses.size());
  if (sb.a < 0 || sb.a >= np || sb.b < 0 || sb.b >= np || sb.a == sb.b) return HG_ERR_INVALID;
#ifdef HG_BIG
  constexpr size_t limit = kMaxSmall;
#else
  constexpr size_t limit = kBigMaxSmall;
#endif
  if (p->small.size() >= limit) {
    set_last_error("too many odometry / IMU blocks (limit " + std::to_string(limit) + ")");
    return HG_ERR_CAPACITY;
  }
  p->small.push_back(sb);
#ifndef HG_BIG
  if (p->promoted) {
    const int rc = sb.type == 1 ? hg_problem_add_odometry_block_big(p->big, sb.a, sb.b, sb.w[0], sb.w[1], sb.delta)
                                : hg_problem_add_imu_block_big(p->big, sb.a, sb.b, sb.w[0], sb.w[1], sb.w[2], sb.dt, sb.delta + 3);
    if (rc < 0) return rc;
  } else if (p->small.size() > static_cast<size_t>(kMaxSmall)) {
    const int rc = promote(p);
    if (rc != HG_OK) return rc;
  }
#endif
  return static_cast<int>(p->small.size()) - 1;
}

int hg_problem_add_odometry_block(hg_problem* p, int pose_a, int pose_b, double translation_weight,
                                  double rotation_weight, const double delta_tq[7]) {
  if (!p || !delta_tq) return HG_ERR_INVALID;
  SmallBlockDev sb;
  std::memset(&sb, 0, sizeof(sb));
  sb.type = 1; sb.a = pose_a; sb.b = pose_b;
  sb.w[0] = translation_weight; sb.w[1] = rotation_weight;
  std::memcpy(sb.delta, delta_tq, sizeof(double) * 7);
  return add_small(p, sb);
}

int hg_problem_add_imu_block(hg_problem* p, int pose_a, int pose_b, double translation_weight,
                             double velocity_weight, double rotation_weight, double delta_time_seconds,
                             const double delta_rotation_wxyz[4]) {
  if (!p || !delta_rotation_wxyz) return HG_ERR_INVALID;
  SmallBlockDev sb;
  std::memset(&sb, 0, sizeof(sb));
  sb.type = 2; sb.a = pose_a; sb.b = pose_b;
  sb.w[0] = translation_weight; sb.w[1] = velocity_weight; sb.w[2] = rotation_weight;
  sb.dt = delta_time_seconds;
  std::memcpy(sb.delta + 3, delta_rotation_wxyz, sizeof(double) * 4);
  return add_small(p, sb);
}

static int add_block_impl(hg_problem* p, const float* xyz, const double* factors, size_t n,
                          int memspace, hg_grid* const* pyramid, int levels, int multi_res,
                          double scaling_factor, int pose_a, int pose_b, double interpolation_ratio) {
  if (!p || !pyramid || levels < 1 || levels > kMaxLevels || (n && !xyz)) return HG_ERR_INVALID;
  HG_REQUIRE_CTX(p);
  const int np = static_cast<int>(p->poses.size());
  if (pose_a < 0 || pose_a >= np || pose_b >= np) return HG_ERR_INVALID;
#ifdef HG_BIG
  constexpr size_t block_limit = kMaxBlocks;
#else
  constexpr size_t block_limit = kBigMaxBlocks;
#endif
  if (p->blocks.size() >= block_limit) {
    set_last_error("too many residual blocks (limit " + std::to_string(block_limit) + ")");
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
    // one allocation: xyz (padded to 8 bytes), then the per-return factors
    const size_t xyz_bytes = (n * 3 * sizeof(float) + 7) & ~static_cast<size_t>(7);
    const size_t bytes = xyz_bytes + (factors ? n * sizeof(double) : 0);
    HG_HIP_CHECK(hipSetDevice(p->ctx->device));
    HG_HIP_CHECK(hipMalloc(&b.owned, bytes));
    hipError_t e = hipMemcpyAsync(b.owned, xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice, p->ctx->stream);
    if (e == hipSuccess && factors)
      e = hipMemcpyAsync(static_cast<char*>(b.owned) + xyz_bytes, factors, n * sizeof(double),
                         hipMemcpyHostToDevice, p->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->ctx->stream);
    if (e != hipSuccess) {
      (void)hipFree(b.owned);
      set_last_error(std::string("upload points: ") + hipGetErrorString(e));
      return HG_ERR_HIP;
    }
    b.d_xyz = static_cast<const float*>(b.owned);
    if (factors) b.d_factor = reinterpret_cast<const double*>(static_cast<char*>(b.owned) + xyz_bytes);
  } else {
    b.d_xyz = xyz;
    b.d_factor = factors;
  }
  p->blocks.push_back(b);
#ifndef HG_BIG
  if (p->promoted) {
    const hg_problem::Block& nb = p->blocks.back();
    const int rc = nb.d_factor ? hg_problem_add_unwarped_block_big(p->big, nb.d_xyz, nb.d_factor, nb.n, HG_DEVICE, nb.pyramid.data(),
                                                                   levels, nb.multi_res, nb.scaling, nb.pose_a, nb.pose_b)
                               : hg_problem_add_block_big(p->big, nb.d_xyz, nb.n, HG_DEVICE, nb.pyramid.data(), levels,
                                                          nb.multi_res, nb.scaling, nb.pose_a, nb.pose_b, nb.factor);
    if (rc < 0) return rc;
  } else if (p->blocks.size() > static_cast<size_t>(kMaxBlocks)) {
    const int rc = promote(p);
    if (rc != HG_OK) return rc;
  }
#endif
  return static_cast<int>(p->blocks.size()) - 1;
}

int hg_problem_add_block(hg_problem* p, const float* xyz, size_t n, int memspace,
                         hg_grid* const* pyramid, int levels, int multi_res, double scaling_factor,
                         int pose_a, int pose_b, double interpolation_ratio) {
  return add_block_impl(p, xyz, nullptr, n, memspace, pyramid, levels, multi_res, scaling_factor,
                        pose_a, pose_b, interpolation_ratio);
}

int hg_problem_add_unwarped_block(hg_problem* p, const float* xyz, const double* interpolation_ratios,
                                  size_t n, int memspace, hg_grid* const* pyramid, int levels,
                                  int multi_res, double scaling_factor, int pose_a, int pose_b) {
  if (!p || pose_b < 0 || pose_a == pose_b || (n && !interpolation_ratios)) return HG_ERR_INVALID;
  if (n == 0) {
    set_last_error("unwarped block without returns");
    return HG_ERR_INVALID;
  }
  return add_block_impl(p, xyz, interpolation_ratios, n, memspace, pyramid, levels, multi_res,
                        scaling_factor, pose_a, pose_b, 0.0);
}

int hg_problem_set_block_width(hg_problem* p, int block, size_t width) {
  if (!p || block < 0 || block >= static_cast<int>(p->blocks.size())) return HG_ERR_INVALID;
  hg_problem::Block& b = p->blocks[block];
  // a structured cloud has n = columns x width returns; anything else keeps the plain order
  b.width = (width > 1 && width < (1u << 20) && b.n % width == 0) ? static_cast<unsigned>(width) : 0u;
#ifndef HG_BIG
  if (p->promoted) return hg_problem_set_block_width_big(p->big, block, width);
#endif
  return HG_OK;
}

int hg_problem_num_residuals(hg_problem* p) {
  if (!p) return HG_ERR_INVALID;
  size_t n = 0;
  for (const auto& b : p->blocks)
    if (block_active(p, b)) n += b.n;
  for (const auto& sb : p->small) {
    const bool free_pose = !p->constant[sb.a] || !p->constant[sb.b];
    const bool free_vel = sb.type == 2 && (p->vfree[sb.a] || p->vfree[sb.b]);
    if (free_pose || free_vel) n += (sb.type == 1 ? 6 : 9);
  }
  return static_cast<int>(n);
}

int hg_problem_num_columns(hg_problem* p) {
  if (!p) return HG_ERR_INVALID;
  int c = 0;
  for (size_t i = 0; i < p->constant.size(); ++i) {
    if (!p->constant[i]) c += 6;
    if (p->vfree[i]) c += 3;
  }
  return c;
}

int hg_problem_evaluate(hg_problem* p, double* cost, double* residuals, double* gradient, double* JtJ) {
  HG_REQUIRE_CTX(p);
  hipStream_t s = p->ctx->stream;
  HG_HIP_CHECK(hipSetDevice(p->ctx->device));
#ifndef HG_BIG
  if (p->promoted) return hg_problem_evaluate_big(p->big, cost, residuals, gradient, JtJ);
#endif
  int rc = upload_state(p, nullptr);
#ifndef HG_BIG
  if (rc == HG_ERR_CAPACITY && promote(p) == HG_OK)  // the band outgrew the LDS-resident solver
    return hg_problem_evaluate_big(p->big, cost, residuals, gradient, JtJ);
#endif
  if (rc != HG_OK) return rc;
  const int nres = hg_problem_num_residuals(p);
  double* d_res = nullptr;
  if (residuals && nres > 0) {
    rc = p->residuals.reserve(sizeof(double) * nres);
    if (rc != HG_OK) return rc;
    d_res = p->residuals.as<double>();
  }
  hipLaunchKernelGGL(k_lm, dim3(1), dim3(kLmBlock), 0, s, p->d_state, p->d_xf, static_cast<const double*>(p->d_loc), p->d_small, MODE_PREPARE, p->d_box, p->up_words, 0u);
  HG_HIP_CHECK(hipGetLastError());
  rc = launch_eval(p, d_res, false);
  if (rc != HG_OK) return rc;
  hipLaunchKernelGGL(k_lm, dim3(1), dim3(kLmBlock), 0, s, p->d_state, p->d_xf, static_cast<const double*>(p->d_loc), p->d_small, MODE_ASSEMBLE, static_cast<const PinBox*>(nullptr), 0u, 0u);
  HG_HIP_CHECK(hipGetLastError());
  HG_HIP_CHECK(hipMemcpyAsync(&p->h_state, p->d_state, sizeof(LmState), hipMemcpyDeviceToHost, s));
  if (d_res) HG_HIP_CHECK(hipMemcpyAsync(residuals, d_res, sizeof(double) * nres, hipMemcpyDeviceToHost, s));
  HG_HIP_CHECK(hipStreamSynchronize(s));
  const LmHead& S = p->h_state.h;
  if (cost) *cost = S.cand_cost;
  if (gradient) std::memcpy(gradient, S.gc, sizeof(double) * S.ncols);
  if (JtJ) {  // band -> dense
    const int n = S.ncols, W = S.bw + 1;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        const int hi = std::max(i, j), lo = std::min(i, j);
        JtJ[i * n + j] = (hi - lo < W) ? p->h_state.Hc[band_index(hi, lo, W)] : 0.0;
      }
  }
  return HG_OK;
}

// `lazy`: the caller settles the solve (solve_settle) before it enqueues anything that depends on it; the entry of
// the C ABI enqueues every launch, because its callers may follow it with work of their own.
static int solve_async_impl(hg_problem* p, const hg_solver_opts* opts, bool lazy);
int hg_problem_solve_async(hg_problem* p, const hg_solver_opts* opts) { return solve_async_impl(p, opts, false); }
static int solve_async_impl(hg_problem* p, const hg_solver_opts* opts, bool lazy) {
  HG_REQUIRE_CTX(p);
  p->lazy_left = 0;
  hipStream_t s = p->ctx->stream;
  HG_HIP_CHECK(hipSetDevice(p->ctx->device));
#ifndef HG_BIG
  if (p->promoted) return hg_problem_solve_async_big(p->big, opts);
#endif
  int rc = upload_state(p, opts);
#ifndef HG_BIG
  if (rc == HG_ERR_CAPACITY && promote(p) == HG_OK)  // the band outgrew the LDS-resident solver
    return hg_problem_solve_async_big(p->big, opts);
#endif
  if (rc != HG_OK) return rc;
  const LmHead& S0 = p->h_state.h;
  p->solve_pending = true;
  if (S0.ncols == 0) return HG_OK;
  const int max_it = S0.opt.max_num_iterations;
  // the single-pose registration shape uploads its head inside the first residual launch (FirstUpload)
  const bool first_uploads = p->single_threads && S0.ncols == 6 && S0.bw == 5 && S0.num_blocks == 1 &&
                             S0.blocks[0].active && S0.num_small == 0 && p->num_eval < 2 &&
                             p->ctx->opt(OPT_PREPARE_KERNEL) == 0;
  if (!first_uploads) {
    hipLaunchKernelGGL(k_lm, dim3(1), dim3(kLmBlock), 0, s, p->d_state, p->d_xf, static_cast<const double*>(p->d_loc), p->d_small, MODE_PREPARE, p->d_box, p->up_words, 0u);
    HG_HIP_CHECK(hipGetLastError());
  }
  // the launches of the single-pose registration step are back-to-back: one event pair brackets all
  // of them (an event pair costs ~8 us of stream serialisation); the window pass is bracketed per launch
  p->prof_grouped = first_uploads || (p->single_threads && S0.ncols == 6 && S0.bw == 5 && S0.blocks[0].active);
  p->persist_last = false;
#ifndef HG_BIG
  p->persist_last = first_uploads && persist_ok(p, S0.blocks[0].num_wg);
#endif
  ProfScope group(p->ctx, HG_K_RESIDUALS, static_cast<unsigned long long>(S0.blocks[0].n) * (max_it + 1),
                  p->persist_last ? 1u : static_cast<unsigned>(max_it + 1), p->prof_grouped);
  // General problems (two launches per iteration, 5 us each even when the solve has terminated; a window of ten
  // control points converges in nine or ten of its twelve iterations): the last three iterations are enqueued only
  // if the solve turns out to need them (solve_settle, called before anything that depends on the solve is
  // enqueued). The single-pose chain keeps its launches together: its insertion follows without the host.
#ifndef HG_BIG
  const bool lazy_ok = p->ctx->opt(OPT_EAGER_SOLVE) == 0;
  const int kLazyTail = static_cast<int>(std::max<long long>(1, p->ctx->opt(OPT_LAZY_TAIL)));
  if (lazy && lazy_ok && !p->prof_grouped && max_it + 1 >= 2 * kLazyTail + 2) p->lazy_left = kLazyTail;
#endif
#ifndef HG_BIG
  if (p->persist_last) {
    rc = launch_persistent(p, static_cast<unsigned>(max_it + 1));
    if (rc != HG_OK) return rc;
  } else
#endif
  for (int it = 0; it <= max_it - p->lazy_left; ++it) {
    // residuals of every block at the candidate + one LM step
    rc = launch_eval(p, nullptr, true, first_uploads && it == 0);
    if (rc != HG_OK) return rc;
  }
  if (p->lazy_left > 0) {
    if (!p->ev_lazy) HG_HIP_CHECK(hipEventCreateWithFlags(&p->ev_lazy, hipEventDisableTiming));
    HG_HIP_CHECK(hipEventRecord(p->ev_lazy, s));
  }
  p->prof_grouped = false;
  return HG_OK;
}

// Behind hg_problem_solve_async of a general problem: returns once the solve has terminated on the device or all
// of its launches are enqueued, so that work which depends on the solve can follow on the stream.
static int solve_settle(hg_problem* p) {
  if (!p || p->lazy_left <= 0) return HG_OK;
  volatile unsigned long long* flag = &p->h_box->flag;
  for (unsigned long long spin = 0;; ++spin) {
    if (*flag == p->seq) break;  // terminated: the held-back launches are not needed
    __builtin_ia32_pause();
    if ((spin & 0xFFull) == 0xFFull) {
      const hipError_t q = hipEventQuery(p->ev_lazy);
      if (q == hipSuccess) {
        if (*flag == p->seq) break;
        const int left = p->lazy_left;
        p->lazy_left = 0;
        for (int it = 0; it < left; ++it) {
          const int rc = launch_eval(p, nullptr, true, false);
          if (rc != HG_OK) return rc;
        }
        return HG_OK;
      }
      if (q != hipErrorNotReady) {
        set_last_error(std::string("solve: ") + hipGetErrorString(q));
        return HG_ERR_HIP;
      }
    }
  }
  p->lazy_left = 0;
  return HG_OK;
}

int hg_problem_fetch(hg_problem* p, hg_solver_summary* summary) {
#ifndef HG_BIG
  if (p && p->promoted) {
    const int rc = hg_problem_fetch_big(p->big, summary);
    if (rc == HG_OK) pull_from_big(p);
    return rc;
  }
#endif
  if (!p || !p->solve_pending) return HG_ERR_INVALID;
  hipStream_t s = p->ctx->stream;
  {
    const int rc = solve_settle(p);
    if (rc != HG_OK) return rc;
  }
  p->solve_pending = false;
  if (p->h_state.h.ncols == 0) {
    if (summary) std::memset(summary, 0, sizeof(*summary));
    return HG_OK;
  }
  {
    // the terminating LM step stores the head into the mailbox and then its sequence number; work
    // enqueued behind the solve (the insertion of hg_register_scan) keeps running meanwhile
    volatile unsigned long long* flag = &p->h_box->flag;
    bool arrived = false;
    // the stream is queried (for errors, or a drained stream) only every 2^20 polls: the flag is what
    // the host waits for
    constexpr unsigned long long query_mask = 0xFFFFFull;
    for (unsigned long long spin = 0; !arrived; ++spin) {
      if (*flag == p->seq) { arrived = true; break; }
      __builtin_ia32_pause();
      if ((spin & query_mask) == query_mask) {
        const hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) break;  // stream drained: decide below
        if (q != hipErrorNotReady) {
          set_last_error(std::string("solve: ") + hipGetErrorString(q));
          return HG_ERR_HIP;
        }
      }
    }
    if (!arrived) arrived = (*flag == p->seq);
    if (arrived) {
      std::atomic_thread_fence(std::memory_order_acquire);
      std::memcpy(&p->h_state.h, &p->h_box->h, sizeof(LmHead));
    } else {
      // stream finished without a terminating step (cannot happen with max_it + 1 launches): read back
      HG_HIP_CHECK(hipMemcpy(&p->h_state.h, p->d_state, sizeof(LmHead), hipMemcpyDeviceToHost));
    }
  }
  const LmHead& S = p->h_state.h;
#ifdef HG_EVAL_STAMPS
  {
    static unsigned long long st[1024][4];
    (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_eval_stamps), sizeof(st));
    unsigned long long t0 = ~0ull, body_end = 0, tail0 = 0, tail1 = 0, last_start = 0;
    const unsigned nwg = std::min(1024u, p->h_state.h.blocks[0].num_wg);
    for (unsigned w = 0; w < nwg; ++w) {
      t0 = std::min(t0, st[w][0]);
      last_start = std::max(last_start, st[w][0]);
      body_end = std::max(body_end, st[w][1]);
      if (st[w][3] > tail1) { tail1 = st[w][3]; tail0 = st[w][2]; }
    }
    fprintf(stderr, "eval timeline (us from first WG start): last WG start %.2f, last body end %.2f, tail start %.2f, tail end %.2f\n",
            (last_start - t0) * 0.01, (body_end - t0) * 0.01, (tail0 - t0) * 0.01, (tail1 - t0) * 0.01);
    {
      static unsigned long long bs[1024][8];
      (void)hipMemcpyFromSymbol(bs, HIP_SYMBOL(g_body_stamps), sizeof(bs));
      double acc[6] = {0, 0, 0, 0, 0, 0}, mx[6] = {0, 0, 0, 0, 0, 0};
      for (unsigned w = 0; w < nwg; ++w)
        for (int k = 1; k <= 5; ++k) {
          const double d = (bs[w][k] - bs[w][k - 1]) * 0.01;
          acc[k] += d / nwg;
          mx[k] = std::max(mx[k], d);
        }
      fprintf(stderr, "  body wave0 (us) mean/max: setup %.2f/%.2f, probes %.2f/%.2f, voxels %.2f/%.2f, interp+row %.2f/%.2f, reduce %.2f/%.2f\n",
              acc[1], mx[1], acc[2], mx[2], acc[3], mx[3], acc[4], mx[4], acc[5], mx[5]);
    }
    unsigned long long ts[8];
    (void)hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_tail_stamps), sizeof(ts));
    fprintf(stderr, "  fast tail (us): reduce %.2f, state %.2f, assemble %.2f, decide %.2f, solve %.2f\n",
            (ts[1] - ts[0]) * 0.01, (ts[2] - ts[1]) * 0.01, (ts[3] - ts[2]) * 0.01, (ts[4] - ts[3]) * 0.01, (ts[5] - ts[4]) * 0.01);
  }
#endif
#ifdef HG_LM_STAMPS
  fprintf(stderr, "lm stamps (cycles of the last step):");
  fprintf(stderr, " [load]%lld", S.stamps[0] - S.stamps[8]);
  for (int i = 1; i < 8; ++i) fprintf(stderr, " [%d]%lld", i, S.stamps[i] - S.stamps[i - 1]);
  fprintf(stderr, " | stamps 0 -> 6: %lld ticks of s_memtime in %lld x 10 ns of s_memrealtime = %.3f ticks per ns", S.stamps[6] - S.stamps[0],
          S.stamps[12] - S.stamps[11], double(S.stamps[6] - S.stamps[0]) / (10.0 * double(S.stamps[12] - S.stamps[11])));
#if !defined(HG_BIG) && HG_LM_STAMPS >= 2
  {
    long long tw[64];
    (void)hipMemcpyFromSymbol(tw, HIP_SYMBOL(g_tw_stamps), sizeof(tw));
    fprintf(stderr, "\n  twisted: entry %lld, elimination slots %lld %lld %lld %lld, middle %lld, unwinding slots %lld %lld %lld %lld",
            tw[1] - tw[0], tw[2] - tw[1], tw[3] - tw[2], tw[4] - tw[3], tw[5] - tw[4], tw[20] - tw[5], tw[21] - tw[20], tw[22] - tw[21],
            tw[23] - tw[22], tw[24] - tw[23]);
    fprintf(stderr, "; slot 0 of wavefront 0: to step %lld, loads %lld + %lld, columns %lld, X store %lld, Schur %lld, to barrier %lld",
            tw[30] - tw[1], tw[31] - tw[30], tw[35] - tw[31], tw[32] - tw[35], tw[33] - tw[32], tw[34] - tw[33], tw[2] - tw[34]);
  }
#endif
  fprintf(stderr, "\n");
#endif
  for (int i = 0; i < S.num_poses; ++i) {
    std::memcpy(p->poses[i].data(), S.x[i], sizeof(double) * 7);
    std::memcpy(p->velocity[i].data(), S.x[i] + 7, sizeof(double) * 3);
  }
#if defined(HG_PERSIST_STAMPS) && !defined(HG_BIG)
  if (p->persist_last) {
    unsigned long long st[16][8];
    (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_persist_stamps), sizeof(st));
    const unsigned long long t0 = st[0][0];
    fprintf(stderr, "persist stamps (us from the first body start): eval: wg0 body start / body end / step end / hand-back stored | wg97 hand-back seen / body end | wg195 body end\n");
    for (int e = 0; e < 14 && e <= S.num_iterations; ++e)
      fprintf(stderr, "  %2d: %7.2f %7.2f %7.2f %7.2f | %7.2f %7.2f | %7.2f\n", e, (st[e][0] - t0) * 0.01, (st[e][1] - t0) * 0.01,
              (st[e][2] - t0) * 0.01, (st[e][3] - t0) * 0.01, (st[e][4] - t0) * 0.01, (st[e][5] - t0) * 0.01, (st[e][6] - t0) * 0.01);
  }
#endif
  if (p->persist_last && S.termination_reason == 7) {
    // the persistent launch gave up waiting for one of its workgroups (they were not resident together): the solve is
    // reported as FAILURE with the last accepted pose, and this context goes back to a launch per evaluation
    p->ctx->persist_failed = true;
    set_last_error("persistent solve timed out waiting for a workgroup; the context now launches per evaluation");
  }
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

int hg_problem_solve(hg_problem* p, const hg_solver_opts* opts, hg_solver_summary* summary) {
  int rc = solve_async_impl(p, opts, true);
  if (rc != HG_OK) return rc;
  return hg_problem_fetch(p, summary);
}

}  // extern "C"

namespace {
// Where a solve leaves the control poses in device memory (pose k at base + k * stride doubles), or nullptr
// for a problem without free parameters (nothing is launched: its poses are the host's).
const double* device_poses(hg_problem* p, int* stride) {
#ifndef HG_BIG
  if (p->promoted) return big_device_poses(p->big, stride);
#endif
  *stride = kState;
  return (p->h_state.h.ncols == 0) ? nullptr : &p->d_state->h.x[0][0];
}

// hg_register_scan*: the scan that is inserted (xyz, n, width) is normally the cloud of one of the problem's
// TSDF blocks; that block learns the scan's structure (hg_problem_set_block_width).
void apply_scan_width(hg_problem* p, const float* xyz, size_t n, size_t width) {
  if (width < 2) return;
  for (size_t b = 0; b < p->blocks.size(); ++b)
    if (p->blocks[b].n == n && p->blocks[b].width == 0 && (p->blocks[b].d_xyz == xyz || p->blocks.size() == 1))
      (void)hg_problem_set_block_width(p, static_cast<int>(b), width);
}

// The gather rate of the residual pass saturates well below this; larger lists go through in groups.
constexpr int kBatchGroup = 64;

// One iteration of a batch of general problems: residual passes + the step kernel over the job table.
static void window_jobs_iteration(hg_ctx* c, const WindowJob* d_jobs) {
  hipStream_t s = c->stream;
  const unsigned max_plain = c->lazy_batch_dims[0], max_unwarp = c->lazy_batch_dims[1], count = c->lazy_batch_dims[2];
  if (max_plain > 0) {
    ProfScope ps(c, HG_K_RESIDUALS, c->lazy_batch_units[0]);
    hipLaunchKernelGGL(k_window_residuals_jobs<false>, dim3(max_plain, count), dim3(kBatchThreads), 0, s, d_jobs);
  }
  if (max_unwarp > 0) {
    ProfScope ps(c, HG_K_RESIDUALS, c->lazy_batch_units[1]);
    hipLaunchKernelGGL(k_window_residuals_jobs<true>, dim3(max_unwarp, count), dim3(kBatchThreads), 0, s, d_jobs);
  }
  {
    ProfScope ps(c, HG_K_LM, count);
    hipLaunchKernelGGL(k_lm_jobs, dim3(count), dim3(kLmBlock), 0, s, d_jobs, static_cast<int>(MODE_STEP));
  }
}
// Behind a batched solve of general problems: returns once every problem has terminated on the device or all
// launches are enqueued (see solve_settle).
// Pinned staging of a batched solve's job table: a ring of kJobSlots slots, so that a batch can be enqueued while the
// table copies of the batches before it are still waiting in the stream (hg_problem_solve_batch_async: the caller
// builds batch k + 1 while batch k runs). A slot is handed out again only after its copy has been read (ev_jobs).
constexpr unsigned kJobSlots = 4;
static int jobs_staging(hg_ctx* c, size_t bytes, void** out, unsigned* slot) {
  if (c->jobs_capacity < bytes) {
    // (growing: no copy may be reading the old ring)
    HG_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->pinned_jobs) (void)hipHostFree(c->pinned_jobs);
    c->pinned_jobs = nullptr;
    c->jobs_capacity = 0;
    const size_t one = kBatchGroup * std::max(2 * sizeof(SingleJob) + sizeof(PartJob), sizeof(WindowJob)) + 256;
    const size_t cap = (std::max(one, bytes) + 255) & ~size_t(255);
    HG_HIP_CHECK(hipHostMalloc(&c->pinned_jobs, cap * kJobSlots));
    c->jobs_capacity = cap;
  }
  *slot = c->jobs_slot++ % kJobSlots;
  if (c->ev_jobs[*slot]) HG_HIP_CHECK(hipEventSynchronize(c->ev_jobs[*slot]));
  *out = static_cast<char*>(c->pinned_jobs) + static_cast<size_t>(*slot) * c->jobs_capacity;
  return HG_OK;
}
static int jobs_staging_sent(hg_ctx* c, unsigned slot) {  // behind the hipMemcpyAsync that reads the slot
  if (!c->ev_jobs[slot]) HG_HIP_CHECK(hipEventCreateWithFlags(&c->ev_jobs[slot], hipEventDisableTiming));
  HG_HIP_CHECK(hipEventRecord(c->ev_jobs[slot], c->stream));
  return HG_OK;
}

static int batch_settle(hg_ctx* c) {
  if (!c || c->lazy_batch_left <= 0) return HG_OK;
  const WindowJob* d_jobs = static_cast<const WindowJob*>(c->ws_misc.ptr);
  for (unsigned long long spin = 0;; ++spin) {
    bool all = true;
    for (size_t i = 0; i < c->lazy_batch_flags.size() && all; ++i) all = *c->lazy_batch_flags[i] == c->lazy_batch_seqs[i];
    if (all) break;
    __builtin_ia32_pause();
    if ((spin & 0xFFull) == 0xFFull) {
      const hipError_t q = hipEventQuery(c->ev_lazy_batch);
      if (q == hipSuccess) {
        const int left = c->lazy_batch_left;
        c->lazy_batch_left = 0;
        for (int it = 0; it < left; ++it) window_jobs_iteration(c, d_jobs);
        HG_HIP_CHECK(hipGetLastError());
        return HG_OK;
      }
      if (q != hipErrorNotReady) {
        set_last_error(std::string("solve: ") + hipGetErrorString(q));
        return HG_ERR_HIP;
      }
    }
  }
  c->lazy_batch_left = 0;
  return HG_OK;
}

// Uploads every problem and, if all of them have the single-pose shape, enqueues their solves with
// shared launches (*batched = true; results are collected by hg_problem_fetch on each problem).
// Otherwise nothing is enqueued (*batched = false). count <= kBatchGroup.
int solve_batch_enqueue(hg_problem* const* problems, int count, const hg_solver_opts* opts, bool* batched) {
  *batched = false;
  for (int i = 0; i < count; ++i)
    if (!problems[i] || problems[i]->ctx != problems[0]->ctx) return HG_ERR_INVALID;
  HG_REQUIRE_CTX(problems[0]);
  hg_ctx* c = problems[0]->ctx;
  hipStream_t s = c->stream;
  HG_HIP_CHECK(hipSetDevice(c->device));
  // the batched launch covers problems that take the register-resident single-pose step
  bool batchable = count >= 2;
  bool windows = count >= 2 && c->opt(OPT_WINDOW_BATCH) != 0;  // general problems sharing their launches
  int rc = HG_OK;
  for (int i = 0; i < count && rc == HG_OK; ++i) {
    hg_problem* p = problems[i];
#ifndef HG_BIG
    if (p->promoted) {  // a big problem is solved on its own
      batchable = windows = false;
      continue;
    }
#endif
    p->batch_share = count;
    rc = upload_state(p, opts);
    p->batch_share = 1;
#ifndef HG_BIG
    if (rc == HG_ERR_CAPACITY) {  // ditto (hg_problem_solve promotes it)
      rc = HG_OK;
      batchable = windows = false;
      continue;
    }
#endif
    if (rc != HG_OK) break;
    const LmHead& S = p->h_state.h;
    if (!(p->single_threads && S.ncols == 6 && S.bw == 5 && S.num_blocks == 1 && S.num_small == 0 &&
          S.blocks[0].active && S.opt.max_num_iterations == problems[0]->h_state.h.opt.max_num_iterations))
      batchable = false;
    if (p->single_threads || S.ncols == 0 || S.opt.max_num_iterations != problems[0]->h_state.h.opt.max_num_iterations)
      windows = false;
  }
  if (rc != HG_OK) return rc;
  if (!batchable && windows) {
    // window batch: k_lm / k_window_residuals over a table of problems (grid row = problem)
    const size_t bytes = static_cast<size_t>(count) * sizeof(WindowJob);
    void* staging = nullptr;
    unsigned slot = 0;
    if ((rc = jobs_staging(c, bytes, &staging, &slot)) != HG_OK) return rc;
    WindowJob* jobs = static_cast<WindowJob*>(staging);
    unsigned max_plain = 0, max_unwarp = 0;
    unsigned long long units_plain = 0, units_unwarp = 0;
    for (int i = 0; i < count; ++i) {
      hg_problem* p = problems[i];
      const LmHead& S = p->h_state.h;
      WindowJob& J = jobs[i];
      std::memset(&J, 0, sizeof(J));
      J.blocks = p->d_eval;
      J.num_plain = p->num_plain;
      J.num_unwarp = p->num_unwarp;
      J.wg_plain = p->wg_plain;
      J.wg_unwarp = p->wg_unwarp;
      J.num_small = static_cast<unsigned>(S.num_small);
      J.tiles = p->tiles;
      J.G = p->d_state;
      J.xf = p->d_xf;
      J.partials = p->partials.as<double>();
      J.small_out = p->d_small;
      J.tickets = p->d_ticket;
      J.loc = p->d_loc;
      J.box = p->d_box;
      J.up_words = p->up_words;
      J.stage = lm_stage_counts(S);
      max_plain = std::max(max_plain, J.wg_plain + J.num_small);
      max_unwarp = std::max(max_unwarp, J.wg_unwarp);
      for (int b = 0; b < S.num_blocks; ++b)
        if (S.blocks[b].active) (p->blocks[b].d_factor ? units_unwarp : units_plain) += S.blocks[b].n;
    }
    if ((rc = c->ws_misc.reserve(bytes)) != HG_OK) return rc;
    const WindowJob* d_jobs = static_cast<const WindowJob*>(c->ws_misc.ptr);
    HG_HIP_CHECK(hipMemcpyAsync(c->ws_misc.ptr, jobs, bytes, hipMemcpyHostToDevice, s));
    if ((rc = jobs_staging_sent(c, slot)) != HG_OK) return rc;
    hipLaunchKernelGGL(k_lm_jobs, dim3(count), dim3(kLmBlock), 0, s, d_jobs, static_cast<int>(MODE_PREPARE));
    HG_HIP_CHECK(hipGetLastError());
    const int max_it = problems[0]->h_state.h.opt.max_num_iterations;
    // the last three iterations are held back until some window is seen to need them (batch_settle), as in the
    // solve of one general problem (solve_async_impl)
    c->lazy_batch_left = 0;
    c->lazy_batch_dims[0] = max_plain; c->lazy_batch_dims[1] = max_unwarp; c->lazy_batch_dims[2] = static_cast<unsigned>(count);
    c->lazy_batch_units[0] = units_plain; c->lazy_batch_units[1] = units_unwarp;
    const bool lazy_ok = c->opt(OPT_EAGER_SOLVE) == 0;
    if (lazy_ok && max_it + 1 >= 8) {
      c->lazy_batch_left = 3;
      c->lazy_batch_flags.clear();
      c->lazy_batch_seqs.clear();
      for (int i = 0; i < count; ++i) {
        c->lazy_batch_flags.push_back(&problems[i]->h_box->flag);
        c->lazy_batch_seqs.push_back(problems[i]->seq);
      }
    }
    for (int it = 0; it <= max_it - c->lazy_batch_left; ++it) window_jobs_iteration(c, d_jobs);
    HG_HIP_CHECK(hipGetLastError());
    if (c->lazy_batch_left > 0) {
      if (!c->ev_lazy_batch) HG_HIP_CHECK(hipEventCreateWithFlags(&c->ev_lazy_batch, hipEventDisableTiming));
      HG_HIP_CHECK(hipEventRecord(c->ev_lazy_batch, s));
    }
    for (int i = 0; i < count; ++i) problems[i]->solve_pending = true;
    *batched = true;
    return HG_OK;
  }
  if (!batchable) return HG_OK;
  // job tables: one slot of the context's pinned staging ring, one asynchronous copy, no stream synchronisation
  void* staging = nullptr;
  unsigned slot = 0;
  if ((rc = jobs_staging(c, 2 * sizeof(SingleJob) * static_cast<size_t>(count) + sizeof(PartJob) * static_cast<size_t>(count) + 192,
                         &staging, &slot)) != HG_OK)
    return rc;
  SingleJob* jobs = static_cast<SingleJob*>(staging);
  unsigned max_wg = 0;
  unsigned long long units = 0;
  // Level partition (see k_level_classify): worth its three launches and one finest-level lookup per return once the
  // batch is large enough to be bound by throughput (partition_min problems: 64 matches 27.3k -> 29.4k matches/s with
  // the classify kernel, 32 and 16 even -- default 48 then; with the classification folded into the residual launch in
  // front 32 matches gain 3 %, 28.4k -> 29.3k, 16 do not: default 32; 0 switches it off)
  const int part_min = static_cast<int>(c->opt(OPT_PARTITION_MIN));
  bool partition = part_min > 0 && count >= part_min;
  for (int i = 0; i < count && partition; ++i) {
    const PyramidView& pv = problems[i]->h_pv[0];
    if (!(pv.multi_res && pv.levels >= 2)) partition = false;
  }
  PartJob* pjobs = nullptr;
  unsigned max_pwg = 0;
  // The classification rides on the residual launch in front of the partition (round 6, `partition_fold`, default 1):
  // that launch holds every return's finest-level validity anyway, so the partition costs a scan and a scatter
  // instead of a lookup pass of its own (k_level_classify: 46 us and 80 MB per 64 matches); the returns are then
  // classified at the candidate BEFORE the one the partitioned launches start with. partition_at = 0 has no launch
  // in front and keeps the classify kernel.
  const int part_at = static_cast<int>(c->opt(OPT_PARTITION_AT));
  const bool fold = partition && c->opt(OPT_PARTITION_FOLD) != 0 && std::min(part_at, problems[0]->h_state.h.opt.max_num_iterations) >= 1;
  if (partition) {
    // (the job tables of the partition live behind the SingleJob table in the same staging slot / device buffer)
    pjobs = reinterpret_cast<PartJob*>(reinterpret_cast<char*>(jobs) + ((sizeof(SingleJob) * static_cast<size_t>(count) + 63) & ~size_t(63)));
    for (int i = 0; i < count; ++i) {
      hg_problem* p = problems[i];
      const BlockInfo& bi = p->h_state.h.blocks[0];
      PartJob& Q = pjobs[i];
      std::memset(&Q, 0, sizeof(Q));
      Q.pv = p->h_pv[0];
      Q.xyz = p->blocks[0].d_xyz;
      Q.n = bi.n;
      Q.nwg = (bi.n + 255u) / 256u;
      Q.width = p->blocks[0].width;
      if ((rc = p->part_xyz.reserve(static_cast<size_t>(bi.n) * 12u)) != HG_OK) return rc;
      if ((rc = p->part_flags.reserve(bi.n)) != HG_OK) return rc;
      // [nwg + 1 counts | per wavefront of 64 returns: 4 nwg counts, 16-byte aligned]
      const size_t wave_at = (static_cast<size_t>(Q.nwg) + 1u + 3u) & ~size_t(3);
      if ((rc = p->part_counts.reserve((wave_at + 4u * static_cast<size_t>(Q.nwg)) * sizeof(unsigned))) != HG_OK) return rc;
      Q.out = p->part_xyz.as<float>();
      Q.flags = p->part_flags.as<unsigned char>();
      Q.counts = p->part_counts.as<unsigned>();
      Q.wave_counts = fold ? Q.counts + wave_at : nullptr;
      Q.G = p->d_state;
      Q.xf = p->d_xf;
      max_pwg = std::max(max_pwg, Q.nwg);
    }
  }
  for (int i = 0; i < count; ++i) {
    hg_problem* p = problems[i];
    const LmHead& S = p->h_state.h;
    const BlockInfo& bi = S.blocks[0];
    SingleJob& J = jobs[i];
    std::memset(&J, 0, sizeof(J));
    J.pv = p->h_pv[0];  // built by upload_state; also resident at p->d_pv (self_mem)
    J.xyz = p->blocks[0].d_xyz;
    J.fast_n = nullptr;
    J.xf = p->d_xf;
    // the batched pass has its own workgroup size; several tiles per workgroup once the batch fills the chip more
    // than once (64 matches of 100k returns: 25.5k / 27.5k / 28.3k / 27.9k matches/s at 1 / 2 / 4 / 8 tiles; 32: 22.7k /
    // 24.6k / 24.5k at 1 / 2 / 4; 8 and 16: + 3 - 5 % at 2, back to where it was at 4). HG_BATCH_TILES overrides.
    const int env_tiles = static_cast<int>(c->opt(OPT_BATCH_TILES));
    J.tiles = env_tiles > 0 ? static_cast<unsigned>(env_tiles) : (count >= 48 ? 4u : count >= 8 ? 2u : 1u);
    J.num_wg = (bi.n + kBatchThreads * J.tiles - 1) / (kBatchThreads * J.tiles);
    if ((rc = p->partials.reserve(static_cast<size_t>(J.num_wg) * kAcc * sizeof(double))) != HG_OK) return rc;
    J.partials = p->partials.as<double>();
    J.G = p->d_state;
    J.ticket = p->d_ticket;
    J.scaling = bi.scaling;
    J.n = bi.n;
    J.box = p->d_box;
    J.up_words = p->up_words;
    J.width = p->blocks[0].width;
    if (fold) {
      J.flags = pjobs[i].flags;
      J.wave_counts = const_cast<unsigned*>(pjobs[i].wave_counts);
    }
    max_wg = std::max(max_wg, J.num_wg);
    units += bi.n;
  }
  // device tables: [SingleJob x count | PartJob x count | SingleJob x count over the partitioned clouds]
  const size_t single_bytes = (sizeof(SingleJob) * static_cast<size_t>(count) + 63) & ~size_t(63);
  const size_t part_bytes = (sizeof(PartJob) * static_cast<size_t>(count) + 63) & ~size_t(63);
  const SingleJob* d_jobs = nullptr;
  const SingleJob* d_jobs_part = nullptr;
  const PartJob* d_pjobs = nullptr;
  {
    size_t table_bytes = sizeof(SingleJob) * static_cast<size_t>(count);
    if (partition) {
      // the third table: the same jobs over the partitioned clouds (lane order already in the copy)
      SingleJob* jobs2 = reinterpret_cast<SingleJob*>(reinterpret_cast<char*>(jobs) + single_bytes + part_bytes);
      for (int i = 0; i < count; ++i) {
        jobs2[i] = jobs[i];
        jobs2[i].xyz = problems[i]->part_xyz.as<float>();
        jobs2[i].fast_n = problems[i]->part_counts.as<unsigned>() + pjobs[i].nwg;
        jobs2[i].width = 0u;
      }
      table_bytes = single_bytes + part_bytes + sizeof(SingleJob) * static_cast<size_t>(count);
    }
    if ((rc = c->ws_misc.reserve(table_bytes)) != HG_OK) return rc;
    HG_HIP_CHECK(hipMemcpyAsync(c->ws_misc.ptr, jobs, table_bytes, hipMemcpyHostToDevice, s));
    if ((rc = jobs_staging_sent(c, slot)) != HG_OK) return rc;
    d_jobs = static_cast<const SingleJob*>(c->ws_misc.ptr);
    if (partition) {
      d_pjobs = reinterpret_cast<const PartJob*>(static_cast<const char*>(c->ws_misc.ptr) + single_bytes);
      d_jobs_part = reinterpret_cast<const SingleJob*>(static_cast<const char*>(c->ws_misc.ptr) + single_bytes + part_bytes);
    }
  }
  hipLaunchKernelGGL(k_lm_prepare_batch, dim3(count), dim3(kLmBlock), 0, s, d_jobs);
  HG_HIP_CHECK(hipGetLastError());
  const int max_it = problems[0]->h_state.h.opt.max_num_iterations;
  // The returns are classified at the candidate the solve evaluates THIRD: the first two steps take the pose from
  // the initial guess (centimetres off) to within a millimetre of where it ends, and a return on the edge of the
  // finest level's known voxels changes sides with every voxel the pose moves -- classified at the guess itself,
  // 35 % of the wavefronts that were expected fast held such a lane a few iterations later and paid the second
  // round trip. HG_PARTITION_AT overrides the iteration.
  // (Round 4, measured and dropped: the batch cut in two halves on two streams, the second one residual pass behind
  // the first, so that one half's step kernel -- `count` workgroups on an otherwise idle chip, 10 us per iteration
  // against 22 us of residual pass for eight 100k-point scans -- would run under the other half's residual pass.
  // The two chains fall into step within two iterations (a residual pass of four scans alone takes 17 us, of eight
  // 22 us: sharing the chip costs it little), both step kernels then run side by side as before: 10.26k against
  // 10.20k scans/s for eight submaps, no gain at 16 or 64.)
  {
    ProfScope group(c, HG_K_RESIDUALS, units * (max_it + 1), static_cast<unsigned>(max_it + 1), true);
    const SingleJob* table = d_jobs;
    for (int it = 0; it <= max_it; ++it) {
      if (partition && it == std::min(part_at, max_it)) {
        if (!fold) hipLaunchKernelGGL(k_level_classify, dim3(max_pwg, count), dim3(256), 0, s, d_pjobs);
        hipLaunchKernelGGL(k_level_scan, dim3(count), dim3(1024), 0, s, d_pjobs);
        hipLaunchKernelGGL(k_level_scatter, dim3(max_pwg, count), dim3(256), 0, s, d_pjobs);
        table = d_jobs_part;
      }
      // (a direct-only kernel at 117 VGPRs with a general-only twin launched behind it was measured 3-5 %
      // slower than this one at 144 VGPRs with the general path as a cold call: the window test up
      // front and the second launch cost more than the fourth wavefront per SIMD brings)
      if (table == d_jobs_part)
        hipLaunchKernelGGL((k_tsdf_residuals_single_batch<kBatchThreads, 1>), dim3(max_wg, count), dim3(kBatchThreads), 0, s, table);
      else if (fold && it + 1 == std::min(part_at, max_it))
        hipLaunchKernelGGL((k_tsdf_residuals_single_batch<kBatchThreads, 2>), dim3(max_wg, count), dim3(kBatchThreads), 0, s, table);
      else
        hipLaunchKernelGGL((k_tsdf_residuals_single_batch<kBatchThreads, 0>), dim3(max_wg, count), dim3(kBatchThreads), 0, s, table);
      hipLaunchKernelGGL(k_lm_single_batch, dim3(count), dim3(kEvalThreads), 0, s, table);
    }
  }
  HG_HIP_CHECK(hipGetLastError());
  for (int i = 0; i < count; ++i) problems[i]->solve_pending = true;
  *batched = true;
  return HG_OK;
}
}  // namespace

static int register_scan_step(hg_problem* p, const hg_solver_opts* sopts, int pose_index,
                              hg_grid* const* grids, const hg_insert_opts* iopts, int levels,
                              const float origin[3], const float* xyz, size_t n, size_t width, int memspace,
                              int insert_mode, double pose_out[7], hg_solver_summary* summary,
                              int (*between)(void*), void* between_arg);

extern "C" {

int hg_problem_solve_batch(hg_problem* const* problems, int count, const hg_solver_opts* opts,
                           hg_solver_summary* summaries) {
  if (!problems || count < 1) return HG_ERR_INVALID;
  if (count > kBatchGroup) {
    for (int i0 = 0; i0 < count; i0 += kBatchGroup) {
      const int rc = hg_problem_solve_batch(problems + i0, std::min(kBatchGroup, count - i0), opts,
                                            summaries ? summaries + i0 : nullptr);
      if (rc != HG_OK) return rc;
    }
    return HG_OK;
  }
  bool batched = false;
  // HG_HOST_TIMES=1: where the host spends the call (stderr; enqueue = uploads + tables + launches, first fetch =
  // the wait for the device)
  const bool host_times = problems && count > 0 && problems[0] && problems[0]->ctx && problems[0]->ctx->opt(OPT_HOST_TIMES) != 0;
  const auto t_begin = std::chrono::steady_clock::now();
  int rc = solve_batch_enqueue(problems, count, opts, &batched);
  if (rc != HG_OK) return rc;
  if (!batched) {
    for (int i = 0; i < count; ++i) {
      rc = hg_problem_solve(problems[i], opts, summaries ? summaries + i : nullptr);
      if (rc != HG_OK) return rc;
    }
    return HG_OK;
  }
  const auto t_enqueued = std::chrono::steady_clock::now();
  if ((rc = batch_settle(problems[0]->ctx)) != HG_OK) return rc;
  auto t_first = t_enqueued;
  for (int i = 0; i < count; ++i) {
    const int r2 = hg_problem_fetch(problems[i], summaries ? summaries + i : nullptr);
    if (r2 != HG_OK) rc = r2;
    if (i == 0) t_first = std::chrono::steady_clock::now();
  }
  if (host_times) {
    const auto t_end = std::chrono::steady_clock::now();
    auto us = [](std::chrono::steady_clock::duration d) { return std::chrono::duration<double, std::micro>(d).count(); };
    fprintf(stderr, "hg_problem_solve_batch(%d): enqueue %.1f us, settle + first fetch %.1f us, other fetches %.1f us\n", count,
            us(t_enqueued - t_begin), us(t_first - t_enqueued), us(t_end - t_first));
  }
#if defined(HG_DIAG_LEVELS) && !defined(HG_BIG)
  {
    unsigned long long st[8], zero[8] = {0};
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(st, HIP_SYMBOL(hg::g_diag_levels), sizeof(st));
    hipMemcpyToSymbol(HIP_SYMBOL(hg::g_diag_levels), zero, sizeof(zero));
    if (st[0])
      fprintf(stderr, "levels: %llu wavefronts, %.1f%% with every lane at the finest level, %.1f%% with at most 4 lanes beyond it; "
              "%.1f%% of the lanes at the finest level; wavefronts predicted fast %llu, of which %.1f%% fell back (%.2f lanes each)\n",
              st[0], 100.0 * st[1] / st[0], 100.0 * st[4] / st[0], 100.0 * st[3] / st[2], st[5], st[5] ? 100.0 * st[6] / st[5] : 0.0,
              st[6] ? double(st[7]) / st[6] : 0.0);
  }
#endif
  return rc;
}

int hg_problem_solve_batch_async(hg_problem* const* problems, int count, const hg_solver_opts* opts) {
  if (!problems || count < 1) return HG_ERR_INVALID;
  if (count > kBatchGroup) {
    for (int i0 = 0; i0 < count; i0 += kBatchGroup) {
      const int rc = hg_problem_solve_batch_async(problems + i0, std::min(kBatchGroup, count - i0), opts);
      if (rc != HG_OK) return rc;
    }
    return HG_OK;
  }
  bool batched = false;
  int rc = solve_batch_enqueue(problems, count, opts, &batched);
  if (rc != HG_OK) return rc;
  if (!batched) {
    // other shapes: each problem's own enqueue-only solve, one behind the other in the context's stream
    for (int i = 0; i < count; ++i)
      if ((rc = hg_problem_solve_async(problems[i], opts)) != HG_OK) return rc;
    return HG_OK;
  }
  // windows sharing their launches hold their last iterations back in a state the context has once: settled here
  // (the call then returns when the windows are nearly solved; single-pose batches return at once)
  return batch_settle(problems[0]->ctx);
}

int hg_register_scan_batch(hg_problem* const* problems, int count, const hg_solver_opts* sopts,
                           const int* pose_index, hg_grid* const* grids, const hg_insert_opts* iopts, int levels,
                           const float* origins, const float* const* xyz, const size_t* n, size_t width,
                           int memspace, double* poses_out, hg_solver_summary* summaries) {
  if (!problems || count < 1 || !pose_index || !grids || !iopts || levels < 1 || !origins || !xyz || !n)
    return HG_ERR_INVALID;
  for (int j = 0; j < count; ++j)
    if (!problems[j] || pose_index[j] < 0 || pose_index[j] >= static_cast<int>(problems[j]->poses.size()))
      return HG_ERR_INVALID;
  if (count > kBatchGroup) {
    for (int j0 = 0; j0 < count; j0 += kBatchGroup) {
      const int m = std::min(kBatchGroup, count - j0);
      const int rc = hg_register_scan_batch(problems + j0, m, sopts, pose_index + j0, grids + j0 * levels, iopts, levels,
                                            origins + 3 * j0, xyz + j0, n + j0, width, memspace,
                                            poses_out ? poses_out + 7 * j0 : nullptr, summaries ? summaries + j0 : nullptr);
      if (rc != HG_OK) return rc;
    }
    return HG_OK;
  }
  HG_REQUIRE_CTX(problems[0]);
  hg_ctx* c = problems[0]->ctx;
  for (int j = 0; j < count; ++j)
    if (problems[j]->ctx != c) return HG_ERR_INVALID;  // one job table, one stream
  int rc = async_status_grids(grids, count * levels);  // an insertion of an earlier step that failed reports here
  if (rc != HG_OK) return rc;
  for (int j = 0; j < count; ++j) apply_scan_width(problems[j], xyz[j], n[j], width);
  bool batched = false;
  if (memspace == HG_DEVICE) {
    rc = solve_batch_enqueue(problems, count, sopts, &batched);
    if (rc != HG_OK) return rc;
  }
  if (!batched) {  // other problem shapes, host points, a single submap: one registration after the other
    for (int j = 0; j < count; ++j) {
      rc = hg_register_scan_mode(problems[j], sopts, pose_index[j], grids + j * levels, iopts, levels, origins + 3 * j,
                                 xyz[j], n[j], width, memspace, HG_INSERT_EXACT, poses_out ? poses_out + 7 * j : nullptr,
                                 summaries ? summaries + j : nullptr);
      if (rc != HG_OK) return rc;
    }
    return HG_OK;
  }
  // insertion of every scan at the pose its solve leaves in device memory, with shared launches
  std::vector<const double*> d_poses(count);
  for (int j = 0; j < count; ++j) d_poses[j] = &problems[j]->d_state->h.x[pose_index[j]][0];
  if ((rc = batch_settle(c)) != HG_OK) return rc;  // (windows hold their last launches back)
  rc = pyramid_insert_jobs(c, count, grids, iopts, levels, origins, xyz, n, width, d_poses.data());
  if (rc == HG_ERR_UNSUPPORTED) {  // insertion options outside the binned path: pyramid by pyramid
    rc = HG_OK;
    for (int j = 0; j < count && rc == HG_OK; ++j) {
      float approx[7];
      for (int k = 0; k < 7; ++k) approx[k] = static_cast<float>(problems[j]->poses[pose_index[j]][k]);
      const uint64_t offsets[2] = {0, n[j]};
      rc = pyramid_insert_impl(grids + j * levels, iopts, levels, origins + 3 * j, xyz[j], offsets, 1, width, approx,
                               d_poses[j], HG_INSERT_EXACT, memspace, nullptr);
    }
  }
  for (int j = 0; j < count; ++j) {
    const int r2 = hg_problem_fetch(problems[j], summaries ? summaries + j : nullptr);
    if (rc == HG_OK) rc = r2;
    if (r2 == HG_OK && poses_out) std::memcpy(poses_out + 7 * j, problems[j]->poses[pose_index[j]].data(), sizeof(double) * 7);
  }
  return rc;
}

int hg_register_scan_sequence(hg_problem* p, const hg_solver_opts* sopts, hg_grid* const* grids,
                              const hg_insert_opts* iopts, int levels, int multi_res, const float* origins,
                              const float* const* xyz, const size_t* n, const double* scaling, size_t width,
                              int memspace, int insert_mode, const double* guesses, int count,
                              int prof_every, double* poses_out, hg_solver_summary* summaries) {
  if (!p || !grids || !iopts || !origins || !xyz || !n || !scaling || !guesses || count < 0) return HG_ERR_INVALID;
  HG_REQUIRE_CTX(p);
  // prof_every > 0 samples kernel durations on some steps; the caller's own hg_prof_enable state comes
  // back on every exit path
  struct ProfRestore {
    hg_ctx* c;
    int saved;
    ~ProfRestore() { c->prof_on = saved; }
  } restore{p->ctx, p->ctx->prof_on};
  // Scans in HOST memory (what a drop-in receives from sensor::RangeData): scan k + 1 is copied to one of
  // two device slots on a copy stream while step k runs on the GPU, started by the host between the
  // enqueue of step k and the wait for its pose; the step itself then works on device pointers. Matching
  // and insertion read the same upload.
  struct Uploader {
    hg_ctx* c;
    const float* const* xyz;
    const size_t* n;
    int count, next;
    static int run(void* self) { return static_cast<Uploader*>(self)->upload(); }
    int upload() {  // scan `next` into slot next & 1
      if (next >= count) return HG_OK;
      const int slot = next & 1;
      const size_t bytes = n[next] * 3 * sizeof(float);
      if (c->ws_seq[slot].bytes < bytes) {
        // (growing a slot frees it: the step that last read it must be done)
        HG_HIP_CHECK(hipStreamSynchronize(c->stream));
        const int rc = c->ws_seq[slot].reserve(bytes);
        if (rc != HG_OK) return rc;
      }
      // through pinned staging: a copy from pageable memory blocks the host for the whole transfer
      // (~330 us per 1.2 MB scan, measured), a memcpy into pinned memory takes a third of that and the
      // DMA behind it is asynchronous. The staging slot's previous DMA (two steps ago) has long finished.
      if (c->pin_seq_bytes[slot] < bytes) {
        HG_HIP_CHECK(hipEventSynchronize(c->ev_up[slot]));
        if (c->pin_seq[slot]) (void)hipHostFree(c->pin_seq[slot]);
        c->pin_seq[slot] = nullptr;
        c->pin_seq_bytes[slot] = 0;
        HG_HIP_CHECK(hipHostMalloc(&c->pin_seq[slot], bytes + bytes / 4 + 256));
        c->pin_seq_bytes[slot] = bytes + bytes / 4 + 256;
      } else {
        HG_HIP_CHECK(hipEventSynchronize(c->ev_up[slot]));
      }
      if (bytes) std::memcpy(c->pin_seq[slot], xyz[next], bytes);
      HG_HIP_CHECK(hipStreamWaitEvent(c->copy_stream, c->ev_used[slot], 0));  // step next - 2 is through with the slot
      if (bytes) HG_HIP_CHECK(hipMemcpyAsync(c->ws_seq[slot].ptr, c->pin_seq[slot], bytes, hipMemcpyHostToDevice, c->copy_stream));
      HG_HIP_CHECK(hipEventRecord(c->ev_up[slot], c->copy_stream));
      ++next;
      return HG_OK;
    }
  } up{p->ctx, xyz, n, count, 0};
  const bool staged = memspace == HG_HOST && count > 0;
  if (staged) {
    hg_ctx* c = p->ctx;
    HG_HIP_CHECK(hipSetDevice(c->device));
    if (!c->copy_stream) {
      HG_HIP_CHECK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
      for (int i = 0; i < 2; ++i) {
        HG_HIP_CHECK(hipEventCreateWithFlags(&c->ev_up[i], hipEventDisableTiming));
        HG_HIP_CHECK(hipEventCreateWithFlags(&c->ev_used[i], hipEventDisableTiming));
      }
    }
    for (int i = 0; i < 2; ++i) {
      HG_HIP_CHECK(hipEventRecord(c->ev_used[i], c->stream));  // both slots free once the stream gets here
      HG_HIP_CHECK(hipEventRecord(c->ev_up[i], c->copy_stream));  // (nothing in flight on the staging slots)
    }
    const int rc = up.upload();
    if (rc != HG_OK) return rc;
  }
  for (int k = 0; k < count; ++k) {
#ifdef HG_HOST_STAMPS
    const auto q0 = std::chrono::steady_clock::now();
#endif
    int rc = hg_problem_reset(p);
    if (rc != HG_OK) return rc;
    const int pi = hg_problem_add_pose(p, guesses + 7 * k, 0);
    if (pi < 0) return pi;
    const float* pts = xyz[k];
    int space = memspace;
#ifdef HG_HOST_STAMPS
    const auto q1 = std::chrono::steady_clock::now();
#endif
    if (staged) {
      HG_HIP_CHECK(hipStreamWaitEvent(p->ctx->stream, p->ctx->ev_up[k & 1], 0));
      pts = static_cast<const float*>(p->ctx->ws_seq[k & 1].ptr);
      space = HG_DEVICE;
    }
#ifdef HG_HOST_STAMPS
    const auto q2 = std::chrono::steady_clock::now();
#endif
    rc = hg_problem_add_block(p, pts, n[k], space, grids, levels, multi_res, scaling[k], pi, -1, 0.0);
    if (rc != HG_OK) return rc;
#ifdef HG_HOST_STAMPS
    {
      const auto q3 = std::chrono::steady_clock::now();
      auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
      if (k % 10 == 5) fprintf(stderr, "sequence loop (us): reset+pose %.1f, wait event %.1f, add block %.1f\n", us(q0, q1), us(q1, q2), us(q2, q3));
    }
#endif
    if (prof_every > 0) p->ctx->prof_on = (k % prof_every) ? 0 : ((k % (5 * prof_every)) ? 2 : 1);
    struct Between {
      Uploader* up;
      hg_ctx* c;
      int slot;
      static int run(void* self) {
        Between* b = static_cast<Between*>(self);
        HG_HIP_CHECK(hipEventRecord(b->c->ev_used[b->slot], b->c->stream));  // solve + insertion of this step are enqueued
        return b->up->upload();
      }
    } between{&up, p->ctx, k & 1};
    rc = register_scan_step(p, sopts, pi, grids, iopts, levels, origins + 3 * k, pts, n[k], width, space, insert_mode,
                            poses_out ? poses_out + 7 * k : nullptr, summaries ? summaries + k : nullptr,
                            staged ? &Between::run : nullptr, &between);
    if (rc != HG_OK) return rc;
  }
  return HG_OK;
}

int hg_register_scan(hg_problem* p, const hg_solver_opts* sopts, int pose_index,
                     hg_grid* const* grids, const hg_insert_opts* iopts, int levels,
                     const float origin[3], const float* xyz, size_t n, size_t width, int memspace,
                     double pose_out[7], hg_solver_summary* summary) {
  return hg_register_scan_mode(p, sopts, pose_index, grids, iopts, levels, origin, xyz, n, width, memspace,
                               HG_INSERT_EXACT, pose_out, summary);
}

int hg_register_scan_mode(hg_problem* p, const hg_solver_opts* sopts, int pose_index,
                          hg_grid* const* grids, const hg_insert_opts* iopts, int levels,
                          const float origin[3], const float* xyz, size_t n, size_t width, int memspace,
                          int insert_mode, double pose_out[7], hg_solver_summary* summary) {
  return register_scan_step(p, sopts, pose_index, grids, iopts, levels, origin, xyz, n, width, memspace, insert_mode,
                            pose_out, summary, nullptr, nullptr);
}

int hg_register_scan_unwarped(hg_problem* p, const hg_solver_opts* sopts, hg_grid* const* grids,
                              const hg_insert_opts* iopts, int levels, const float* points, size_t n,
                              size_t width, int memspace, const hg_timed_cloud* clouds, int n_clouds,
                              const int* pose_index, const int64_t* control_times, int n_control,
                              const float* pose_tq, int insert_mode, double* poses_out,
                              hg_solver_summary* summary) {
  if (!p || !p->ctx || !grids || !iopts || levels < 1 || !pose_index || n_control < 2) return HG_ERR_INVALID;
  for (int k = 0; k < n_control; ++k)
    if (pose_index[k] < 0 || pose_index[k] >= static_cast<int>(p->poses.size())) return HG_ERR_INVALID;
  int rc = async_status_grids(grids, levels);
  if (rc != HG_OK) return rc;
  rc = solve_async_impl(p, sopts, true);
  if (rc != HG_OK) return rc;
  // the unwarping reads the solved control poses where the solve leaves them (LmHead::x, kState doubles per
  // control point); the host's copy (the initial guesses) only sizes the key window. A problem without free
  // parameters launches nothing: its poses are the host's.
  std::vector<double> guess(static_cast<size_t>(n_control) * 7);
  for (int k = 0; k < n_control; ++k) std::memcpy(&guess[7 * k], p->poses[pose_index[k]].data(), sizeof(double) * 7);
  int stride = kState;
  const double* d_poses = device_poses(p, &stride);
  if ((rc = solve_settle(p)) != HG_OK) return rc;  // (a general problem holds its last launches back)
  rc = unwarp_insert(grids, iopts, levels, points, n, width, memspace, clouds, n_clouds, guess.data(), d_poses,
                     d_poses ? pose_index : nullptr, stride, control_times, n_control, pose_tq, insert_mode, nullptr);
  const int rc2 = hg_problem_fetch(p, summary);
  if (rc == HG_OK) rc = rc2;
  if (rc == HG_OK && poses_out)
    for (int k = 0; k < n_control; ++k) std::memcpy(poses_out + 7 * k, p->poses[pose_index[k]].data(), sizeof(double) * 7);
  return rc;
}

}  // extern "C"

// One registration step; `between` (if any) runs on the host after solve and insertion have been enqueued
// and before the wait for the pose: the place to start work for the NEXT step.
static int register_scan_step(hg_problem* p, const hg_solver_opts* sopts, int pose_index,
                              hg_grid* const* grids, const hg_insert_opts* iopts, int levels,
                              const float origin[3], const float* xyz, size_t n, size_t width, int memspace,
                              int insert_mode, double pose_out[7], hg_solver_summary* summary,
                              int (*between)(void*), void* between_arg) {
  if (!p || !grids || !iopts || !origin || pose_index < 0 ||
      pose_index >= static_cast<int>(p->poses.size()))
    return HG_ERR_INVALID;
  HG_REQUIRE_CTX(p);
#ifdef HG_HOST_STAMPS
  static double acc[4] = {0, 0, 0, 0};
  static int calls = 0;
  static std::chrono::steady_clock::time_point last_ret;
  const auto h0 = std::chrono::steady_clock::now();
#endif
  // an insertion of an earlier step that ran out of blocks (or left the index range) reports here:
  // those steps return before their insertion has finished
  if (levels < 1) return HG_ERR_INVALID;
  int rc = async_status_grids(grids, levels);
  if (rc != HG_OK) return rc;
  apply_scan_width(p, xyz, n, width);
  rc = solve_async_impl(p, sopts, true);
  if (rc != HG_OK) return rc;
  if ((rc = solve_settle(p)) != HG_OK) return rc;  // (a general problem holds its last launches back)
#ifdef HG_HOST_STAMPS
  const auto h1 = std::chrono::steady_clock::now();
#endif
  // insertion at the pose the solve leaves in device memory; the host only knows the initial
  // guess, which sizes the key window
  float approx[7];
  for (int k = 0; k < 7; ++k) approx[k] = static_cast<float>(p->poses[pose_index][k]);
  int stride = kState;
  const double* d_pose = device_poses(p, &stride);
  if (d_pose) d_pose += static_cast<size_t>(pose_index) * stride;
  const uint64_t offsets[2] = {0, n};
  rc = pyramid_insert_impl(grids, iopts, levels, origin, xyz, offsets, 1, width, approx, d_pose,
                           insert_mode, memspace, nullptr);
#ifdef HG_HOST_STAMPS
  const auto h2 = std::chrono::steady_clock::now();
#endif
  if (rc == HG_OK && between) rc = between(between_arg);
  const int rc2 = hg_problem_fetch(p, summary);
  if (rc == HG_OK) rc = rc2;
  if (rc == HG_OK && pose_out) std::memcpy(pose_out, p->poses[pose_index].data(), sizeof(double) * 7);
#ifdef HG_HOST_STAMPS
  {
    const auto h3 = std::chrono::steady_clock::now();
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    if (calls > 0) acc[0] += us(last_ret, h0);
    acc[1] += us(h0, h1); acc[2] += us(h1, h2); acc[3] += us(h2, h3);
    last_ret = h3;
    if (++calls % 20 == 0) {
      fprintf(stderr, "host per step (us): outside call %.1f, enqueue solve %.1f, enqueue insert %.1f, wait for pose %.1f\n",
              acc[0] / 20, acc[1] / 20, acc[2] / 20, acc[3] / 20);
      acc[0] = acc[1] = acc[2] = acc[3] = 0;
    }
  }
#endif
  return rc;
}

extern "C" {

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
