// hg_insert.hip — TSDFRangeDataInserter3D::Insert on the device, exact mode.
//
// Reference semantics (mapping/3d/tsdf_range_data_inserter_3d.cc:294-342,395-404,698-737):
// a sequential loop over returns; each return walks num_samples+1 voxels along the ray
// segment [hit - tau, hit + tau] and applies UpdateCell, a read-modify-write that
// re-quantises tsd and weight to uint16 on every update. The result of a voxel depends
// only on that voxel's own ordered update sequence, so the device path
//   1. expands every return into its ordered update records (k_ray_count/k_ray_expand),
//   2. stable-sorts the records by (block key, voxel) — generation order = reference order,
//   3. allocates missing blocks (k_alloc_blocks) and
//   4. lets one thread per voxel run apply its updates sequentially (k_apply_runs),
// which reproduces the reference bit for bit while different voxels proceed in parallel.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include <string.h>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "hg_internal.h"
#include "hg_chain.h"

namespace hg {

struct InsertParams {
  double min_range, max_range;
  float truncation_distance;  // float(rel_trunc(double) * resolution(float))   (:298-299)
  float maximum_weight;       // static_cast<float>(options_.maximum_weight())   (:735)
  float epsilon, sigma;
  int free_space;             // num_free_space_voxels > 0                        (:303)
  int has_pose;
  int project_normals;        // project_sdf_distance_to_scan_normal, CLOUD_STRUCTURE normals (:502-607)
  unsigned vertical_stride;   // normal_computation_vertical_stride
  unsigned horizontal_stride; // normal_computation_horizontal_stride * width
  unsigned width;
};

struct ScanTable {        // per scan of a batch
  unsigned long long begin;  // first point index
  unsigned long long count;  // number of points of the scan
  float origin[3];
  float pose[7];          // t xyz, q wxyz (float Rigid3f)
};

struct Ray {
  int bx, by, bz;  // begin cell
  int dx, dy, dz;  // end - begin
  int n;           // num_samples
  float range;
  float ox, oy, oz;  // origin (grid frame)
  bool valid;
  bool use_normal;   // InsertHitWithNormal: tsd measured along the normal through the hit
  float hx, hy, hz;  // hit
  float nx, ny, nz;  // unit normal
  float ndir;        // normal_direction
};

__device__ inline float norm3(float x, float y, float z) {
  // Eigen fixed-size reduction order: x0 + (x1 + x2)
  return sqrtf(x * x + (y * y + z * z));
}

__device__ inline uint32_t find_scan(const ScanTable* scans, uint32_t n_scans, unsigned long long i) {
  uint32_t lo = 0, hi = n_scans;  // last scan with begin <= i
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (scans[mid].begin <= i) lo = mid; else hi = mid;
  }
  return lo;
}

// The scan of return i inside a workgroup whose first return is wg_first (uniform). The returns of a workgroup almost
// always belong to one scan, so the search runs on the uniform index -- scalar loads through the scalar cache,
// and the table entry arrives in scalar registers -- and only lanes beyond that scan's end search for themselves.
// (Per lane, the binary search was three dependent vector loads in front of everything else a count or scatter
// workgroup does: k_bin_count of an eight-scan chunk 10.6 us per scan against 6.7 with the scan as a kernel argument.)
__device__ inline ScanTable scan_lookup(const ScanTable* scans, uint32_t n_scans, unsigned long long i, unsigned long long wg_first) {
  const uint32_t k = __builtin_amdgcn_readfirstlane(find_scan(scans, n_scans, wg_first));
  ScanTable sc = scans[k];
  if (k + 1u < n_scans && i >= scans[k + 1u].begin) sc = scans[find_scan(scans, n_scans, i)];
  return sc;
}

// Gates of Insert (:703-716) and the setup of InsertHit (:294-317).
__device__ inline Ray ray_setup(const GridView& g, const InsertParams& p, const ScanTable& sc,
                                const float* xyz, unsigned long long i, const uint8_t* gate) {
  Ray r;
  r.valid = false;
  if (gate && gate[i] == 0) return r;  // insertion_ratio decimation (:703-710), precomputed
  float hx = xyz[3 * i], hy = xyz[3 * i + 1], hz = xyz[3 * i + 2];
  float ox = sc.origin[0], oy = sc.origin[1], oz = sc.origin[2];
  if (p.has_pose) {
    transform_point(sc.pose, hx, hy, hz);
    transform_point(sc.pose, ox, oy, oz);
  }
  if (isnan(hx) || isnan(hy) || isnan(hz)) return r;
  const float rx = hx - ox, ry = hy - oy, rz = hz - oz;
  const float r0 = norm3(rx, ry, rz);
  if (static_cast<double>(r0) < p.min_range) return r;
  if (static_cast<double>(r0) > p.max_range) return r;
  const float range = r0;  // same expression in InsertHit
  const float tau = p.truncation_distance;
  r.use_normal = false;
  float b_x, b_y, b_z, e_x, e_y, e_z;
  if (p.project_normals) {
    // CLOUD_STRUCTURE normal from the structured neighbours (:502-607), then InsertHitWithNormal
    // (:197-241). Neighbour indices stay inside the scan [0, count).
    const unsigned long long li = i - sc.begin, cnt = sc.count;
    const float max_range_delta = 1.f * g.resolution / 0.05f;
    auto load = [&](unsigned long long j, float& x, float& y, float& z) {
      x = xyz[3 * (sc.begin + j)]; y = xyz[3 * (sc.begin + j) + 1]; z = xyz[3 * (sc.begin + j) + 2];
      if (p.has_pose) transform_point(sc.pose, x, y, z);
    };
    auto bad = [&](unsigned long long j) {
      float x, y, z;
      load(j, x, y, z);
      if (isnan(x) || isnan(y) || isnan(z)) return true;
      return fabsf(r0 - norm3(x - ox, y - oy, z - oz)) > max_range_delta;
    };
    unsigned long long off = p.vertical_stride;
    while (off > 0 && ((li + off >= cnt) || bad(li + off))) --off;
    const unsigned long long i_vu = li + off;
    off = p.vertical_stride;
    // the reference tests `point_idx - offset < 0` on size_t (:548, always false, then reads out
    // of bounds); like the oracle this treats an index before the scan as "no neighbour"
    while (off > 0 && ((li < off) || bad(li - off))) --off;
    const unsigned long long i_vl = li - off;
    if (i_vl == i_vu || p.width == 0) return r;
    off = p.horizontal_stride;
    while (off > 0 && ((li + off >= cnt) || bad(li + off))) off -= p.width;
    const unsigned long long i_hu = li + off;
    off = p.horizontal_stride;
    while (off > 0 && ((li < off) || bad(li - off))) off -= p.width;
    const unsigned long long i_hl = li - off;
    if (i_hl == i_hu) return r;
    float ax, ay, az, bx2, by2, bz2, cx, cy, cz, dx, dy, dz;
    load(i_hl, ax, ay, az); load(i_hu, bx2, by2, bz2); load(i_vl, cx, cy, cz); load(i_vu, dx, dy, dz);
    const float hx_ = ax - bx2, hy_ = ay - by2, hz_ = az - bz2;
    const float vx_ = cx - dx, vy_ = cy - dy, vz_ = cz - dz;
    if ((hx_ == 0.f && hy_ == 0.f && hz_ == 0.f) || (vx_ == 0.f && vy_ == 0.f && vz_ == 0.f)) return r;
    float nx = hy_ * vz_ - hz_ * vy_, ny = hz_ * vx_ - hx_ * vz_, nz = hx_ * vy_ - hy_ * vx_;
    const float nn = norm3(nx, ny, nz);
    if (nn > 0.f) { nx = nx / nn; ny = ny / nn; nz = nz / nn; }
    if (nx == 0.f && ny == 0.f && nz == 0.f) return r;
    if (range < tau) return r;
    float ndir = 1.f;
    if (nx * rx + (ny * ry + nz * rz) > 0.f) ndir = -1.f;
    const float s = ndir * tau;
    b_x = hx - s * nx; b_y = hy - s * ny; b_z = hz - s * nz;
    e_x = hx + s * nx; e_y = hy + s * ny; e_z = hz + s * nz;
    r.use_normal = true;
    r.hx = hx; r.hy = hy; r.hz = hz;
    r.nx = nx; r.ny = ny; r.nz = nz;
    r.ndir = ndir;
  } else {
    if (range < tau) return r;
    const float ratio = tau / range;
    if (p.free_space) {
      b_x = ox; b_y = oy; b_z = oz;
    } else {
      const float s = 1.0f - ratio;
      b_x = ox + s * rx; b_y = oy + s * ry; b_z = oz + s * rz;
    }
    const float e = 1.0f + ratio;
    e_x = ox + e * rx; e_y = oy + e * ry; e_z = oz + e * rz;
  }
  // GetCellIndex of both ends (:311-312): six divisions by the resolution, spelled out with one refined
  // reciprocal (cell_index_fast: the bits of the IEEE division) unless a coordinate of the wavefront is
  // beyond anything a grid can index
  const bool tame = fabsf(b_x) < 1e30f && fabsf(b_y) < 1e30f && fabsf(b_z) < 1e30f &&
                    fabsf(e_x) < 1e30f && fabsf(e_y) < 1e30f && fabsf(e_z) < 1e30f;
  if (__ballot(!tame) == 0ull) {
    const float rr = refined_rcp(g.resolution);
    r.bx = cell_index_fast(b_x, g.resolution, rr);
    r.by = cell_index_fast(b_y, g.resolution, rr);
    r.bz = cell_index_fast(b_z, g.resolution, rr);
    r.dx = cell_index_fast(e_x, g.resolution, rr) - r.bx;
    r.dy = cell_index_fast(e_y, g.resolution, rr) - r.by;
    r.dz = cell_index_fast(e_z, g.resolution, rr) - r.bz;
  } else {
    r.bx = cell_index_1d(b_x, g.resolution);
    r.by = cell_index_1d(b_y, g.resolution);
    r.bz = cell_index_1d(b_z, g.resolution);
    r.dx = cell_index_1d(e_x, g.resolution) - r.bx;
    r.dy = cell_index_1d(e_y, g.resolution) - r.by;
    r.dz = cell_index_1d(e_z, g.resolution) - r.bz;
  }
  r.n = max(abs(r.dx), max(abs(r.dy), abs(r.dz)));
  r.range = range;
  r.ox = ox; r.oy = oy; r.oz = oz;
  r.valid = r.n > 0 && r.n < (1 << 15);
  return r;
}

// tsd and weight of the update of cell (cx, cy, cz) along ray r (:318-342).
// UNIT: the caller has established weight_function_epsilon >= 1 (the binned paths): tsd / tau >= -1 >= -epsilon for the
// clamped tsd (a correctly rounded quotient of |a| <= |b| never exceeds 1), so the weight is 1 and neither the division
// nor the comparison is evaluated. APPROX (tolerance mode only): the distance by the hardware's square root estimate
// (1 ulp) instead of the correctly rounded one.
template <bool UNIT = false, bool APPROX = false>
__device__ inline void ray_sample_cell(const GridView& g, const InsertParams& p, const Ray& r,
                                       int cx, int cy, int cz, float& tsd, float& weight);
__device__ inline void ray_sample(const GridView& g, const InsertParams& p, const Ray& r, int pos,
                                  int& cx, int& cy, int& cz, float& tsd, float& weight) {
  const float fp = static_cast<float>(pos), fn = static_cast<float>(r.n);
  cx = r.bx + static_cast<int>(roundf(static_cast<float>(r.dx) * fp / fn));
  cy = r.by + static_cast<int>(roundf(static_cast<float>(r.dy) * fp / fn));
  cz = r.bz + static_cast<int>(roundf(static_cast<float>(r.dz) * fp / fn));
  ray_sample_cell(g, p, r, cx, cy, cz, tsd, weight);
}
template <bool UNIT, bool APPROX>
__device__ inline void ray_sample_cell(const GridView& g, const InsertParams& p, const Ray& r,
                                       int cx, int cy, int cz, float& tsd, float& weight) {
  const float ccx = static_cast<float>(cx) * g.resolution;
  const float ccy = static_cast<float>(cy) * g.resolution;
  const float ccz = static_cast<float>(cz) * g.resolution;
  const float tau = p.truncation_distance;
  weight = 1.0f;
  if (r.use_normal) {  // :235-239
    const float dx = ccx - r.hx, dy = ccy - r.hy, dz = ccz - r.hz;
    tsd = clampf(r.ndir * (dx * r.nx + (dy * r.ny + dz * r.nz)), -tau, tau);
    return;
  }
  const float ex = ccx - r.ox, ey = ccy - r.oy, ez = ccz - r.oz;
  const float dist = APPROX ? __builtin_amdgcn_sqrtf(ex * ex + (ey * ey + ez * ez)) : norm3(ex, ey, ez);
  tsd = clampf(r.range - dist, -tau, tau);
  if (UNIT) return;
  const float normalized = tsd / tau;
  if (normalized < -p.epsilon) {
    // :333-340, evaluated in double as std::exp/std::pow promote
    const double d = static_cast<double>(-normalized - p.epsilon);
    weight = static_cast<float>(exp(static_cast<double>(-p.sigma) * (d * d)));
  }
}

__global__ void k_ray_count(GridView g, InsertParams p, const ScanTable* scans, uint32_t n_scans,
                            const float* xyz, unsigned long long n, const uint8_t* gate,
                            uint32_t* counts) {
  const unsigned long long i = blockIdx.x * static_cast<unsigned long long>(blockDim.x) + threadIdx.x;
  bool hit = false;
  if (i < n) {
    const ScanTable& sc = scans[find_scan(scans, n_scans, i)];
    const Ray r = ray_setup(g, p, sc, xyz, i, gate);
    counts[i] = r.valid ? static_cast<uint32_t>(r.n + 1) : 0u;
    hit = r.valid;
  }
  __shared__ unsigned wg_hits;
  if (threadIdx.x == 0) wg_hits = 0;
  __syncthreads();
  const unsigned long long m = __ballot(hit);
  if ((threadIdx.x & (kWave - 1)) == 0 && m) atomicAdd(&wg_hits, static_cast<unsigned>(__popcll(m)));
  __syncthreads();
  if (threadIdx.x == 0 && wg_hits) atomicAdd(&g.counters[2], wg_hits);
}

// key = block_key << 9 | voxel ; value = tsd bits | weight bits << 32
__global__ void k_ray_expand(GridView g, InsertParams p, const ScanTable* scans, uint32_t n_scans,
                             const float* xyz, unsigned long long n, const uint8_t* gate,
                             const unsigned long long* offsets, unsigned long long* keys,
                             unsigned long long* vals) {
  const unsigned long long i = blockIdx.x * static_cast<unsigned long long>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const ScanTable& sc = scans[find_scan(scans, n_scans, i)];
  const Ray r = ray_setup(g, p, sc, xyz, i, gate);
  if (!r.valid) return;
  unsigned long long o = offsets[i];
  for (int pos = 0; pos <= r.n; ++pos, ++o) {
    int cx, cy, cz;
    float tsd, w;
    ray_sample(g, p, r, pos, cx, cy, cz, tsd, w);
    unsigned long long key = ~0ull;  // dropped records sort last
    if (cell_in_range(cx, cy, cz)) {
      if (w != 0.f) key = (block_key(cx, cy, cz) << 9) | voxel_in_block(cx, cy, cz);  // :728
    } else {
      atomicOr(&g.counters[1], kFlagRange);
    }
    keys[o] = key;
    vals[o] = static_cast<unsigned long long>(__float_as_uint(tsd)) |
              (static_cast<unsigned long long>(__float_as_uint(w)) << 32);
  }
}

__global__ void k_alloc_blocks(GridView g, const unsigned long long* keys, unsigned long long n) {
  const unsigned long long i = blockIdx.x * static_cast<unsigned long long>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  if (k == ~0ull) return;
  if (i > 0 && (keys[i - 1] >> 9) == (k >> 9)) return;
  insert_block_unique(g, k >> 9);
}

// UpdateCell (:725-737) + SetCell (hybrid_grid_tsdf.h:87-92) on raw codes.
__device__ inline uint32_t update_cell(const GridView& g, float maximum_weight, uint32_t code,
                                       float update_sdf, float update_weight) {
  const float old_weight = value_to_weight(g, code >> 16);
  const float old_sdf = value_to_tsd(g, code & 0xFFFFu);
  float updated_weight = old_weight + update_weight;
  const float updated_sdf = (old_sdf * old_weight + update_sdf * update_weight) / updated_weight;
  updated_weight = (maximum_weight < updated_weight) ? maximum_weight : updated_weight;  // std::min
  return (tsd_to_value(g, updated_sdf) + kUpdateMarker) | (weight_to_value(g, updated_weight) << 16);
}

// (the exact per-voxel update chain -- UnitChain, chain_run, seg_chains -- lives in hg_chain.h)

__global__ void k_apply_runs(GridView g, InsertParams p, const unsigned long long* keys,
                             const unsigned long long* vals, unsigned long long n) {
  const unsigned long long i = blockIdx.x * static_cast<unsigned long long>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  if (k == ~0ull) return;
  if (i > 0 && keys[i - 1] == k) return;  // not a run head
  const uint32_t slot = find_block(g, k >> 9);
  if (slot >= g.pool_blocks) return;  // capacity exceeded (flag already set)
  uint32_t* cell = g.voxels + static_cast<size_t>(slot) * kVoxelsPerBlock + (k & 511u);
  uint32_t code = *cell;
  uint32_t applied = 0;
  for (unsigned long long j = i; j < n && keys[j] == k; ++j) {
    const unsigned long long v = vals[j];
    code = update_cell(g, p.maximum_weight, code, __uint_as_float(static_cast<uint32_t>(v)),
                       __uint_as_float(static_cast<uint32_t>(v >> 32)));
    ++applied;
  }
  *cell = code;
  atomicAdd(reinterpret_cast<unsigned long long*>(&g.counters[4]), static_cast<unsigned long long>(applied));
}


// ==========================================================================================
// Fused pyramid path: all levels of a pyramid in one expand / sort / alloc / apply sequence,
// fixed 8 record slots per (return, level) so no prefix scan and no host read-back is needed.
// ==========================================================================================
constexpr int kMaxInsLevels = 4;
constexpr int kSlots = 8;  // >= num_samples + 1 for relative_truncation_distance <= 3

struct LevelIns {
  GridView g;
  InsertParams p;
  int base[3];          // block coordinate the 32-bit keys are relative to
  const uint8_t* gate;  // insertion_ratio decimation mask or nullptr
  // deferred long chains of the binned apply pass (k_bin_apply, "Long chains" there), or null: the ordered update
  // values of the deferred voxels (bump-allocated through g.call[3]) and, per apply workgroup, its list of them
  uint32_t* heavy_vals;
  uint4* heavy_list;        // kHeavyPerWg entries {cell, offset, updates, -} per workgroup of the level's grid row
  uint32_t heavy_capacity;  // values heavy_vals holds (>= the level's records of the call: nothing can overflow)
};
// The kernels over a job table read their pointers from device memory, where nothing tells the compiler what they point
// to: every access through them became a FLAT operation, which counts against the LDS / scalar counter as well as the
// vector-memory one -- a wait for an LDS read then waits for the loads and stores in flight (k_stream_units, round 6:
// an owner's 64 loads one round trip after the other). Such a kernel passes its pointers through the device-memory
// address space once (a cast there and back: the compiler then follows the pointer as a global one).
template <typename T>
__device__ __forceinline__ T* as_device(T* p) {  // p: uniform (a word of the job table)
  __attribute__((address_space(1))) T* g = (__attribute__((address_space(1))) T*)p;
  asm volatile("" : "+s"(g));  // (keeps the pair of casts from being folded away)
  return (T*)g;
}
__device__ __forceinline__ LevelIns level_as_device(LevelIns L) {
  L.g.table = as_device(L.g.table); L.g.voxels = as_device(L.g.voxels); L.g.block_keys = as_device(L.g.block_keys);
  L.g.block_list = as_device(L.g.block_list); L.g.counters = as_device(L.g.counters); L.g.call = as_device(L.g.call);
  L.g.bin_count = as_device(L.g.bin_count); L.g.bin_offset = as_device(L.g.bin_offset); L.g.touched = as_device(L.g.touched);
  L.g.work = as_device(L.g.work); L.gate = as_device(L.gate); L.heavy_vals = as_device(L.heavy_vals);
  L.heavy_list = as_device(L.heavy_list);
  return L;
}
struct PyramidIns {
  LevelIns lv[kMaxInsLevels];
  int levels;
  ScanTable scan0;  // the scan when a call carries exactly one (no table upload needed)
  const double* d_pose;  // optional: pose (t xyz, q wxyz, fp64) in device memory, e.g. the pose a
                         // solve left there; cast to float as Rigid3d::cast<float>() does
  const float* d_origin; // optional: origin of the scan in device memory (the per-point unwarping leaves it
                         // there, hg_unwarp.hip), read instead of scan0.origin
  int accumulate;        // not the first chunk of a call: hit / update counters add up
  uint32_t* host_flags;  // the context's mapped pinned flag words: sticky error flags of calls that do not
                         // read their stats back (written only when a flag is set), one word per grid
  uint16_t flag_slot[kMaxInsLevels];  // word of each level's grid
  int slice_records;     // records per voxel slice of a large bin (0 = by bin size), see k_bin_offsets
  int shared;            // several scans in flight on the same grids (scan stream): per-call statistics are
                         // added atomically
};

// One thread per level: hands the level's sticky error flags to the host without a read-back.
__device__ inline void publish_flags(const PyramidIns& P, int level) {
  const uint32_t f = P.lv[level].g.counters[1];
  if (f != 0u && P.host_flags) {
    P.host_flags[P.flag_slot[level]] = f;
    __threadfence_system();
  }
}
__global__ void k_publish_flags(PyramidIns P) { publish_flags(P, threadIdx.x); }

enum : uint32_t { kFlagStride = 4u };

// Record key: sorts by (level, block, voxel); equal keys keep generation order (stable sort).
template <typename K> struct KeyCodec;
template <> struct KeyCodec<uint32_t> {  // [level:2][rz:7][ry:7][rx:7][voxel:9], level 3 = invalid
  static constexpr uint32_t kInvalid = 0xFFFFFFFFu;
  static constexpr unsigned kBits = 32;
  __device__ static uint32_t make(const LevelIns& L, int level, int cx, int cy, int cz, bool* range_err) {
    const int rx = ((cx + kIndexOffset) >> 3) - L.base[0] + 64;
    const int ry = ((cy + kIndexOffset) >> 3) - L.base[1] + 64;
    const int rz = ((cz + kIndexOffset) >> 3) - L.base[2] + 64;
    if ((static_cast<unsigned>(rx) | static_cast<unsigned>(ry) | static_cast<unsigned>(rz)) > 127u) {
      *range_err = true;
      return kInvalid;
    }
    return (static_cast<uint32_t>(level) << 30) | (static_cast<uint32_t>(rz) << 23) |
           (static_cast<uint32_t>(ry) << 16) | (static_cast<uint32_t>(rx) << 9) | voxel_in_block(cx, cy, cz);
  }
  __device__ static bool valid(uint32_t k) { return (k >> 30) != 3u; }
  __device__ static int level(uint32_t k) { return static_cast<int>(k >> 30); }
  __device__ static unsigned long long block(const LevelIns& L, uint32_t k) {
    const unsigned long long bx = static_cast<unsigned long long>(static_cast<int>((k >> 9) & 127u) - 64 + L.base[0]);
    const unsigned long long by = static_cast<unsigned long long>(static_cast<int>((k >> 16) & 127u) - 64 + L.base[1]);
    const unsigned long long bz = static_cast<unsigned long long>(static_cast<int>((k >> 23) & 127u) - 64 + L.base[2]);
    return (bz << 22) | (by << 11) | bx;
  }
};
template <> struct KeyCodec<unsigned long long> {  // [level:2][block key:33][voxel:9]
  static constexpr unsigned long long kInvalid = ~0ull;
  static constexpr unsigned kBits = 45;
  __device__ static unsigned long long make(const LevelIns&, int level, int cx, int cy, int cz, bool*) {
    return (static_cast<unsigned long long>(level) << 42) | (block_key(cx, cy, cz) << 9) | voxel_in_block(cx, cy, cz);
  }
  __device__ static bool valid(unsigned long long k) { return k != kInvalid; }
  __device__ static int level(unsigned long long k) { return static_cast<int>((k >> 42) & 3ull); }
  __device__ static unsigned long long block(const LevelIns&, unsigned long long k) {
    return (k >> 9) & ((1ull << 33) - 1ull);
  }
};

// Record value: tsd only (update weight is provably 1: weight_function_epsilon >= 1) or tsd + weight.
template <typename V> struct ValCodec;
template <> struct ValCodec<uint32_t> {
  __device__ static uint32_t make(float tsd, float) { return __float_as_uint(tsd); }
  __device__ static float tsd(uint32_t v) { return __uint_as_float(v); }
  __device__ static float weight(uint32_t) { return 1.0f; }
};
template <> struct ValCodec<unsigned long long> {
  __device__ static unsigned long long make(float tsd, float w) {
    return static_cast<unsigned long long>(__float_as_uint(tsd)) |
           (static_cast<unsigned long long>(__float_as_uint(w)) << 32);
  }
  __device__ static float tsd(unsigned long long v) { return __uint_as_float(static_cast<uint32_t>(v)); }
  __device__ static float weight(unsigned long long v) { return __uint_as_float(static_cast<uint32_t>(v >> 32)); }
};

// grid (ceil(n/256), levels): one thread per (return, level) writes its 8 record slots.
template <typename K, typename V>
__global__ __launch_bounds__(256) void k_expand_fixed(PyramidIns P, const ScanTable* scans,
                                                      uint32_t n_scans, const float* xyz,
                                                      unsigned long long n, K* keys, V* vals,
                                                      unsigned* wg_hits) {
  const int level = blockIdx.y;
  const LevelIns& L = P.lv[level];
  const unsigned long long i = blockIdx.x * 256ull + threadIdx.x;
  __shared__ unsigned s_hits;
  if (threadIdx.x == 0) s_hits = 0;
  __syncthreads();
  bool hit = false;
  if (i < n) {
    ScanTable sc = (n_scans == 1) ? P.scan0 : scans[find_scan(scans, n_scans, i)];
    if (P.d_origin) { sc.origin[0] = P.d_origin[0]; sc.origin[1] = P.d_origin[1]; sc.origin[2] = P.d_origin[2]; }
    if (P.d_pose) {
#pragma unroll
      for (int k = 0; k < 7; ++k) sc.pose[k] = static_cast<float>(P.d_pose[k]);
    }
    const Ray r = ray_setup(L.g, L.p, sc, xyz, i, L.gate);
    K kk[kSlots];
    V vv[kSlots];
    bool range_err = false, stride_err = false;
    if (r.valid && r.n + 1 > kSlots) stride_err = true;
    hit = r.valid && !stride_err;
#pragma unroll
    for (int pos = 0; pos < kSlots; ++pos) {
      kk[pos] = KeyCodec<K>::kInvalid;
      vv[pos] = 0;
      if (hit && pos <= r.n) {
        int cx, cy, cz;
        float tsd, w;
        ray_sample(L.g, L.p, r, pos, cx, cy, cz, tsd, w);
        if (!cell_in_range(cx, cy, cz)) range_err = true;
        else if (w != 0.f) kk[pos] = KeyCodec<K>::make(L, level, cx, cy, cz, &range_err);
        vv[pos] = ValCodec<V>::make(tsd, w);
      }
    }
    if (range_err) atomicOr(&L.g.counters[1], kFlagRange);
    if (stride_err) atomicOr(&L.g.counters[1], kFlagStride);
    const unsigned long long o = (static_cast<unsigned long long>(level) * n + i) * kSlots;
    typedef K __attribute__((ext_vector_type(16 / sizeof(K)))) KVec;
    typedef V __attribute__((ext_vector_type(16 / sizeof(V)))) VVec;
    constexpr int kPerK = 16 / sizeof(K), kPerV = 16 / sizeof(V);
    KVec* ko = reinterpret_cast<KVec*>(keys + o);
    VVec* vo = reinterpret_cast<VVec*>(vals + o);
#pragma unroll
    for (int c = 0; c < kSlots / kPerK; ++c) {
      KVec t;
#pragma unroll
      for (int e = 0; e < kPerK; ++e) t[e] = kk[c * kPerK + e];
      ko[c] = t;
    }
#pragma unroll
    for (int c = 0; c < kSlots / kPerV; ++c) {
      VVec t;
#pragma unroll
      for (int e = 0; e < kPerV; ++e) t[e] = vv[c * kPerV + e];
      vo[c] = t;
    }
  }
  const unsigned long long m = __ballot(hit);
  if ((threadIdx.x & (kWave - 1)) == 0 && m) atomicAdd(&s_hits, static_cast<unsigned>(__popcll(m)));
  __syncthreads();
  if (threadIdx.x == 0) wg_hits[level * gridDim.x + blockIdx.x] = s_hits;
}

// After the sort each distinct (level, block) starts exactly one run: that record's thread
// inserts the block (no two threads insert the same key).
template <typename K>
__global__ void k_alloc_sorted(PyramidIns P, const K* keys, unsigned long long n) {
  const unsigned long long i = blockIdx.x * static_cast<unsigned long long>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const K k = keys[i];
  if (!KeyCodec<K>::valid(k)) return;
  if (i > 0 && (keys[i - 1] >> 9) == (k >> 9)) return;
  const LevelIns& L = P.lv[KeyCodec<K>::level(k)];
  insert_block_unique(L.g, KeyCodec<K>::block(L, k));
}

// One wavefront per 64 sorted records. Every run head applies its run sequentially; record
// values reach the head lane by cross-lane shuffles (coalesced loads, no per-update memory
// latency). A run that continues past the chunk is finished by its head lane from further
// 64-record chunks.
template <typename K, typename V>
__global__ __launch_bounds__(256) void k_apply_wave(PyramidIns P, const K* __restrict__ keys,
                                                    const V* __restrict__ vals, unsigned long long n,
                                                    unsigned* wg_updates /* [gridDim.x][4] */) {
  __shared__ unsigned s_upd[kMaxInsLevels];
  if (threadIdx.x < kMaxInsLevels) s_upd[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & (kWave - 1);
  const unsigned long long base = (blockIdx.x * 4ull + (threadIdx.x >> 6)) * kWave;
  if (base < n) {
    const unsigned long long i = base + lane;
    const K k = (i < n) ? keys[i] : KeyCodec<K>::kInvalid;
    const V v = (i < n) ? vals[i] : V(0);
    const bool valid = KeyCodec<K>::valid(k);
    K kprev = __shfl_up(k, 1);
    bool first = false;
    if (lane == 0) {
      if (base == 0) first = true; else kprev = keys[base - 1];
    }
    const bool head = valid && (first || k != kprev);
    const unsigned long long heads = __ballot(head);
    const unsigned long long valids = __ballot(valid);
    const int nv = __popcll(valids);  // invalid records sort last: valid lanes are a prefix
    const int lvl = valid ? KeyCodec<K>::level(k) : 0;
    for (int l = 0; l < P.levels; ++l) {
      const unsigned long long ml = __ballot(valid && lvl == l);
      if (lane == 0 && ml) atomicAdd(&s_upd[l], static_cast<unsigned>(__popcll(ml)));
    }
    int len = 0;
    uint32_t* cell = nullptr;
    uint32_t code = 0;
    const LevelIns& L = P.lv[lvl];
    constexpr bool kUnit = sizeof(V) == 4;  // 32-bit values: every update weight is 1
    UnitChain chain;
    if (head) {
      const unsigned long long rest = (lane == 63) ? 0ull : (heads >> (lane + 1));
      const int next = rest ? lane + 1 + __builtin_ctzll(rest) : kWave;
      len = min(next, nv) - lane;
      const uint32_t slot = find_block(L.g, KeyCodec<K>::block(L, k));
      if (slot < L.g.pool_blocks) {
        cell = L.g.voxels + static_cast<size_t>(slot) * kVoxelsPerBlock + (static_cast<uint32_t>(k) & 511u);
        code = *cell;
      }
    }
    chain.begin(L.g, code);
    for (int s = 0; __ballot(head && s < len); ++s) {
      const V vs = __shfl(v, lane + s);
      if (head && s < len) {
        if (kUnit) chain.step(L.g, L.p.maximum_weight, ValCodec<V>::tsd(vs));
        else code = update_cell(L.g, L.p.maximum_weight, code, ValCodec<V>::tsd(vs), ValCodec<V>::weight(vs));
      }
    }
    // continuation of the chunk's last run into the following chunks
    if (heads != 0ull && nv == kWave) {
      const int hl = 63 - __builtin_clzll(heads);
      const K krun = __shfl(k, 63);
      unsigned long long nb = base + kWave;
      bool raised = false;
      while (nb < n) {
        const unsigned long long i2 = nb + lane;
        const K k2 = (i2 < n) ? keys[i2] : KeyCodec<K>::kInvalid;
        const V v2 = (i2 < n) ? vals[i2] : V(0);
        const unsigned long long mm = __ballot(k2 == krun);
        const int cnt = (mm == ~0ull) ? kWave : __builtin_ctzll(~mm);
        if (cnt == kWave && !raised) {  // a long per-voxel chain is the critical path of the call
          __builtin_amdgcn_s_setprio(3);
          raised = true;
        }
        for (int j = 0; j < cnt; ++j) {
          const V vj = __shfl(v2, j);
          if (lane == hl) {
            if (kUnit) chain.step(L.g, L.p.maximum_weight, ValCodec<V>::tsd(vj));
            else code = update_cell(L.g, L.p.maximum_weight, code, ValCodec<V>::tsd(vj), ValCodec<V>::weight(vj));
          }
        }
        if (cnt < kWave) break;
        nb += kWave;
      }
      if (raised) __builtin_amdgcn_s_setprio(0);
    }
    if (kUnit) code = chain.end();
    if (head && cell) *cell = code;
  }
  __syncthreads();
  if (threadIdx.x < kMaxInsLevels) wg_updates[blockIdx.x * kMaxInsLevels + threadIdx.x] = s_upd[threadIdx.x];
}

// Adds the per-workgroup hit / update counts of one call into each level's grid counters.
__global__ void k_sum_stats(PyramidIns P, const unsigned* wg_hits, unsigned n_expand_wg,
                            const unsigned* wg_updates, unsigned n_apply_wg) {
  __shared__ unsigned long long red[256];
  for (int l = 0; l < P.levels; ++l) {
    unsigned long long h = 0, u = 0;
    for (unsigned i = threadIdx.x; i < n_expand_wg; i += blockDim.x) h += wg_hits[l * n_expand_wg + i];
    for (unsigned i = threadIdx.x; i < n_apply_wg; i += blockDim.x) u += wg_updates[i * kMaxInsLevels + l];
    red[threadIdx.x] = h;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    const unsigned long long hs = red[0];
    __syncthreads();
    red[threadIdx.x] = u;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      unsigned long long* upd = reinterpret_cast<unsigned long long*>(&P.lv[l].g.counters[4]);
      P.lv[l].g.counters[2] = static_cast<uint32_t>(hs) + (P.accumulate ? P.lv[l].g.counters[2] : 0u);
      if (wg_updates) *upd = red[0] + (P.accumulate ? *upd : 0ull);
    }
    __syncthreads();
  }
}


// ==========================================================================================
// Binned path (single scan, unit update weight): no global sort.
//   k_bin_count    per (return, level): walks the ray, inserts the blocks it touches into the
//                  hash, counts its records per block bin (one atomic per run of samples inside
//                  one block) and collects the touched blocks.
//   k_bin_offsets  per level: exclusive scan of the touched bins' counts -> bin offsets.
//   k_bin_scatter  per (return, level): walks the ray again and writes {voxel | seq, tsd} records
//                  into its blocks' bins (arrival order inside a bin is arbitrary).
//   k_bin_apply    one workgroup per touched block: groups the bin by voxel in LDS (counting
//                  sort), orders every voxel's records by seq (rank counting), then one thread
//                  per voxel applies its updates in reference order on the block's voxels.
// seq = return index * 8 + sample position restores the reference's update order exactly.
// ==========================================================================================
constexpr int kBinCap = 2048;       // records per LDS pass of k_bin_apply (4 per thread)
constexpr int kBinThreads = 512;    // one thread per voxel of a block
constexpr unsigned kSmallBin = 512; // a bin of at most this many records is one wavefront's work (k_bin_apply_small)
constexpr unsigned kSmallBinInKernel = 256;  // the same inside k_bin_apply
#ifndef HG_APPLY_WAVES
#define HG_APPLY_WAVES 8
#endif
#ifndef HG_SLICE_THRESH  // (tuning switches of the single-chain slice sizes)
#define HG_SLICE_THRESH 4096u
#endif
#ifndef HG_SLICE_BELOW
#define HG_SLICE_BELOW 2048u
#endif
#ifndef HG_SLICE_ABOVE
// (512 until late in round 6; measured on the headline, one box, interleaved: 512 -> 4606 - 4628 scans/s, 640 -> 4681 - 4705,
// 768 -> 4662 - 4680, 896 -> 4500 - 4511, 1024 -> 4323 - 4335: every slice of a bin reads the whole bin, so fewer slices of the
// mid-size bins are fewer re-reads, until a slice no longer fits one LDS pass)
#define HG_SLICE_ABOVE 640u
#endif
  // bins from this size on head the work list (their slices carry the long chains)
constexpr unsigned kSeqBits = 23;   // seq < 2^23: at most 2^20 returns per call on this path
// HG_DEFER_LONG_CHAINS (off by default, see "Long chains" in bin_apply_body and DESIGN 3.1): compile the deferral of
// long per-voxel chains and their segmented evaluation into k_bin_apply.
#ifndef HG_HEAVY_MIN
#define HG_HEAVY_MIN 512u
#endif
constexpr unsigned kHeavyMin = HG_HEAVY_MIN;      // updates of one voxel in one pass from which its chain is deferred
constexpr unsigned kHeavyPerWg = 16;     // deferred voxels per apply workgroup (more: applied in place as before)
constexpr unsigned kHeavyLdsVals = 8192; // update values the tail of k_bin_apply holds in LDS at a time

enum : uint32_t { kFlagBinOverflow = 8u, kFlagWorkOverflow = 16u };

__device__ inline ScanTable scan_of(const PyramidIns& P) {
  ScanTable sc = P.scan0;
  if (P.d_origin) { sc.origin[0] = P.d_origin[0]; sc.origin[1] = P.d_origin[1]; sc.origin[2] = P.d_origin[2]; }
  if (P.d_pose) {
#pragma unroll
    for (int k = 0; k < 7; ++k) sc.pose[k] = static_cast<float>(P.d_pose[k]);
  }
  return sc;
}

constexpr int kMaxRuns = 4;  // a straight 8-sample walk visits at most 4 blocks (monotone per axis)

// The sample cells of a SHORT ray (n <= 7: the binned path) without floating point. The reference rounds
// fl(fl(d * pos) / n) half away from zero (:318-320; ray_sample). For |d| <= n <= 7, pos <= 7 the
// product is a small integer, exact in fp32, and the quotient of two such integers is either exactly k + 1/2
// -- representable, so the division returns it and the tie goes away from zero -- or at least 1/(2n) >= 1/14
// from any tie, which no rounding of the division can bridge. Hence the offset is sign(d) * floor((2 |d| pos
// + n) / (2 n)) in exact integer arithmetic, and along pos = 0, 1, ... it is a digital differential: the
// remainder grows by 2 |d| <= 2 n per step and carries at most once. Twelve integer operations per sample
// instead of three fp32 divisions and three roundf (~ 57): the front end of a scan stream and of batched
// registration is bound by instruction issue.
struct RayWalk {
  int c[3];    // cell of the current position
  int rem[3];  // (2 |d| pos + n) mod 2 n
  int a[3];    // 2 |d|
  int s[3];    // sign(d)
  int n2;
  __device__ inline void begin(const Ray& r) {
    const int d[3] = {r.dx, r.dy, r.dz};
    c[0] = r.bx; c[1] = r.by; c[2] = r.bz;
    n2 = 2 * r.n;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      rem[k] = r.n;
      a[k] = 2 * abs(d[k]);
      s[k] = d[k] < 0 ? -1 : 1;
    }
  }
  __device__ inline void step() {  // pos -> pos + 1
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      rem[k] += a[k];
      const bool carry = rem[k] >= n2;
      rem[k] -= carry ? n2 : 0;
      c[k] += carry ? s[k] : 0;
    }
  }
};

// Splits the samples of a ray into runs that stay inside one block. Returns the number of runs.
// (Sample by sample: the form for rays that leave the index range; ray_block_runs below.)
__device__ inline int ray_block_runs_walk(const Ray& r, unsigned long long* run_key, int* run_begin,
                                                             int* run_len, bool* range_err) {
  int nr = 0;
  RayWalk w;
  w.begin(r);
  bool open = false;   // the previous sample lies in range (its run can be continued)
  int pbx = 0, pby = 0, pbz = 0;  // block coordinates of the previous sample
  for (int pos = 0; pos <= r.n; ++pos) {
    const int cx = w.c[0], cy = w.c[1], cz = w.c[2];
    w.step();
    if (!cell_in_range(cx, cy, cz)) {
      *range_err = true;
      open = false;
      continue;
    }
    const int bx = (cx + kIndexOffset) >> 3, by = (cy + kIndexOffset) >> 3, bz = (cz + kIndexOffset) >> 3;
    if (!open || bx != pbx || by != pby || bz != pbz) {
      if (nr == kMaxRuns) break;  // cannot happen for a straight walk; guards the arrays
      run_key[nr] = block_key(cx, cy, cz);
      run_begin[nr] = pos;
      run_len[nr] = 0;
      ++nr;
      open = true;
      pbx = bx; pby = by; pbz = bz;
    }
    ++run_len[nr - 1];
  }
  return nr;
}

// The same without walking the samples, for a SHORT ray (n <= 7) whose two ends lie in the index range (then
// every sample does): a coordinate moves |d| <= 7 cells monotonically, so it crosses at most ONE block face,
// at the first position whose offset floor((2 |d| pos + n) / 2 n) reaches t = cells up to the face:
// pos = ceil(n (2 t - 1) / 2 |d|). The runs start at position 0 and at the distinct crossing positions; a
// run's block key is the previous one stepped along the axes that cross there (~ 130 operations instead of
// ~ 50 per sample: k_bin_count of a scan stream is bound by instruction issue).
__device__ inline int ray_block_runs(const Ray& r, unsigned long long* run_key, int* run_begin,
                                     int* run_len, bool* range_err) {
  const bool inside = cell_in_range(r.bx, r.by, r.bz) && cell_in_range(r.bx + r.dx, r.by + r.dy, r.bz + r.dz);
  if (__ballot(!inside) != 0ull) return ray_block_runs_walk(r, run_key, run_begin, run_len, range_err);
  const int b[3] = {r.bx, r.by, r.bz}, d[3] = {r.dx, r.dy, r.dz};
  int cross[3];  // position of the axis' block crossing, 8 = none
  unsigned chg = 1u;  // bit pos: a run starts at pos
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int ad = abs(d[k]);
    const int u = (b[k] + kIndexOffset) & 7;
    const int t = d[k] > 0 ? 8 - u : u + 1;  // cells to move until the coordinate leaves its block
    // ceil(x / y), x = n (2 t - 1) <= 105, y = 2 |d| <= 14, by a 16-bit reciprocal: floor(32768 / |d|) + 1
    // (the truncated product with v_rcp is that floor: exact for powers of two, far from an integer otherwise)
    const unsigned y = 2u * static_cast<unsigned>(ad);
    const unsigned x = static_cast<unsigned>(r.n * (2 * t - 1)) + y - 1u;
    // (an axis the ray does not move along never crosses: its reciprocal is taken of 1, not of 0 -- an infinity
    // converted to unsigned and then used as a shift count is undefined, even though the result is discarded)
    const unsigned m = static_cast<unsigned>(32768.0f * __builtin_amdgcn_rcpf(static_cast<float>(ad ? ad : 1))) + 1u;
    const int pos = static_cast<int>((x * m) >> 16) & 15;
    cross[k] = (ad >= t) ? pos : 8;
    chg |= (ad >= t) ? (1u << pos) : 0u;
  }
  // the (up to three) later run starts, in position order
  const unsigned m1 = chg & ~1u, m2 = m1 & (m1 - 1u), m3 = m2 & (m2 - 1u);
  const int q1 = m1 ? __builtin_ctz(m1) : 8, q2 = m2 ? __builtin_ctz(m2) : 8, q3 = m3 ? __builtin_ctz(m3) : 8;
  const int nr = 1 + (m1 ? 1 : 0) + (m2 ? 1 : 0) + (m3 ? 1 : 0);
  const int last = r.n + 1;
  run_begin[0] = 0;  run_len[0] = min(q1, last);
  run_begin[1] = q1; run_len[1] = min(q2, last) - q1;
  run_begin[2] = q2; run_len[2] = min(q3, last) - q2;
  run_begin[3] = q3; run_len[3] = last - q3;
  unsigned long long key = block_key(r.bx, r.by, r.bz);
  run_key[0] = key;
  const int q[3] = {q1, q2, q3};
#pragma unroll
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // (a block coordinate in range stays inside its 11 bits)
      const long long step = static_cast<long long>(d[k] > 0 ? 1 : -1) << (11 * k);
      key += (cross[k] == q[j]) ? static_cast<unsigned long long>(step) : 0ull;
    }
    run_key[j + 1] = key;
  }
  return nr;
}

// Per-thread result of k_bin_count, consumed by k_bin_scatter: for each of <= 4 runs which block it falls into and
// where its records start inside the block's bin. Round 4: 16 bytes instead of 32. The workgroup writes ONE table
// entry per distinct block it touches -- {slot, first record of the workgroup's share in the bin} -- and a run names
// the entry (10 bits) and its offset inside the share (13 bits: a workgroup has at most 1024 runs of <= 8 records)
// next to its begin (3 bits) and length (4 bits, 0 = no run). The run info was 21 of the insert family's 55 MB of
// memory traffic per scan; the scatter pass now resolves ~40 table entries per workgroup (one bin-offset load each)
// instead of four bin-offset gathers per return.
struct RunInfo {
  uint32_t run[kMaxRuns];  // entry | offset << 10 | begin << 23 | len << 26
};
constexpr unsigned kWgTableEntries = 1024u;  // = the entries of the count pass's LDS table
struct WgTable {  // one per (workgroup, level), behind the run info of the call (wg_table)
  uint32_t count, pad[3];
  uint2 entry[kWgTableEntries];  // {slot, base of the workgroup's share inside the bin}
};
static_assert(sizeof(WgTable) % sizeof(RunInfo) == 0, "tables live in the run-info buffer");
// Run-info buffer of a scan of n returns (nwg workgroups of 256) and `levels` levels, in RunInfo units:
// [levels x n run infos][levels x nwg tables]
__host__ __device__ inline size_t run_info_units(size_t n, size_t nwg, size_t levels) {
  return levels * (n + nwg * (sizeof(WgTable) / sizeof(RunInfo)));
}
__device__ inline WgTable* wg_table(const RunInfo* runs, unsigned n, unsigned nwg, int levels, int level, unsigned wg) {
  WgTable* t = reinterpret_cast<WgTable*>(const_cast<RunInfo*>(runs) + static_cast<size_t>(levels) * n);
  return t + static_cast<size_t>(level) * nwg + wg;
}

// Bodies of the four kernels of the binned path, shared by the single-pyramid launches (pyramid in
// the kernel arguments) and the batched launches (a table of jobs in device memory, one job = one
// pyramid with its own scan: hg_register_scan_batch). `bx` of `nbx` = workgroup index inside the job.
#ifdef HG_COUNT_STAMPS
// diagnostics: wall time (s_memrealtime, 10 ns) the wavefronts of k_bin_count* spend per phase, summed over all
// wavefronts: [0] ray set-up (point load), [1] direct-slot key loads, [2] hash path (probe / insert / publish),
// [3] LDS aggregation + barrier, [4] bin reservation (returning atomics), [5] tables, run info, touched list;
// [6] wavefronts, [7] wavefronts that entered the hash path, [8] runs resolved by the hash path
__device__ unsigned long long g_count_stamps[16];
#define COUNT_STAMP(k) do { const long long now_ = __builtin_amdgcn_s_memrealtime(); \
    if ((threadIdx.x & 63u) == 0) atomicAdd(&g_count_stamps[k], static_cast<unsigned long long>(now_ - cs_t)); cs_t = now_; } while (0)
#else
#define COUNT_STAMP(k) do {} while (0)
#endif
__device__ __forceinline__ void bin_count_body(const PyramidIns& P, const LevelIns& L, int level, unsigned bx, unsigned nbx,
                                               const ScanTable* scans, uint32_t n_scans, const float* xyz,
                                               unsigned n, RunInfo* runs, unsigned* wg_hits) {
#ifdef HG_COUNT_STAMPS
  long long cs_t = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63u) == 0) atomicAdd(&g_count_stamps[6], 1ull);
#endif
  const unsigned i = xcd_chunk(bx, nbx) * 256u + threadIdx.x;  // see hg_device.h
  const int lane = threadIdx.x & (kWave - 1);
  WgTable* const tab = wg_table(runs, n, nbx, P.levels, level, bx);
  __shared__ unsigned s_hits, s_first, s_first_base, s_ntab;
  if (threadIdx.x == 0) { s_hits = 0; s_first = 0; s_ntab = 0; }
  __syncthreads();
  bool hit = false;
  unsigned long long run_key[kMaxRuns];
  int run_begin[kMaxRuns], run_len[kMaxRuns];
  int nr = 0;
  if (i < n) {
    // a chunk of several scans: returns are concatenated in scan order, so seq = i * 8 + sample is the
    // reference's update order across the whole chunk
    const ScanTable sc = scans ? scan_lookup(scans, n_scans, i, xcd_chunk(bx, nbx) * 256u) : scan_of(P);
    const Ray r = ray_setup(L.g, L.p, sc, xyz, i, L.gate);
    hit = r.valid && r.n + 1 <= kSlots;
    if (r.valid && !hit) atomicOr(&L.g.counters[1], kFlagStride);
    if (hit) {
      bool range_err = false;
      nr = ray_block_runs(r, run_key, run_begin, run_len, &range_err);
      if (range_err) atomicOr(&L.g.counters[1], kFlagRange);
    }
  }
  // block slots: the block's DIRECT slot first (round 4) -- block_keys[direct_slot(key)] == key + 1 says the
  // block exists and sits there: one load, no probe chain, and true for every block of a map inside its
  // window (what pyramid_tsd_direct does for the matcher). Round 3 probed the hash table first, where a
  // quarter of the keys need a second probe, so nearly every wavefront went through the probe / insert loop
  // (k_bin_count_jobs on maps in HBM: 85 % of its wave cycles waiting). All loads of the wave in flight
  // together; the hash path only for blocks that are new or live in the overflow area. (A block's key word is
  // set before its hash entry is published: seeing it is enough, the pool slot is pre-zeroed.)
  COUNT_STAMP(0);
  unsigned long long entry[kMaxRuns], tent[kMaxRuns];
  uint32_t slot[kMaxRuns], dslot[kMaxRuns];
  // A grid with blocks in the overflow area (a map wider than its direct window, or several maps folded into one
  // pool) finds those through the hash table. Round 5: their FIRST probe is issued together with the direct-slot
  // loads of all runs -- one round trip for a block wherever it lives, where the hash path used to start (one run
  // after the other, each probe awaited) only after the direct check had come back: on a stream over 400 room
  // copies half of the wavefronts' time (in-kernel stamps). Grids without overflow blocks skip it.
  const bool overflowed = __hip_atomic_load(&L.g.counters[8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
#pragma unroll
  for (int k = 0; k < kMaxRuns; ++k) {
    dslot[k] = (k < nr) ? direct_slot(L.g, run_key[k]) : 0u;
    entry[k] = (k < nr) ? __hip_atomic_load(&L.g.block_keys[dslot[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
  }
  if (overflowed) {
#pragma unroll
    for (int k = 0; k < kMaxRuns; ++k)
      tent[k] = (k < nr) ? __hip_atomic_load(&L.g.table[hash_key(run_key[k]) & L.g.table_mask], __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT)
                         : 0ull;
  } else {
#pragma unroll
    for (int k = 0; k < kMaxRuns; ++k) tent[k] = 0ull;
  }
#ifdef HG_COUNT_STAMPS
  {
    bool miss = false;
    unsigned nmiss = 0;
    for (int k = 0; k < kMaxRuns; ++k)
      if (k < nr && entry[k] != run_key[k] + 1ull &&
          !((tent[k] >> 24) == run_key[k] + 1ull && static_cast<uint32_t>(tent[k] & 0xFFFFFFu) != kSlotPending)) { miss = true; ++nmiss; }
    COUNT_STAMP(1);
    const unsigned long long mm = __ballot(miss);
    if (mm && (threadIdx.x & 63u) == 0) atomicAdd(&g_count_stamps[7], 1ull);
    {  // (one atomic per wavefront: per-lane atomics on one word would be the measurement)
      unsigned tot = nmiss;
      for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
      if (tot && (threadIdx.x & 63u) == 0) atomicAdd(&g_count_stamps[8], static_cast<unsigned long long>(tot));
    }
  }
#endif
#pragma unroll
  for (int k = 0; k < kMaxRuns; ++k) {
    slot[k] = 0xFFFFFFFFu;
    if (k < nr) {
      const unsigned long long tag = run_key[k] + 1ull;
      if (entry[k] == tag)
        slot[k] = dslot[k];
      else if ((tent[k] >> 24) == tag && static_cast<uint32_t>(tent[k] & 0xFFFFFFu) != kSlotPending)
        slot[k] = static_cast<uint32_t>(tent[k] & 0xFFFFFFu);
      else
        slot[k] = insert_block_shared(L.g, run_key[k]);
    }
  }
  COUNT_STAMP(2);
  // Record ranges inside the bins: the runs of the WORKGROUP are added up per block in an LDS hash
  // table (one LDS atomic per run: it returns the run's offset inside the workgroup's share, in any
  // order -- the order of the records inside a bin is free, the apply pass orders them by seq), then ONE
  // device-scope atomic per (workgroup, block) reserves the share in the bin. Grouping the lanes of every
  // wavefront by block with ballots and issuing one atomic per (wavefront, block, run index) took a
  // third of this kernel and four times as many returning atomics.
  constexpr unsigned kTable = 1024u;  // entries; a workgroup has at most 1024 runs, a few dozen distinct blocks
  constexpr unsigned kNone = 0xFFFFFFFFu;
  __shared__ uint32_t t_key[kTable];  // slot + 1, 0 = free
  __shared__ uint32_t t_cnt[kTable];  // records of the workgroup in the block; after the scan: their base in the bin
  for (unsigned e = threadIdx.x; e < kTable; e += 256u) { t_key[e] = 0u; t_cnt[e] = 0u; }
  __syncthreads();
  bool want[kMaxRuns];
  unsigned ent[kMaxRuns], prefix[kMaxRuns];
  unsigned first_pos[kMaxRuns + kTable / 256u], first_slot[kMaxRuns + kTable / 256u];
#pragma unroll
  for (int j = 0; j < kMaxRuns + static_cast<int>(kTable / 256u); ++j) first_pos[j] = kNone;
#pragma unroll
  for (int k = 0; k < kMaxRuns; ++k) {
    want[k] = k < nr && slot[k] < L.g.pool_blocks;
    ent[k] = kNone;
    prefix[k] = 0;
    if (want[k]) {
      unsigned h = (slot[k] * 2654435761u) >> 22;  // 10 bits
      // (at most 1024 runs, hence at most 1024 distinct blocks, in 1024 entries: the probe always ends)
      for (unsigned tries = 0; tries < kTable; ++tries) {
        const uint32_t old = atomicCAS(&t_key[h], 0u, slot[k] + 1u);
        if (old == 0u || old == slot[k] + 1u) { ent[k] = h; break; }
        h = (h + 1u) & (kTable - 1u);
      }
      prefix[k] = atomicAdd(&t_cnt[ent[k]], static_cast<unsigned>(run_len[k]));
    }
  }
  __syncthreads();
  COUNT_STAMP(3);
  // one returning device-scope atomic per occupied entry, all of a thread's in flight together
  uint32_t e_key[kTable / 256u], e_base[kTable / 256u];
#pragma unroll
  for (unsigned j = 0; j < kTable / 256u; ++j) {
    const unsigned e = threadIdx.x + 256u * j;
    e_key[j] = t_key[e];
    e_base[j] = e_key[j] ? atomicAdd(&L.g.bin_count[e_key[j] - 1u], t_cnt[e]) : 0u;
  }
#pragma unroll
  for (unsigned j = 0; j < kTable / 256u; ++j) {
    const unsigned e = threadIdx.x + 256u * j;
    if (e_key[j]) {
      // the workgroup's table entry of the block: {slot, base of its share}; the runs name it by its position
      const unsigned cid = atomicAdd(&s_ntab, 1u);
      tab->entry[cid] = make_uint2(e_key[j] - 1u, e_base[j]);
      t_cnt[e] = cid;
      // blocks that receive their first records of this call are enlisted in `touched`. The list's
      // cursor is ONE device-wide word: the workgroup reserves its entries with a single atomic
      if (e_base[j] == 0u) { first_pos[kMaxRuns + j] = atomicAdd(&s_first, 1u); first_slot[kMaxRuns + j] = e_key[j] - 1u; }
    }
  }
  COUNT_STAMP(4);
  const unsigned long long m = __ballot(hit);
  if (lane == 0 && m) atomicAdd(&s_hits, static_cast<unsigned>(__popcll(m)));
  __syncthreads();
  if (threadIdx.x == 0) {
    wg_hits[level * nbx + bx] = s_hits;
    s_first_base = s_first ? atomicAdd(&L.g.call[0], s_first) : 0u;
    tab->count = s_ntab;
  }
  RunInfo info;
#pragma unroll
  for (int k = 0; k < kMaxRuns; ++k)
    info.run[k] = want[k] ? (t_cnt[ent[k]] | (prefix[k] << 10) | (static_cast<uint32_t>(run_begin[k]) << 23) |
                             (static_cast<uint32_t>(run_len[k]) << 26))
                          : 0u;
  if (i < n) runs[static_cast<size_t>(level) * n + i] = info;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kMaxRuns + static_cast<int>(kTable / 256u); ++j)
    if (first_pos[j] != kNone) L.g.touched[s_first_base + first_pos[j]] = first_slot[j];
  COUNT_STAMP(5);
}


// One pyramid with its own scan inside a batched launch.
struct InsertJob {
  PyramidIns P;
  const float* xyz;
  RunInfo* runs;
  uint32_t* rec_keys;
  uint32_t* rec_vals;
  unsigned* wg_hits;
  unsigned n;                  // returns of the job's scan
  unsigned nwg;                // workgroups of 256 returns
  unsigned records_per_level;  // n * kSlots
  unsigned pad;
};

__global__ __launch_bounds__(256) void k_bin_count(PyramidIns P, const ScanTable* scans, uint32_t n_scans,
                                                   const float* xyz, unsigned n, RunInfo* runs, unsigned* wg_hits) {
  bin_count_body(P, P.lv[blockIdx.y], blockIdx.y, blockIdx.x, gridDim.x, scans, n_scans, xyz, n, runs, wg_hits);
}
// grid (max nwg, jobs * levels)
__global__ __launch_bounds__(256) void k_bin_count_jobs(const InsertJob* __restrict__ jobs, int levels) {
  const InsertJob& J = jobs[blockIdx.y / levels];
  if (blockIdx.x >= J.nwg) return;
  const LevelIns L = level_as_device(J.P.lv[blockIdx.y % levels]);  // a copy, see k_bin_apply_small_jobs
  bin_count_body(J.P, L, blockIdx.y % levels, blockIdx.x, J.nwg, nullptr, 1u, as_device(J.xyz), J.n, as_device(J.runs), as_device(J.wg_hits));
}

// ==========================================================================================
// Tolerance path (HG_INSERT_FAST, unit update weight): the updates a voxel receives in one chunk of a call are SUMMED
// (order-free integer sums: count << 44 | sum of (tsd + tau) in units of 2 tau / (2^23 - 1)) and applied once in
// closed form, instead of the reference's chain of re-quantised updates. Results differ from the exact mode by the
// re-quantisation noise the reference accumulates per update (tests/test_gpu_insert_fast.py states the tolerance);
// they do not depend on arrival order, so the mode is deterministic. Kernels: "Tolerance path on the bins" below
// (round 6; rounds 2-5 formed the sums with device-scope atomics into an 8-byte accumulator per voxel of the pool:
// k_fast_accumulate / k_fast_apply, git history).
// ==========================================================================================
constexpr unsigned kFastCountShift = 44;
constexpr unsigned kFastUnits = (1u << 23) - 1u;

// Product over the m updates of w_{k-1} / (w_{k-1} + 1): the share of the voxel's previous tsd that survives
// (UpdateCell :725-737). The reference re-quantises the weight after every update, so its weight CODE grows by
// round(weight_resolution) per update (33 for max weight 1000, i.e. 1.0071 instead of 1) until the clamp pins it at
// 32767; with that sequence the product is a ratio of Gamma functions. `code` is the weight code before the call
// (0 = unknown = weight 0). Rounds 2-5 evaluated it with lgamma / exp / pow in fp64 (240 VGPRs: most of the apply kernel).
// Round 6: with alpha = (c0 - 1) / step and
// beta = alpha + 1 + delta, delta = 1 / (kw step) - 1 (-0.0071 for maximum weight 1000: the stored weight of an
// update is 1.0071, not 1), the growth phase is
//   prod_{j<n} (alpha + j) / (beta + j) = alpha / (alpha + n) * [G(a + delta) / G(a)] / [G(b + delta) / G(b)],
//   a = alpha + 1, b = alpha + n + 1 (G = Gamma), and log(G(x + delta) / G(x)) = delta psi(x) + delta^2 / 2 psi'(x) + O(delta^3)
// with the asymptotic digamma / trigamma series (x >= 1; error < 3e-3 in psi at x = 1, times delta: < 3e-5 in A, a
// tsd code is 3e-5 of the range, and far less for the x >= 2 every weight after the first update has).
// The saturated phase is exp((m - n) log(wmax / (wmax + 1))). fp32 throughout: the relative error of A stays
// below 1e-6, the updates' mean and the blend are formed in fp64 by the caller as before.
__device__ inline float fast_survival_f(const GridView& g, uint32_t code, unsigned m, float log_q_sat, uint32_t* code_out) {
#pragma clang fp contract(fast)
  const int step = static_cast<int>(roundf(g.weight_resolution));
  const int c0 = static_cast<int>(code & 0x7FFFu) == 0 ? 1 : static_cast<int>(code & 0x7FFFu);
  const unsigned grow_all = c0 >= 32767 ? 0u : static_cast<unsigned>((32767 - c0 + step - 1) / step);
  const unsigned n = min(m, grow_all);
  const unsigned long long c_end = static_cast<unsigned long long>(c0) + static_cast<unsigned long long>(step) * m;
  *code_out = static_cast<uint32_t>(c_end < 32767ull ? c_end : 32767ull);
  float A = 1.0f;
  if (n > 0u) {
    if (c0 == 1) {
      A = 0.0f;  // weight 0: the first update replaces the value
    } else {
      const float fstep = static_cast<float>(step);
      const float alpha = static_cast<float>(c0 - 1) / fstep;
      const float delta = 1.0f / (g.weight_scale * fstep) - 1.0f;
      const float fn = static_cast<float>(n);
      const float a = alpha + 1.0f, b = alpha + fn + 1.0f;
      const float ra = 1.0f / a, rb = 1.0f / b;
      const float ra2 = ra * ra, rb2 = rb * rb;
      // psi(a) - psi(b), psi'(a) - psi'(b)
      const float dpsi = __logf(a * rb) - 0.5f * (ra - rb) - (1.0f / 12.0f) * (ra2 - rb2) + (1.0f / 120.0f) * (ra2 * ra2 - rb2 * rb2);
      const float dtri = (ra - rb) + 0.5f * (ra2 - rb2) + (1.0f / 6.0f) * (ra2 * ra - rb2 * rb);
      const float x = delta * dpsi + 0.5f * delta * delta * dtri;
      A = alpha / (alpha + fn) * __expf(x);
    }
  }
  if (m > n) A *= __expf(static_cast<float>(m - n) * log_q_sat);
  return A;
}

// Same quantities by walking the weight sequence update by update: for weight resolutions whose
// fraction is too close to one half for the closed form above (not the case for the defaults).
__device__ inline double fast_survival_walk(const GridView& g, float maxw, uint32_t code, double m, uint32_t* code_out) {
  float w = value_to_weight(g, code);
  double A = 1.0;
  uint32_t c = code & 0x7FFFu;
  for (double k = 0; k < m; k += 1.0) {
    const float W = w + 1.0f;
    A *= static_cast<double>(w) / static_cast<double>(W);
    c = weight_to_value(g, fminf(W, maxw));
    w = value_to_weight(g, c);
  }
  *code_out = c;
  return A;
}

// Lanes of the wavefront (among `valid` ones) that hold the same 9-bit value as this lane.
__device__ inline unsigned long long match_voxel(unsigned v, bool valid) {
  unsigned long long m = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 9; ++b) {
    const bool bit = (v >> b) & 1u;
    const unsigned long long bb = __ballot(bit);
    m &= bit ? bb : ~bb;
  }
  return valid ? m : 0ull;
}

// Exclusive prefix sum over the workgroup (blockDim.x a multiple of 64, <= 1024): wave scan by
// shuffles, wave totals through LDS. Returns the exclusive prefix; *total = sum over the block.
__device__ inline unsigned block_exclusive_scan(unsigned v, unsigned* s_wave /*[16]*/, unsigned* total) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int nwaves = blockDim.x >> 6;
  unsigned incl = v;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const unsigned t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  __syncthreads();  // s_wave may still be read from a previous call
  if (lane == kWave - 1) s_wave[wave] = incl;
  __syncthreads();
  unsigned before = 0, all = 0;
  for (int w = 0; w < nwaves; ++w) {
    const unsigned t = s_wave[w];
    if (w < wave) before += t;
    all += t;
  }
  *total = all;
  return before + incl - v;
}

// grid (levels), 1024 threads: bin offsets by an exclusive scan over the touched list, and the
// apply work list: a bin larger than one LDS pass is split into voxel slices handled by
// different workgroups (voxels are independent of each other); slice k takes the voxels
// v whose mixed low bits (slice_of in k_bin_apply) equal k.
// Reserves `mine` consecutive work items for every lane with ONE LDS atomic per wavefront (a prefix
// sum over the lanes): thousands of same-address atomics of a level's touched blocks serialised.
__device__ inline unsigned reserve_items(unsigned* counter, unsigned mine) {
  const int lane = threadIdx.x & (kWave - 1);
  unsigned incl = mine;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const unsigned t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  const unsigned total = __shfl(incl, kWave - 1);
  unsigned base = 0;
  if (lane == kWave - 1 && total) base = atomicAdd(counter, total);
  return __shfl(base, kWave - 1) + incl - mine;
}

__device__ __forceinline__ void bin_offsets_body(const PyramidIns& P, const LevelIns& L, int level, unsigned records_per_level) {
  // A large bin is cut into voxel slices (voxels are independent). Single registration chain
  // (slice_records = 0, latency): bins below 4096 records are cut into slices of up to 2048 records
  // (one LDS pass each), larger ones into slices of 512 records -- the long per-voxel chains of a heavy
  // bin then sit on as many workgroups as possible, while the many mid-size bins are not read by four
  // to eight workgroups each (measured against 512 throughout: +3 % per registration step, +13 % on
  // the scan stream; 1024 / 768 / 384 / 256-record slices and thresholds 3072 - 8192 were worse). Batched
  // registration (slice_records = 2048, throughput): one LDS pass per slice, every bin read four times
  // less often.
  const unsigned slice_records = P.slice_records > 0 ? static_cast<unsigned>(P.slice_records) : 512u;
  // one wavefront per small bin: inside k_bin_apply (eight per workgroup, LDS for 256 records each) or,
  // batched registration, in k_bin_apply_small_jobs (512 records)
  const unsigned small_cap = P.slice_records <= 0 ? kSmallBinInKernel : kSmallBin;
  // slice_records < 0: sizes by bin as for 0, large bins in slices of -slice_records records (scan stream)
  const unsigned slice_above = P.slice_records < 0 ? static_cast<unsigned>(-P.slice_records) : HG_SLICE_ABOVE;
  // bits of the call's seq values (seq < records_per_level): the apply pass cuts a voxel's records into seq buckets
  const unsigned seq_bits = 32u - static_cast<unsigned>(__builtin_clz((records_per_level > 2u ? records_per_level : 2u) - 1u));
  __shared__ unsigned s_scan[16];
  __shared__ unsigned s_base, s_work, s_large;
  const unsigned nt = L.g.call[0];
  if (threadIdx.x == 0) { s_base = 0; s_work = 0; s_large = 0; }
  __syncthreads();
  // two rounds over the touched list: round 0 assigns offsets and emits the work items of large
  // bins (their long per-voxel chains are the critical path, so they are scheduled first),
  // round 1 emits the rest. Up to 4096 touched blocks are held in registers (two dependent memory
  // round trips for the whole list instead of two per 1024-block chunk and round).
  constexpr int kRegChunks = 4;
  const bool in_regs = nt <= 1024u * kRegChunks;
  unsigned r_slot[kRegChunks], r_cnt[kRegChunks], r_off[kRegChunks];
  if (in_regs) {
#pragma unroll
    for (int c = 0; c < kRegChunks; ++c) {
      const unsigned i = c * 1024u + threadIdx.x;
      r_slot[c] = i < nt ? L.g.touched[i] : 0u;
    }
#pragma unroll
    for (int c = 0; c < kRegChunks; ++c) {
      const unsigned i = c * 1024u + threadIdx.x;
      r_cnt[c] = i < nt ? L.g.bin_count[r_slot[c]] : 0u;
    }
  }
  if (in_regs) {
    // (Round 6) the whole list in ONE pass: records, large-bin items and small-bin items are three sums per thread
    // (its up to four entries), scanned together behind one pair of barriers; offsets and item positions follow
    // from the three prefixes. The two rounds over four chunks it replaces were ~30 barriers, two per block scan and
    // two more per chunk and round, for a list of 200 - 1800 blocks: 10.3 us per scan on the single chain, between
    // two kernels that wait for it.
    unsigned sl[kRegChunks];
    unsigned my_recs = 0, my_big = 0, my_small = 0;
#pragma unroll
    for (int c = 0; c < kRegChunks; ++c) {
      const unsigned i = c * 1024u + threadIdx.x;
      const bool valid = i < nt;
      const unsigned cnt = r_cnt[c];
      unsigned slices = valid ? 1u : 0u;
      const unsigned per_slice = P.slice_records > 0 ? slice_records : (cnt < HG_SLICE_THRESH ? HG_SLICE_BELOW : slice_above);
      while (valid && slices < 128 && cnt > slices * per_slice) slices <<= 1;  // 1 slice while cnt <= per_slice
      sl[c] = slices;
      my_recs += cnt;  // (0 beyond the list)
      if (cnt > small_cap) my_big += slices; else my_small += slices;
    }
    __shared__ unsigned s_scan3[16][3];
    unsigned e_recs, e_big, e_small, t_recs = 0, t_big = 0, t_small = 0;
    {
      const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
      unsigned a = my_recs, b = my_big, d = my_small;
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        const unsigned ta = __shfl_up(a, off), tb = __shfl_up(b, off), td = __shfl_up(d, off);
        if (lane >= off) { a += ta; b += tb; d += td; }
      }
      if (lane == kWave - 1) { s_scan3[wave][0] = a; s_scan3[wave][1] = b; s_scan3[wave][2] = d; }
      __syncthreads();
      unsigned ba = 0, bb = 0, bd = 0;
      for (int w = 0; w < 16; ++w) {
        const unsigned xa = s_scan3[w][0], xb = s_scan3[w][1], xd = s_scan3[w][2];
        if (w < wave) { ba += xa; bb += xb; bd += xd; }
        t_recs += xa; t_big += xb; t_small += xd;
      }
      e_recs = ba + a - my_recs;
      e_big = bb + b - my_big;
      e_small = bd + d - my_small;
    }
#pragma unroll
    for (int c = 0; c < kRegChunks; ++c) {
      const unsigned i = c * 1024u + threadIdx.x;
      if (i < nt) {
        const unsigned slot = r_slot[c], cnt = r_cnt[c], slices = sl[c];
        const unsigned bin_off = static_cast<unsigned>(level) * records_per_level + e_recs;
        L.g.bin_offset[slot] = bin_off;
        L.g.bin_count[slot] = 0;  // ready for the next call (items carry n)
        const bool big = cnt > small_cap;
        const unsigned w0 = big ? e_big : t_big + e_small;
        const unsigned step = 512u / slices;
        for (unsigned k = 0; k < slices; ++k) {
          if (w0 + k < L.g.work_capacity)
            L.g.work[w0 + k] = make_uint4(slot, (k * step) | (((k + 1) * step) << 10) | (seq_bits << 20), cnt, bin_off);
          else
            atomicOr(&L.g.counters[1], kFlagWorkOverflow);  // cannot happen: the host sizes the list from the records
        }
        e_recs += cnt;
        if (big) e_big += slices; else e_small += slices;
      }
    }
    if (threadIdx.x == 0) { s_base = t_recs; s_large = t_big; s_work = t_big + t_small; }
    __syncthreads();
  }
  for (int round = 0; !in_regs && round < 2; ++round) {
#pragma unroll
    for (int c = 0; c < kRegChunks; ++c) {
      const unsigned c0 = c * 1024u;
      if (!in_regs || c0 >= nt) break;
      const unsigned i = c0 + threadIdx.x;
      const unsigned slot = r_slot[c], cnt = r_cnt[c];
      unsigned chunk_total = 0;
      unsigned bin_off = 0;  // carried by the work items: the apply pass does not read bin_offset[]
      if (round == 0) {
        const unsigned excl = block_exclusive_scan(cnt, s_scan, &chunk_total);
        bin_off = static_cast<unsigned>(level) * records_per_level + s_base + excl;
        if (i < nt) L.g.bin_offset[slot] = bin_off;
        r_off[c] = bin_off;
      } else {
        bin_off = r_off[c];
      }
      // smaller bins are whole-bin items for one wavefront each. (The heaviest bins in a round of their own, ahead
      // of the other large ones, start their slices 13 us earlier and leave the launch as long as it was.)
      const int tier = cnt > small_cap ? 0 : 1;
      const bool emit = i < nt && round == tier;
      unsigned slices = emit ? 1u : 0u;
      const unsigned per_slice = P.slice_records > 0 ? slice_records : (cnt < HG_SLICE_THRESH ? HG_SLICE_BELOW : slice_above);
      while (emit && slices < 128 && cnt > slices * per_slice) slices <<= 1;  // 1 slice while cnt <= per_slice
      const unsigned w0 = reserve_items(&s_work, slices);
      if (emit) {
        const unsigned step = 512u / slices;
        for (unsigned k = 0; k < slices; ++k) {
          if (w0 + k < L.g.work_capacity)
            L.g.work[w0 + k] = make_uint4(slot, (k * step) | (((k + 1) * step) << 10) | (seq_bits << 20), cnt, bin_off);
          else
            atomicOr(&L.g.counters[1], kFlagWorkOverflow);  // cannot happen: the host sizes the list from the records
        }
      }
      if (round == 0 && i < nt) L.g.bin_count[slot] = 0;  // ready for the next call (items carry n)
      if (round == 0) {
        __syncthreads();
        if (threadIdx.x == 0) s_base += chunk_total;
        __syncthreads();
      }
    }
    for (unsigned c0 = 0; !in_regs && c0 < nt; c0 += 1024) {
      const unsigned i = c0 + threadIdx.x;
      const unsigned slot = i < nt ? L.g.touched[i] : 0u;
      unsigned cnt = 0;
      cnt = i < nt ? L.g.bin_count[slot] : 0u;
      unsigned chunk_total = 0;
      unsigned bin_off = 0;  // carried by the work items: the apply pass does not read bin_offset[]
      if (round == 0) {
        const unsigned excl = block_exclusive_scan(cnt, s_scan, &chunk_total);
        bin_off = static_cast<unsigned>(level) * records_per_level + s_base + excl;
        if (i < nt) L.g.bin_offset[slot] = bin_off;
      } else if (i < nt) {
        bin_off = L.g.bin_offset[slot];  // written by this thread in round 0
      }
      // smaller bins are whole-bin items for one wavefront each. (The heaviest bins in a round of their own, ahead
      // of the other large ones, start their slices 13 us earlier and leave the launch as long as it was.)
      const int tier = cnt > small_cap ? 0 : 1;
      const bool emit = i < nt && round == tier;
      unsigned slices = emit ? 1u : 0u;
      const unsigned per_slice = P.slice_records > 0 ? slice_records : (cnt < HG_SLICE_THRESH ? HG_SLICE_BELOW : slice_above);
      while (emit && slices < 128 && cnt > slices * per_slice) slices <<= 1;  // 1 slice while cnt <= per_slice
      const unsigned w0 = reserve_items(&s_work, slices);
      if (emit) {
        const unsigned step = 512u / slices;
        for (unsigned k = 0; k < slices; ++k) {
          if (w0 + k < L.g.work_capacity)
            L.g.work[w0 + k] = make_uint4(slot, (k * step) | (((k + 1) * step) << 10) | (seq_bits << 20), cnt, bin_off);
          else
            atomicOr(&L.g.counters[1], kFlagWorkOverflow);  // cannot happen: the host sizes the list from the records
        }
      }
      if (round == 1 && i < nt) L.g.bin_count[slot] = 0;  // ready for the next call (items carry n)
      __syncthreads();
      if (round == 0 && threadIdx.x == 0) s_base += chunk_total;
      __syncthreads();
    }
    __syncthreads();
    if (round == 0 && threadIdx.x == 0) s_large = s_work;  // items [0, s_large): slices of large bins
    __syncthreads();  // snapshot taken before any wavefront reserves round-1 items
  }
  if (threadIdx.x == 0) {
    // items [call[2], call[1]) are whole bins for k_bin_apply_small (batched inserts); one
    // registration chain keeps them in k_bin_apply: a second kernel behind it costs more than it saves
    L.g.call[2] = min((P.slice_records >= 2048 || P.slice_records <= 0) ? s_large : s_work, L.g.work_capacity);
    L.g.call[1] = min(s_work, L.g.work_capacity);  // consumed by k_bin_apply
    L.g.call[3] = 0;                               // values of deferred long chains (k_bin_apply)
    L.g.call[0] = 0;                               // next call collects from scratch
    unsigned long long* upd = reinterpret_cast<unsigned long long*>(&L.g.counters[4]);
    if (P.shared) atomicAdd(upd, static_cast<unsigned long long>(s_base));
    else *upd = s_base + (P.accumulate ? *upd : 0ull);  // U of this call
    publish_flags(P, level);
  }
}

__global__ __launch_bounds__(1024) void k_bin_offsets(PyramidIns P, unsigned records_per_level) {
  bin_offsets_body(P, P.lv[blockIdx.x], blockIdx.x, records_per_level);
}
// grid (jobs * levels)
__global__ __launch_bounds__(1024) void k_bin_offsets_jobs(const InsertJob* __restrict__ jobs, int levels) {
  const InsertJob& J = jobs[blockIdx.x / levels];
  const LevelIns L = level_as_device(J.P.lv[blockIdx.x % levels]);
  bin_offsets_body(J.P, L, blockIdx.x % levels, J.records_per_level);
}

// FAST (tolerance mode on the bins, k_fast_bin_apply): one word per record, voxel | units << 9 with the sample's
// tsd + tau in units of 2 tau / (2^23 - 1) -- the order of a voxel's updates does not matter there, so no seq.
template <bool FAST = false>
__device__ __forceinline__ void bin_scatter_body(const PyramidIns& P, const LevelIns& L, int level, unsigned bx, unsigned nbx,
                                                 const ScanTable* scans, uint32_t n_scans, const float* xyz,
                                                 unsigned n, const RunInfo* runs, uint32_t* rec_keys,
                                                 uint32_t* rec_vals) {
  const unsigned i = xcd_chunk(bx, nbx) * 256u + threadIdx.x;
  // the workgroup's block table: entry -> first record of the workgroup's share (bin offset + base inside the bin)
  __shared__ uint32_t s_base[kWgTableEntries];
  RunInfo info;
#pragma unroll
  for (int k = 0; k < kMaxRuns; ++k) info.run[k] = 0u;
  if (i < n) info = runs[static_cast<size_t>(level) * n + i];  // in flight while the table is resolved
  {
    const WgTable* tab = wg_table(runs, n, nbx, P.levels, level, bx);
    const unsigned nt = tab->count;
    for (unsigned e = threadIdx.x; e < nt; e += 256u) {
      const uint2 v = tab->entry[e];
      s_base[e] = L.g.bin_offset[v.x] + v.y;
    }
  }
  __syncthreads();
  if ((info.run[0] | info.run[1] | info.run[2] | info.run[3]) == 0u) return;
  const ScanTable sc = scans ? scan_lookup(scans, n_scans, i, xcd_chunk(bx, nbx) * 256u) : scan_of(P);
  const Ray r = ray_setup(L.g, L.p, sc, xyz, i, L.gate);
  // One pass over the sample positions for all lanes (the per-run loops made a wavefront walk
  // every run as long as its longest lane: ~10-12 sample evaluations where 8 suffice). The runs of a ray
  // are consecutive stretches of in-range samples inside one block, in position order, so a lane only
  // tracks which run it is in and how many of that run's samples it has written.
  // Run k holds the samples [begin_k, begin_k + len_k) and writes them to consecutive records from dst_k on:
  // sample pos of run k goes to (dst_k - begin_k) + pos. Samples outside every run (beyond the ray's end,
  // outside the index range, in a block that could not be allocated) are not written.
  unsigned base[kMaxRuns], len[kMaxRuns];
  int beg[kMaxRuns];
#pragma unroll
  for (int k = 0; k < kMaxRuns; ++k) {
    len[k] = info.run[k] >> 26;
    beg[k] = static_cast<int>((info.run[k] >> 23) & 7u);
    base[k] = len[k] ? s_base[info.run[k] & 1023u] + ((info.run[k] >> 10) & 8191u) - static_cast<unsigned>(beg[k]) : 0u;
  }
  RayWalk walk;  // (every ray that left run info has n <= 7: the integer walk is exact)
  walk.begin(r);
#pragma unroll
  for (int pos = 0; pos < kSlots; ++pos) {
    const int cx = walk.c[0], cy = walk.c[1], cz = walk.c[2];
    walk.step();
    bool in = false;
    unsigned at = 0;
#pragma unroll
    for (int k = 0; k < kMaxRuns; ++k) {
      const bool ink = static_cast<unsigned>(pos - beg[k]) < len[k];
      in = in || ink;
      at = ink ? base[k] : at;
    }
    if (!in) continue;
    float tsd, w;
    ray_sample_cell<true, FAST>(L.g, L.p, r, cx, cy, cz, tsd, w);  // (unit weights: a condition of the binned paths)
    if (FAST) {
      const float tau = L.p.truncation_distance;
      const float to_units = static_cast<float>(kFastUnits) / (tau + tau);
      const unsigned units = min(static_cast<unsigned>(__float2int_rn((tsd + tau) * to_units)), kFastUnits);
      rec_keys[at + pos] = voxel_in_block(cx, cy, cz) | (units << 9);
    } else {
      rec_keys[at + pos] = (voxel_in_block(cx, cy, cz) << kSeqBits) | (i * kSlots + pos);
      rec_vals[at + pos] = __float_as_uint(tsd);
    }
  }
}

__global__ __launch_bounds__(256) void k_bin_scatter(PyramidIns P, const ScanTable* scans, uint32_t n_scans,
                                                     const float* xyz, unsigned n, const RunInfo* runs,
                                                     uint32_t* rec_keys, uint32_t* rec_vals) {
  bin_scatter_body(P, P.lv[blockIdx.y], blockIdx.y, blockIdx.x, gridDim.x, scans, n_scans, xyz, n, runs, rec_keys, rec_vals);
}
__global__ __launch_bounds__(256) void k_bin_scatter_jobs(const InsertJob* __restrict__ jobs, int levels) {
  const InsertJob& J = jobs[blockIdx.y / levels];
  if (blockIdx.x >= J.nwg) return;
  const LevelIns L = level_as_device(J.P.lv[blockIdx.y % levels]);
  bin_scatter_body(J.P, L, blockIdx.y % levels, blockIdx.x, J.nwg, nullptr, 1u, as_device(J.xyz), J.n, as_device(J.runs),
                   as_device(J.rec_keys), as_device(J.rec_vals));
}

// ==========================================================================================
// Tolerance path on the bins (round 6; replaces k_fast_accumulate's one device-scope 64-bit atomic per distinct
// voxel of a workgroup). Same semantics -- the updates a voxel receives in one chunk of a call are summed as
// integers and applied once in closed form -- but the sums are formed where they cost nothing:
//   k_bin_count          (the exact path's) block runs, block slots, bin reservations
//   k_fast_offsets       per level: exclusive scan of the touched bins; ONE apply item per bin, bins beyond
//                        kFastItemRecords records cut into parts by RECORD range (no order to keep)
//   k_fast_scatter       one 4-byte record per sample: voxel | units << 9
//   k_fast_bin_apply     per item: the records stream in coalesced, an LDS tile of 512 x {count << 44 | sum} takes
//                        them by LDS atomics, one closed-form UpdateCell per touched voxel, 2 KiB block read and
//                        written once. Parts of a cut bin leave their tiles in a scratch area; the part that
//                        arrives last (a ticket per bin) adds them up and applies.
// No accumulator array (8 B per voxel of the pool), no global atomics on voxels.
// ==========================================================================================
constexpr unsigned kFastItemRecords = 8192;  // records of one apply item (32 per thread)
constexpr unsigned kFastApplyThreads = 256;
constexpr unsigned kFastMaxParts = 256;      // parts of one bin (8 bits in the item)

// Tiles of the parts of cut bins: per level `tile_capacity` tiles of 512 x u64 behind each other.
struct FastScratch {
  unsigned long long* tiles;
  unsigned tile_capacity;  // per level
};

// grid (levels), 1024 threads. Work item {slot | part << 24, first record, records | (parts - 1) << 24, first tile
// of the bin}.
__global__ __launch_bounds__(1024) void k_fast_offsets(PyramidIns P, unsigned records_per_level, FastScratch fs) {
  const int level = blockIdx.x;
  const LevelIns& L = P.lv[level];
  __shared__ unsigned s_scan[16];
  __shared__ unsigned s_base, s_work, s_tiles;
  const unsigned nt = L.g.call[0];
  if (threadIdx.x == 0) { s_base = 0; s_work = 0; s_tiles = 0; }
  __syncthreads();
  for (unsigned c0 = 0; c0 < nt; c0 += 1024u) {
    const unsigned i = c0 + threadIdx.x;
    const unsigned slot = i < nt ? L.g.touched[i] : 0u;
    const unsigned cnt = i < nt ? L.g.bin_count[slot] : 0u;
    unsigned chunk_total = 0;
    const unsigned excl = block_exclusive_scan(cnt, s_scan, &chunk_total);
    const unsigned bin_off = static_cast<unsigned>(level) * records_per_level + s_base + excl;
    if (i < nt) {
      L.g.bin_offset[slot] = bin_off;
      L.g.bin_count[slot] = 0;  // the apply pass counts the arrivals of a cut bin's parts here and leaves 0 behind
    }
    unsigned parts = cnt ? 1u : 0u;
    unsigned per_part = cnt;
    if (cnt > kFastItemRecords) {
      parts = min((cnt + kFastItemRecords - 1u) / kFastItemRecords, kFastMaxParts);
      per_part = (cnt + parts - 1u) / parts;
    }
    const unsigned w0 = reserve_items(&s_work, parts);
    const unsigned t0 = reserve_items(&s_tiles, parts > 1u ? parts : 0u);
    bool fits = w0 + parts <= L.g.work_capacity && (parts <= 1u || t0 + parts <= fs.tile_capacity);
    if (parts && !fits) atomicOr(&L.g.counters[1], kFlagWorkOverflow);  // cannot happen: the host sizes both from the records
    for (unsigned k = 0; fits && k < parts; ++k) {
      const unsigned b = k * per_part, e = min(cnt, b + per_part);
      L.g.work[w0 + k] = make_uint4(slot | (k << 24), bin_off + b, (e - b) | ((parts - 1u) << 24),
                                    static_cast<unsigned>(level) * fs.tile_capacity + t0);
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base += chunk_total;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    L.g.call[1] = min(s_work, L.g.work_capacity);
    L.g.call[0] = 0;  // next call collects from scratch
    unsigned long long* upd = reinterpret_cast<unsigned long long*>(&L.g.counters[4]);
    if (P.shared) atomicAdd(upd, static_cast<unsigned long long>(s_base));
    else *upd = s_base + (P.accumulate ? *upd : 0ull);  // U of this call
    publish_flags(P, level);
  }
}

__global__ __launch_bounds__(256) void k_fast_scatter(PyramidIns P, const ScanTable* scans, uint32_t n_scans,
                                                      const float* xyz, unsigned n, const RunInfo* runs, uint32_t* recs) {
  bin_scatter_body<true>(P, P.lv[blockIdx.y], blockIdx.y, blockIdx.x, gridDim.x, scans, n_scans, xyz, n, runs, recs, nullptr);
}

// grid (G, levels), 256 threads, two voxels per thread.
__global__ __launch_bounds__(kFastApplyThreads) void k_fast_bin_apply(PyramidIns P, const uint32_t* __restrict__ recs,
                                                                     FastScratch fs) {
  const LevelIns& L = P.lv[blockIdx.y];
  const GridView& g = L.g;
  const unsigned n_items = g.call[1];
  const double tau = static_cast<double>(L.p.truncation_distance);
  const double unit = (tau + tau) / static_cast<double>(kFastUnits);
  const float maxw = L.p.maximum_weight;
  const float frac = g.weight_resolution - floorf(g.weight_resolution);
  const bool closed_form = fabsf(frac - 0.5f) > 0.05f && g.weight_resolution >= 1.0f && maxw == g.max_weight;
  // log(wmax / (wmax + 1)): the share of a saturated voxel's value that survives one update
  const float wmax = 32766.f * g.weight_scale;
  const float log_q_sat = static_cast<float>(log(static_cast<double>(wmax) / (static_cast<double>(wmax) + 1.0)));
  __shared__ unsigned long long s_acc[kVoxelsPerBlock];
  __shared__ int s_last;
  const unsigned t = threadIdx.x;
  for (unsigned w = blockIdx.x; w < n_items; w += gridDim.x) {
    const uint4 item = g.work[w];
    const uint32_t slot = item.x & 0xFFFFFFu, part = item.x >> 24;
    const unsigned cnt = item.z & 0xFFFFFFu, parts = (item.z >> 24) + 1u;
    const uint32_t* __restrict__ r = recs + item.y;
    s_acc[t] = 0ull;
    s_acc[t + kFastApplyThreads] = 0ull;
    __syncthreads();
    // records: eight loads of a thread in flight, then their LDS atomics
    for (unsigned b = 0; b < cnt; b += 8u * kFastApplyThreads) {
      uint32_t v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const unsigned j = b + k * kFastApplyThreads + t;
        v[k] = j < cnt ? r[j] : 0xFFFFFFFFu;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const unsigned j = b + k * kFastApplyThreads + t;
        if (j < cnt) atomicAdd(&s_acc[v[k] & 511u], (1ull << kFastCountShift) | static_cast<unsigned long long>(v[k] >> 9));
      }
    }
    __syncthreads();
    unsigned long long a[2] = {s_acc[t], s_acc[t + kFastApplyThreads]};
    if (parts > 1u) {
      // a part of a cut bin: the tile goes to the scratch area (write-through, drained before the arrival is
      // counted: the hand-over of hg_match.hip's partial sums); the last part to arrive sums all of them
      unsigned long long* tiles = fs.tiles + static_cast<size_t>(item.w) * kVoxelsPerBlock;
      unsigned long long* mine = tiles + static_cast<size_t>(part) * kVoxelsPerBlock;
      __hip_atomic_store(mine + t, a[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mine + t + kFastApplyThreads, a[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) {
        const unsigned arrived = atomicAdd(&g.bin_count[slot], 1u);
        s_last = arrived == parts - 1u ? 1 : 0;
        if (s_last) g.bin_count[slot] = 0u;  // ready for the next call's count pass
      }
      __syncthreads();
      const bool last = s_last != 0;
      if (last) {
        a[0] = 0ull;
        a[1] = 0ull;
        for (unsigned q = 0; q < parts; ++q) {
          const unsigned long long* tq = tiles + static_cast<size_t>(q) * kVoxelsPerBlock;
          a[0] += __hip_atomic_load(tq + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          a[1] += __hip_atomic_load(tq + t + kFastApplyThreads, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __syncthreads();  // (s_last and s_acc are reused by the next item)
      if (!last) continue;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (a[h] == 0ull) continue;
      const double m = static_cast<double>(a[h] >> kFastCountShift);
      const double sum = static_cast<double>(a[h] & ((1ull << kFastCountShift) - 1ull)) * unit - m * tau;
      const double mean = sum / m;
      uint32_t* cell = g.voxels + static_cast<size_t>(slot) * kVoxelsPerBlock + t + h * kFastApplyThreads;
      const uint32_t code = *cell;
      const double d0 = static_cast<double>(value_to_tsd(g, code & 0xFFFFu));
      uint32_t wcode;
      const double A = closed_form ? static_cast<double>(fast_survival_f(g, code >> 16, static_cast<unsigned>(a[h] >> kFastCountShift), log_q_sat, &wcode))
                                   : fast_survival_walk(g, maxw, code >> 16, m, &wcode);
      // every update is a convex combination, so the m updates carry the weight 1 - A together; they enter at
      // their mean (their individual shares depend on the arrival order)
      const double d = d0 * A + mean * (1.0 - A);
      *cell = (tsd_to_value(g, static_cast<float>(d)) | kUpdateMarker) | (wcode << 16);
    }
    // (a thread clears and re-reads only its own two words of s_acc: no barrier needed before the next item)
  }
}

// Bitonic sort of m (power of two) key/value pairs in LDS by all kBinThreads threads. Keys are
// unique ((voxel, seq) records; padding is 0xFFFFFFFF), so the result is the (voxel, seq) order.
__device__ inline void bitonic_sort_kv(uint32_t* k, uint32_t* v, unsigned m, unsigned tid) {
  for (unsigned size = 2; size <= m; size <<= 1) {
    for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
      for (unsigned t = tid; t < (m >> 1); t += kBinThreads) {
        const unsigned i = ((t & ~(stride - 1u)) << 1) | (t & (stride - 1u));
        const unsigned j = i | stride;
        const bool up = (i & size) == 0u;
        const uint32_t a = k[i], b = k[j];
        if ((a > b) == up) {
          k[i] = b;
          k[j] = a;
          const uint32_t va = v[i];
          v[i] = v[j];
          v[j] = va;
        }
      }
      __syncthreads();
    }
  }
}
constexpr unsigned kRankMaxGroup = 1024;  // larger per-voxel groups are ordered by the bitonic sort (break-even ~1100)

// grid (G, levels), 512 threads, loops over the level's work items (block, voxel range).
#ifdef HG_BIN_STAMPS
#define BIN_STAMP(i) do { if (threadIdx.x == 0) stamps[(static_cast<size_t>(order) * 4096 + wi) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BIN_STAMP(i) do {} while (0)
#endif
// A bin of at most kSmallBin records is handled by ONE wavefront with wavefront-level LDS traffic only
// (no workgroup barrier): most touched blocks of a scan are of this kind (mean bin size 280 / 700
// records at 0.05 / 0.10 m), and as 512-thread work items they spent their time in a dozen barriers
// around a few hundred records each while holding a third of a CU. Same steps as the workgroup path:
// voxel histogram, exclusive scan, grouping by voxel, rank by seq inside each group, one chain per
// voxel in reference order (the non-empty voxels are dealt out to the lanes 64 at a time).
__device__ inline void wave_sync_lds() {  // lanes of ONE wavefront exchanging data through LDS
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// CAP = records the bin may hold (512, or 256 inside k_bin_apply where eight wavefronts share the
// workgroup's LDS); lds: 512 + 3 * CAP words.
template <unsigned CAP>
__device__ inline void apply_small_bin(const GridView& g, float maximum_weight, uint32_t slot, unsigned n,
                                       const uint32_t* __restrict__ bk, const uint32_t* __restrict__ bv,
                                       uint32_t* lds) {
  const unsigned lane = threadIdx.x & (kWave - 1);
  // (Round 4, measured and dropped: requesting the block's 16 lines here, and the work items' voxels at item
  // start, so that the chains' reads at the very end hit the L2 -- a scan stream over 64 rooms, 320 MB of voxel
  // blocks, ran at 8.9k scans/s either way and k_bin_apply took 7 spills at its 64-register cap: whatever makes
  // the apply pass take 88 us per scan on maps in HBM against 37 us on cached ones, it is not these reads.)
  uint32_t* vox = g.voxels + static_cast<size_t>(slot) * kVoxelsPerBlock;
  uint32_t* cb = lds;                 // per voxel: count (after the scan: count | base << 16)
  uint32_t* gk = lds + kVoxelsPerBlock;  // records grouped by voxel; later the list of non-empty voxels
  uint32_t* gv = gk + CAP;
  uint32_t* sv = gv + CAP;            // values in (voxel, seq) order
  constexpr unsigned kPer = CAP / kWave;              // records per lane
  constexpr unsigned kVox = kVoxelsPerBlock / kWave;  // voxels per lane: 8
#pragma unroll
  for (unsigned i = 0; i < kVox; ++i) cb[lane + kWave * i] = 0u;
  wave_sync_lds();
  uint32_t k[kPer];
#pragma unroll
  for (unsigned i = 0; i < kPer; ++i) {
    const unsigned r = lane + kWave * i;
    k[i] = r < n ? bk[r] : 0xFFFFFFFFu;
  }
#pragma unroll
  for (unsigned i = 0; i < kPer; ++i)
    if (k[i] != 0xFFFFFFFFu) atomicAdd(&cb[k[i] >> kSeqBits], 1u);
  wave_sync_lds();
  // exclusive scan over the 512 counts: lane l owns voxels [8 l, 8 l + 8)
  {
    uint32_t c[kVox];
    unsigned sum = 0;
#pragma unroll
    for (unsigned j = 0; j < kVox; ++j) { c[j] = cb[kVox * lane + j]; sum += c[j]; }
    unsigned incl = sum;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const unsigned t = __shfl_up(incl, off);
      if (static_cast<int>(lane) >= off) incl += t;
    }
    unsigned run = incl - sum;
#pragma unroll
    for (unsigned j = 0; j < kVox; ++j) { cb[kVox * lane + j] = c[j] | (run << 16); run += c[j]; }
  }
  wave_sync_lds();
  // group by voxel: the base half of the voxel's word doubles as its cursor (it ends at base + count)
  {
    uint32_t v[kPer];
#pragma unroll
    for (unsigned i = 0; i < kPer; ++i) {
      const unsigned r = lane + kWave * i;
      v[i] = r < n ? bv[r] : 0u;
    }
#pragma unroll
    for (unsigned i = 0; i < kPer; ++i) {
      if (k[i] != 0xFFFFFFFFu) {
        const unsigned p = atomicAdd(&cb[k[i] >> kSeqBits], 0x10000u) >> 16;
        gk[p] = k[i];
        gv[p] = v[i];
      }
    }
  }
  wave_sync_lds();
  // rank inside the voxel's group = number of its records with a smaller seq (groups are small); a
  // lane ranks the records at the positions it owns in the grouped arrays
#pragma unroll 1
  for (unsigned i = 0; i < kPer; ++i) {
    const unsigned p = lane + kWave * i;
    if (p < n) {
      const uint32_t key = gk[p];
      const uint32_t e = cb[key >> kSeqBits];
      const unsigned b1 = e >> 16, b0 = b1 - (e & 0xFFFFu);
      unsigned rank = 0;
      for (unsigned j = b0; j < b1; j += 8) {  // 8 independent LDS reads in flight
        uint32_t a[8];
#pragma unroll
        for (unsigned u = 0; u < 8; ++u) a[u] = gk[min(j + u, b1 - 1u)];
#pragma unroll
        for (unsigned u = 0; u < 8; ++u) rank += (j + u < b1 && a[u] < key) ? 1u : 0u;
      }
      sv[b0 + rank] = gv[p];
    }
  }
  wave_sync_lds();
  // list of the non-empty voxels (gk is free now), then one chain per voxel, 64 voxels at a time
  unsigned m;
  {
    unsigned mine = 0;
#pragma unroll
    for (unsigned j = 0; j < kVox; ++j) mine += (cb[kVox * lane + j] & 0xFFFFu) ? 1u : 0u;
    unsigned incl = mine;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const unsigned t = __shfl_up(incl, off);
      if (static_cast<int>(lane) >= off) incl += t;
    }
    m = __shfl(incl, kWave - 1);
    unsigned pos = incl - mine;
#pragma unroll
    for (unsigned j = 0; j < kVox; ++j)
      if (cb[kVox * lane + j] & 0xFFFFu) gk[pos++] = kVox * lane + j;
  }
  wave_sync_lds();
  for (unsigned base = 0; base < m; base += kWave) {
    const unsigned idx = base + lane;
    if (idx < m) {
      const unsigned vx = gk[idx];
      const uint32_t e = cb[vx];
      vox[vx] = update_chain_unit(g, maximum_weight, vox[vx], sv + ((e >> 16) - (e & 0xFFFFu)), e & 0xFFFFu);
    }
  }
  wave_sync_lds();  // the next bin of this wavefront reuses the arrays
}

// The whole bins of a level (work items [call[2], call[1])), one wavefront each (batched inserts).
constexpr int kSmallThreads = 128;  // two bins per workgroup
__device__ __forceinline__ void bin_apply_small_body(const LevelIns& L, unsigned bx, unsigned gstride,
                                                     const uint32_t* __restrict__ rec_keys,
                                                     const uint32_t* __restrict__ rec_vals) {
  const GridView& g = L.g;
  __shared__ uint32_t smem[(kSmallThreads / kWave) * 4 * kSmallBin];
  const unsigned nwork = g.call[1], n_large = min(g.call[2], nwork);
  const unsigned wave = threadIdx.x / kWave;
  for (unsigned idx = n_large + bx * (kSmallThreads / kWave) + wave; idx < nwork; idx += gstride * (kSmallThreads / kWave)) {
    const uint4 it = g.work[idx];
    apply_small_bin<kSmallBin>(g, L.p.maximum_weight, it.x, it.z, rec_keys + it.w, rec_vals + it.w,
                               smem + wave * 4 * kSmallBin);
  }
}
// grid (G, jobs * levels)
__global__ __launch_bounds__(kSmallThreads) void k_bin_apply_small_jobs(const InsertJob* __restrict__ jobs, int levels) {
  // the level's description is copied out of the job table: read through the table pointer the
  // compiler reloads it inside the loops (possible aliasing with the voxel stores), which made the
  // apply pass 40 % slower than with the pyramid as a kernel argument
  const InsertJob& J = jobs[blockIdx.y / levels];
  const LevelIns L = level_as_device(J.P.lv[blockIdx.y % levels]);
  bin_apply_small_body(L, blockIdx.x, gridDim.x, as_device(J.rec_keys), as_device(J.rec_vals));
}

// `order` = position of the level in dispatch order (0 = coarsest), `bx` of `gstride` = workgroup of
// the level's grid-stride loop over its work items.
// Units of a MERGED stream apply (round 6, k_stream_units): a unit is a run of consecutive work items that ONE
// workgroup (wg_units) or ONE wavefront (wave_units) applies in list order -- the items of one (block, voxel slice)
// for the scans of a group, in scan order. Different units never share a voxel, so the launch needs no order between
// workgroups, and a voxel still receives its updates scan by scan, return by return. Null: every item on its own.
struct ApplyUnits {
  const uint2* wg_units;    // {first item, items}; kUnitTiers tables of `tier_stride` entries, heaviest blocks first
  const uint2* wave_units;
  const uint32_t* counts;   // [0 .. kUnitTiers) workgroup units per tier, [kUnitTiers] wavefront units
  unsigned tier_stride;
};
constexpr unsigned kUnitTiers = 4;  // by the largest bin of the block: >= 16384, >= 4096, >= 1024 records, the rest
__device__ __forceinline__ void bin_apply_body(const LevelIns& L, unsigned order, unsigned bx, unsigned gstride,
                                               const uint32_t* __restrict__ rec_keys,
                                               const uint32_t* __restrict__ rec_vals, bool small_in_kernel
#ifdef HG_BIN_STAMPS
                                               , long long* stamps
#endif
                                               , const ApplyUnits* units = nullptr) {
  const GridView& g = L.g;
  // one LDS pool, laid out for a workgroup item (3 x 512 + 4 x kBinCap words) or for eight small bins
  // (one per wavefront, 512 + 3 x 256 words each)
  constexpr unsigned kSmallWords = kVoxelsPerBlock + 3u * kSmallBinInKernel;
  constexpr unsigned kItemWords = 3u * 512u + 4u * kBinCap;
  constexpr unsigned kPoolWords = kItemWords > (kBinThreads / kWave) * kSmallWords ? kItemWords : (kBinThreads / kWave) * kSmallWords;
  __shared__ uint32_t pool[kPoolWords];
  unsigned* hist = pool;              // records per voxel (inside the item's voxel range)
  unsigned* base = pool + 512;        // exclusive prefix of hist
  unsigned* cursor = pool + 1024;
  uint32_t* gk = pool + 1536;         // grouped by voxel, arbitrary order inside a group
  uint32_t* gv = gk + kBinCap;
  uint32_t* sv = gv + kBinCap;        // values in (voxel, seq) order
  uint32_t* tk = sv + kBinCap;        // keys of the slice's records as read (compact list)
  uint32_t* tv = sv;                  // their values: sv is free until the rank step
  // (the item path's few scalars live behind its arrays inside the pool, which the small-bin layout sizes:
  // 40 KiB in all, four workgroups per CU as far as LDS goes)
  static_assert(kItemWords + 20u + 3u * kSegUnits <= kPoolWords, "pool");
  static_assert(kBinThreads / kWave == static_cast<int>(kSegUnits), "seg_chains");
  static_assert(kHeavyLdsVals + kSegWords + kSegListWords <= kPoolWords, "heavy tail");
  unsigned& s_hi = pool[kItemWords];
  unsigned& s_m = pool[kItemWords + 1];
  unsigned& s_big = pool[kItemWords + 2];
  unsigned* s_scan16 = pool + kItemWords + 3;
  // Long chains. A voxel with hundreds or thousands of updates in one scan (next to the sensor, next to a wall) is a
  // sequential chain of 0.061 us per update on ONE lane and used to be the end of the whole launch. Such voxels are
  // not applied in their pass: the pass writes their ordered update values to device memory (heavy_vals) and notes
  // them in the workgroup's list (heavy_list); when the workgroup has run out of work items it applies its
  // deferred voxels by the segmented evaluation of hg_chain.h (seg_chains: eight wavefronts x 64 candidate start
  // codes, bit-identical results, n / 8 steps instead of n). Behind the item loop, not inside it: there the code
  // -- as much again as the rest of the kernel -- cost the ordinary path 27 to 44 spilled registers, inlined or
  // called; here nothing else is live.
  unsigned& s_nheavy = pool[kItemWords + 19];     // long chains of the running pass
  uint32_t* def_b0 = pool + kItemWords + 20;      // per long chain of the pass: first value in sv,
  uint32_t* def_cnt = def_b0 + kSegUnits;         //   number of updates,
  uint32_t* def_off = def_cnt + kSegUnits;        //   offset of its values in heavy_vals
  unsigned n_deferred = 0;                        // entries of this workgroup's list (uniform)
#ifdef HG_DEFER_LONG_CHAINS
  const bool defer_ok = L.heavy_vals != nullptr && div_in_range_ok(g);
#else
  constexpr bool defer_ok = false;  // (everything that hangs off it, the tail included, is compiled away)
#endif
  uint4* const my_heavy = defer_ok ? L.heavy_list + static_cast<size_t>(bx) * kHeavyPerWg : nullptr;
  unsigned tier_end[kUnitTiers] = {0u, 0u, 0u, 0u};  // (units) running totals of the tiers: the list is their concatenation
  if (units) {
    unsigned acc = 0;
#pragma unroll
    for (unsigned t = 0; t < kUnitTiers; ++t) { acc += units->counts[t]; tier_end[t] = acc; }
  }
  const unsigned nwork = units ? tier_end[kUnitTiers - 1u] : min(g.call[2], g.call[1]);  // the slices of large bins; whole bins: k_bin_apply_small
  const unsigned tid = threadIdx.x;
  for (unsigned oi = bx; oi < nwork; oi += gstride) {
   unsigned wi_first = oi, wi_end = oi + 1u;
   if (units) {
     unsigned t = 0, before = 0;
#pragma unroll
     for (unsigned q = 0; q + 1u < kUnitTiers; ++q)
       if (oi >= tier_end[q]) { t = q + 1u; before = tier_end[q]; }
     const uint2 u = units->wg_units[static_cast<size_t>(t) * units->tier_stride + (oi - before)];
     wi_first = u.x;
     wi_end = u.x + u.y;
   }
   for (unsigned wi = wi_first; wi < wi_end; ++wi) {
    const uint4 item = g.work[wi];
    const uint32_t slot = item.x;
    const unsigned n = item.z;  // all 32 bits: a bin may hold every record of the scan
    // The slices of a bin interleave its voxels (slice_of below): the heavy voxels
    // of a block are spatial neighbours, contiguous slices would queue their chains in one workgroup.
    // A slice has 512 / S voxels, so the 512 counters of the item are spent on S seq BUCKETS per voxel:
    // a record is filed under l = (v / S) * S + bucket, bucket = (seq - first seq of the voxel) >> shift of the
    // voxel (monotone in seq). Order by (l, seq) is order by (voxel, seq), and the rank step -- quadratic in the
    // group size -- works on groups up to S times smaller: the few voxels of a heavy slice hold hundreds of
    // records each (a slice of a 15k-record bin spent 12.6 of its 45 us ranking before). Items that cannot
    // keep their records in LDS (below) use bucket 0 only.
    const unsigned per_slice = ((item.y >> 10) & 1023u) - (item.y & 1023u);  // 512 / S, a power of two
    const unsigned s_bits = 9u - (31u - __builtin_clz(per_slice));  // log2(S)
    const unsigned s_mask = (1u << s_bits) - 1u;
    const unsigned slice = (item.y & 1023u) >> (9u - s_bits);  // this item's voxels: slice_of(v) == slice
    const uint32_t seq_mask_all = (1u << kSeqBits) - 1u;
    // slice of voxel v = x | y << 3 | z << 6: the low log2(S) bits of v ^ v >> 3 ^ v >> 6 (low three bits:
    // x ^ y ^ z). Plain v mod S put a wall square to the x axis -- one x for the whole block -- into ONE slice
    // whenever S <= 8, whose records then no longer fitted one pass. (v >> log2 S, slice) still names the voxel.
    auto slice_of = [&](uint32_t v) { return (v ^ (v >> 3) ^ (v >> 6)) & s_mask; };
    auto is_mine = [&](uint32_t k) { return k != 0xFFFFFFFFu && slice_of(k >> kSeqBits) == slice; };
    auto to_l = [&](uint32_t k) { return ((k >> kSeqBits) >> s_bits) << s_bits; };  // bucket 0 of the record's voxel
    auto voxel_of_l = [&](unsigned l) {
      unsigned v = (l >> s_bits) << s_bits;
      for (int i = static_cast<int>(s_bits) - 1; i >= 0; --i)  // bit i of v from the slice and v's higher bits
        v |= (((slice >> i) ^ (v >> (i + 3)) ^ (v >> (i + 6))) & 1u) << i;
      return v;
    };
    constexpr unsigned v_lo = 0u, v_hi = 512u;  // the item's counters: all 512 values of l
    const uint32_t* bk = rec_keys + item.w;
    const uint32_t* bv = rec_vals + item.w;
    BIN_STAMP(0);
#ifdef HG_BIN_STAMPS
    if (threadIdx.x == 0) stamps[(static_cast<size_t>(order) * 4096 + wi) * 8 + 6] = n;
#endif
    hist[tid] = 0;
    cursor[tid] = 0;  // (the first pass of the item finds it zero; later passes clear it again)
    {
      // first / last seq per voxel of a compact slice (<= 256 voxels): all ones / zero. Computed from a copy of
      // tid the optimiser cannot see through: hoisted out of the item loop the value was spilled to scratch
      // memory in the prologue of EVERY wavefront (6 MB of writes per launch for one select).
      unsigned t2 = tid;
      asm volatile("" : "+v"(t2));
      base[tid] = t2 < 256u ? 0xFFFFFFFFu : 0u;
    }
    if (tid == 0) { s_big = 0; s_m = 0; s_nheavy = 0; }
    __syncthreads();
    // a whole bin that fits one pass: its records go straight into the LDS list (no filter, no buckets)
    const bool single = n <= static_cast<unsigned>(kBinCap) && s_bits == 0u;
    bool compact = false;
    if (single) {
      uint32_t k4[4], v4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned i = u * kBinThreads + tid;
        k4[u] = i < n ? bk[i] : 0xFFFFFFFFu;
        v4[u] = i < n ? bv[i] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned i = u * kBinThreads + tid;
        if (i < n) {
          tk[i] = k4[u];  // (one slice: l is the voxel, the key is keyed by l as it stands)
          tv[i] = v4[u];
          atomicAdd(&hist[k4[u] >> kSeqBits], 1u);
        }
      }
      if (tid == 0) s_m = n;
      compact = true;
      __syncthreads();
    } else {
      // A slice of a larger bin scans the WHOLE bin for its voxels' records; with S slices per bin that scan
      // is most of the pass's instructions, so it does nothing but filter: keys only, eight in flight, the
      // slice's records appended to a list in LDS (tk: key, tv: record index) by ballots, with one LDS atomic
      // per wavefront and eight rows of keys. Everything else -- values, buckets, histogram, grouping -- then
      // works on the list.
      const unsigned lane = tid & (kWave - 1);
      for (unsigned i0 = 0; i0 < n; i0 += 8 * kBinThreads) {
        uint32_t k8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const unsigned i = i0 + u * kBinThreads + tid;
          k8[u] = i < n ? bk[i] : 0xFFFFFFFFu;
        }
        unsigned long long mb[8];
        unsigned tot = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const unsigned i = i0 + u * kBinThreads + tid;
          mb[u] = __ballot(i < n && slice_of(k8[u] >> kSeqBits) == slice);
          tot += static_cast<unsigned>(__popcll(mb[u]));
        }
        if (tot) {  // wave-uniform
          unsigned p0 = 0;
          if (lane == 0) p0 = atomicAdd(&s_m, tot);
          p0 = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(p0)));
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const unsigned p = p0 + __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mb[u] >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mb[u]), 0u));
            if (((mb[u] >> lane) & 1ull) && p < static_cast<unsigned>(kBinCap)) { tk[p] = k8[u]; tv[p] = i0 + u * kBinThreads + tid; }
            p0 += static_cast<unsigned>(__popcll(mb[u]));
          }
        }
      }
      __syncthreads();
      const unsigned m_slice = s_m;
      compact = m_slice <= static_cast<unsigned>(kBinCap);
      if (compact) {
        // slots of this thread: j = tid + 512 u, u < 4
        uint32_t kk[4], vv[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned j = tid + u * kBinThreads;
          ok[u] = j < m_slice;
          kk[u] = ok[u] ? tk[j] : 0u;
          vv[u] = ok[u] ? bv[tv[j]] : 0u;  // the values of the slice's records only
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (ok[u]) {
            const unsigned vl = (kk[u] >> kSeqBits) >> s_bits, sq = kk[u] & seq_mask_all;
            atomicMin(&base[vl], sq);
            atomicMax(&base[256u + vl], sq);
          }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned j = tid + u * kBinThreads;
          if (ok[u]) {
            const unsigned vl = (kk[u] >> kSeqBits) >> s_bits, sq = kk[u] & seq_mask_all;
            const unsigned first = base[vl], range = base[256u + vl] - first;
            const unsigned bits = range ? 32u - static_cast<unsigned>(__builtin_clz(range)) : 0u;
            const unsigned sh = bits > s_bits ? bits - s_bits : 0u;  // (range >> sh) < S
            const unsigned l = (vl << s_bits) | ((sq - first) >> sh);
            atomicAdd(&hist[l], 1u);
            tk[j] = (l << kSeqBits) | sq;  // keyed by l from here on
            tv[j] = vv[u];
          }
        }
        __syncthreads();
      } else {
        // the slice's records do not fit the list (thousands of records on its voxels): histogram per
        // voxel by a second scan, the grouping passes read the bin again
        for (unsigned i0 = 0; i0 < n; i0 += 4 * kBinThreads) {
          uint32_t k4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const unsigned i = i0 + u * kBinThreads + tid;
            k4[u] = i < n ? bk[i] : 0xFFFFFFFFu;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (is_mine(k4[u])) atomicAdd(&hist[to_l(k4[u])], 1u);
        }
        __syncthreads();
      }
    }
    BIN_STAMP(1);
    // exclusive prefix of hist over the 512 voxels -> base
    {
      unsigned tot;
      // (a voxel group too large for the rank step sends the item's passes through the bitonic sort)
      if (hist[tid] > kRankMaxGroup) s_big = 1;  // benign race: same value
      base[tid] = block_exclusive_scan(hist[tid], s_scan16, &tot);
    }
    __syncthreads();
    unsigned lo = v_lo;
    while (lo < v_hi) {
      // voxel range [lo, hi): the longest prefix of voxels whose records fit one LDS pass
      const unsigned b_lo = base[lo];
      unsigned hi = v_hi;
      if (!compact) {  // (a whole small bin, or a slice whose records fit one pass, is ONE range)
        if (tid == 0) s_hi = 1024;
        __syncthreads();
        {
          const unsigned end_v = base[tid] + hist[tid] - b_lo;  // records in [lo, tid]
          const bool fits = tid >= lo && tid < v_hi && end_v <= static_cast<unsigned>(kBinCap);
          const bool next_fits = tid + 1 < v_hi && (base[tid + 1] + hist[tid + 1] - b_lo) <= static_cast<unsigned>(kBinCap);
          if (fits && !next_fits) s_hi = tid + 1;
          if (tid == lo && !fits) s_hi = lo;  // the first voxel alone does not fit
        }
        __syncthreads();
        hi = s_hi;
      }
      unsigned cnt = 0;
      if (hi == lo) {
        // One voxel alone holds more records than an LDS pass (degenerate geometry: thousands of
        // rays through one voxel). Its chain is applied in seq-ordered rounds of <= kBinCap records:
        // a 512-bucket histogram of the remaining seq range picks each round's threshold (if even
        // the first bucket is too large, the histogram is refined inside it; seq is unique, so
        // buckets of one seq value hold at most one record), the round is gathered, sorted, applied.
        const unsigned total_v = hist[lo];
        unsigned done_v = 0;
        const uint32_t seq_mask = (1u << kSeqBits) - 1u;
        const unsigned lo_v = voxel_of_l(lo);  // the voxel id as the records carry it
        uint32_t* cell = g.voxels + static_cast<size_t>(slot) * kVoxelsPerBlock + lo_v;
        // (one long chain: deferred when there is room -- every round's sorted values go to heavy_vals in order --,
        // else applied round by round by one lane as before)
        UnitChain chain;
        unsigned off_v = 0xFFFFFFFFu;
        if (defer_ok && n_deferred < kHeavyPerWg) {
          __syncthreads();
          if (tid == 0) {
            unsigned o = atomicAdd(&g.call[3], total_v);
            if (o + total_v > L.heavy_capacity) o = 0xFFFFFFFFu;
            s_hi = o;
          }
          __syncthreads();
          off_v = s_hi;
        }
        if (off_v == 0xFFFFFFFFu && tid == 0) chain.begin(g, *cell);
        uint32_t cur_lo = 0;  // records with seq < cur_lo are applied (this form files a voxel under ONE l)
        unsigned* bucket = sv;            // 512 counters (sv is free until a round is sorted)
        unsigned* bucket_pre = sv + 512;  // their exclusive prefix
        while (done_v < total_v) {
          uint32_t range_hi = seq_mask + 1u;
          uint32_t T = range_hi;
          while (true) {
            unsigned S = 0;
            while (((range_hi - cur_lo - 1u) >> S) >= 512u) ++S;
            __syncthreads();
            bucket[tid] = 0;
            if (tid == 0) s_hi = 0;
            __syncthreads();
            for (unsigned i = tid; i < n; i += kBinThreads) {
              const uint32_t k = bk[i];
              const uint32_t sq = k & seq_mask;
              if ((k >> kSeqBits) == lo_v && sq >= cur_lo && sq < range_hi) atomicAdd(&bucket[(sq - cur_lo) >> S], 1u);
            }
            __syncthreads();
            {
              unsigned tot;
              const unsigned excl = block_exclusive_scan(bucket[tid], s_scan16, &tot);
              bucket_pre[tid] = excl;
              // number of leading buckets whose records fit one pass
              if (excl + bucket[tid] <= static_cast<unsigned>(kBinCap)) atomicMax(&s_hi, tid + 1u);
            }
            __syncthreads();
            // s_hi counts buckets b with prefix_incl(b) <= cap; prefix sums are monotone, so these
            // are exactly the first s_hi buckets
            const unsigned nb = s_hi;
            if (nb == 0u) {  // the first bucket alone is too large: refine inside it
              range_hi = cur_lo + (1u << S);
              continue;
            }
            const unsigned long long t_end = static_cast<unsigned long long>(cur_lo) +
                                             (static_cast<unsigned long long>(nb) << S);
            T = t_end < range_hi ? static_cast<uint32_t>(t_end) : range_hi;
            break;
          }
          __syncthreads();
          if (tid == 0) s_hi = 0;
          __syncthreads();
          for (unsigned i = tid; i < n; i += kBinThreads) {
            const uint32_t k = bk[i];
            const uint32_t sq = k & seq_mask;
            if ((k >> kSeqBits) == lo_v && sq >= cur_lo && sq < T) {
              const unsigned p = atomicAdd(&s_hi, 1u);
              if (p < static_cast<unsigned>(kBinCap)) { gk[p] = k; gv[p] = bv[i]; }
            }
          }
          __syncthreads();
          const unsigned m = min(s_hi, static_cast<unsigned>(kBinCap));
          unsigned m2 = 2;
          while (m2 < m) m2 <<= 1;
          for (unsigned i = m + tid; i < m2; i += kBinThreads) gk[i] = 0xFFFFFFFFu;
          __syncthreads();
          bitonic_sort_kv(gk, gv, m2, tid);
          if (off_v != 0xFFFFFFFFu) {
            for (unsigned i = tid; i < m; i += kBinThreads) L.heavy_vals[off_v + done_v + i] = gv[i];
          } else if (tid == 0) {
            chain.run(g, L.p.maximum_weight, gv, m);
          }
          __syncthreads();
          done_v += m;
          cur_lo = T;
          if (T > seq_mask) break;  // all seq values covered
        }
        if (off_v != 0xFFFFFFFFu) {
          if (tid == 0) my_heavy[n_deferred] = make_uint4(slot * kVoxelsPerBlock + lo_v, off_v, done_v, 0u);
          ++n_deferred;
        } else if (tid == 0) {
          *cell = chain.end();
        }
        hi = lo + 1;
      } else {
        cnt = base[hi - 1] + hist[hi - 1] - b_lo;
      }
      if (lo != v_lo) {  // not the first pass: the grouping of the previous one left its cursors behind
        cursor[tid] = 0;
        __syncthreads();
      }
      if (cnt) {
        // owners of long chains put them on the pass's list (thread t < 512 / S owns voxel t of the slice: its buckets
        // [t S, (t + 1) S) inside this pass) and reserve room for their values; the reservation's round trip runs
        // behind the grouping and ranking below
        const unsigned room = defer_ok ? min(kSegUnits, kHeavyPerWg - n_deferred) : 0u;
        bool listed = false;
        if (room) {
          const unsigned l0 = max(lo, tid << s_bits), l1 = min(hi, (tid + 1u) << s_bits);
          if (tid < per_slice && l0 < l1) {
            const unsigned b0 = base[l0] - b_lo;
            const unsigned mine = base[l1 - 1u] + hist[l1 - 1u] - b_lo - b0;
            if (mine >= kHeavyMin) {
              const unsigned e = atomicAdd(&s_nheavy, 1u);
              if (e < room) {
                unsigned off = atomicAdd(&g.call[3], mine);
                unsigned keep = mine;
                if (off + mine > L.heavy_capacity) { keep = 0u; off = 0u; }  // (cannot happen: capacity >= the level's records)
                def_b0[e] = b0;
                def_cnt[e] = keep;
                def_off[e] = off;
                my_heavy[n_deferred + e] = make_uint4(slot * kVoxelsPerBlock + voxel_of_l(l0), off, keep, 0u);
                listed = keep != 0u;
              }
            }
          }
        }
        // group by voxel (arbitrary order inside a group)
        if (compact) {  // (one pass: lo = 0, hi = 512; the list is keyed by l already)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const unsigned j = tid + u * kBinThreads;
            if (j < s_m) {
              const uint32_t k = tk[j];
              const unsigned l = k >> kSeqBits;
              const unsigned p = base[l] + atomicAdd(&cursor[l], 1u);
              gk[p] = k;
              gv[p] = tv[j];
            }
          }
        }
        for (unsigned i0 = 0; !compact && i0 < n; i0 += 4 * kBinThreads) {
          uint32_t k4[4], v4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const unsigned i = i0 + u * kBinThreads + tid;
            k4[u] = i < n ? bk[i] : 0xFFFFFFFFu;
            v4[u] = i < n ? bv[i] : 0u;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const unsigned v = to_l(k4[u]);
            if (is_mine(k4[u]) && v >= lo && v < hi) {
              const unsigned p = base[v] - b_lo + atomicAdd(&cursor[v], 1u);
              gk[p] = (v << kSeqBits) | (k4[u] & seq_mask_all);  // keyed by l from here on
              gv[p] = v4[u];
            }
          }
        }
        __syncthreads();
        BIN_STAMP(2);
        if (s_big) {
          // large groups: one bitonic sort of the whole pass by (voxel, seq); the groups keep their
          // places because the grouped layout is already ordered by voxel
          unsigned m2 = 2;
          while (m2 < cnt) m2 <<= 1;
          for (unsigned i = cnt + tid; i < m2; i += kBinThreads) gk[i] = 0xFFFFFFFFu;
          __syncthreads();
          bitonic_sort_kv(gk, gv, m2, tid);
          for (unsigned i = tid; i < cnt; i += kBinThreads) sv[i] = gv[i];
        } else
        // order each group by seq: rank = number of group members with a smaller key
        for (unsigned i = tid; i < cnt; i += kBinThreads) {
          const uint32_t k = gk[i];
          const unsigned v = k >> kSeqBits;
          const unsigned b0 = base[v] - b_lo, b1 = b0 + hist[v];
          unsigned rank = 0;
          unsigned j = b0;
          for (; j + 4 <= b1; j += 4) {
            const uint32_t a0 = gk[j], a1 = gk[j + 1], a2 = gk[j + 2], a3 = gk[j + 3];
            rank += (a0 < k ? 1u : 0u) + (a1 < k ? 1u : 0u) + (a2 < k ? 1u : 0u) + (a3 < k ? 1u : 0u);
          }
          for (; j < b1; ++j) rank += (gk[j] < k) ? 1u : 0u;
          sv[b0 + rank] = gv[i];
        }
        __syncthreads();
        BIN_STAMP(3);
        // one thread per voxel applies its updates in reference order; a wavefront that holds a
        // long chain is the critical path of the whole insert: give it issue priority
        const unsigned heavy = min(s_nheavy, room);  // (uniform: written before the barriers above)
        for (unsigned e = 0; e < heavy; ++e) {
          const unsigned c = def_cnt[e], o = def_off[e], b = def_b0[e];
          for (unsigned i = tid; i < c; i += kBinThreads) L.heavy_vals[o + i] = sv[b + i];
        }
        n_deferred += heavy;
        {
          // thread t < 512 / S owns voxel t of the slice: its buckets [t S, (t + 1) S) inside this pass
          const unsigned l0 = max(lo, tid << s_bits), l1 = min(hi, (tid + 1u) << s_bits);
          unsigned b0 = 0, mine = 0;
          if (tid < per_slice && l0 < l1 && !listed) {
            b0 = base[l0] - b_lo;
            mine = base[l1 - 1u] + hist[l1 - 1u] - b_lo - b0;
          }
          if (__ballot(mine > 48u)) __builtin_amdgcn_s_setprio(3);
          if (mine) {
            uint32_t* cell = g.voxels + static_cast<size_t>(slot) * kVoxelsPerBlock + voxel_of_l(l0);
            *cell = update_chain_unit(g, L.p.maximum_weight, *cell, sv + b0, mine);
          }
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        if (tid == 0) s_nheavy = 0;  // (the next pass of the item registers behind its cursor barrier)
        BIN_STAMP(4);
      }
      lo = hi;
    }
    BIN_STAMP(5);  // (every pass ends behind a barrier)
   }
  }
  if (small_in_kernel) {
    // the wavefronts' small-bin scratch aliases hist / base of the work items above, and an item whose
    // slice holds no records ends without a barrier: none may start on a small bin before all have left
    // the last item
    __syncthreads();
    // the whole small bins of the level, one wavefront each: most touched blocks of a scan are of
    // this kind (65 % / 45 % of the bins at 0.05 / 0.10 m hold <= 256 records), and as 512-thread
    // items they queued for workgroup slots with a dozen barriers around a few hundred records
    const unsigned wave = tid / kWave;
    constexpr unsigned kWaves = kBinThreads / kWave;
    if (units) {
      const unsigned n_wave = units->counts[kUnitTiers];
      for (unsigned ui = bx * kWaves + wave; ui < n_wave; ui += gstride * kWaves) {
        const uint2 u = units->wave_units[ui];
        for (unsigned idx = u.x; idx < u.x + u.y; ++idx) {
          const uint4 it = g.work[idx];
          apply_small_bin<kSmallBinInKernel>(g, L.p.maximum_weight, it.x, it.z, rec_keys + it.w, rec_vals + it.w,
                                             pool + wave * kSmallWords);
          // the next item of the unit reads voxels this one wrote (other lanes of this wavefront)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
    } else {
    const unsigned n_all = g.call[1];
    for (unsigned idx = nwork + bx * kWaves + wave; idx < n_all; idx += gstride * kWaves) {
      const uint4 it = g.work[idx];
      apply_small_bin<kSmallBinInKernel>(g, L.p.maximum_weight, it.x, it.z, rec_keys + it.w, rec_vals + it.w,
                                         pool + wave * kSmallWords);
    }
    }
  }
#ifdef HG_DEFER_LONG_CHAINS
  if (n_deferred) {
    // ---- the workgroup's deferred long chains ("Long chains" above) ----
    // Up to eight voxels at a time whose values fit the LDS share the eight wavefronts (seg_chains hands out the
    // segments in proportion to their lengths); a voxel with more values than that is applied chunk by chunk, its
    // code handed on in the list. Everything below is derived from a copy of the thread index the optimiser
    // cannot see through, or it hoists lane predicates of this tail into the kernel's prologue and spills them.
    unsigned t2 = tid;
    asm volatile("" : "+v"(t2));
    uint32_t* const hvals = pool;
    uint32_t* const hscratch = pool + kHeavyLdsVals;
    uint32_t* const hlist = hscratch + kSegWords;
#ifdef HG_BIN_STAMPS
    if (threadIdx.x == 0 && bx < nwork) {
      stamps[(static_cast<size_t>(order) * 4096 + bx) * 8 + 7] = __builtin_amdgcn_s_memrealtime();
      stamps[(static_cast<size_t>(order) * 4096 + 3072 + (bx & 1023u)) * 8 + 0] = __builtin_amdgcn_s_memrealtime();
      unsigned long long tot = 0;
      for (unsigned q = 0; q < n_deferred; ++q) tot += my_heavy[q].z;
      stamps[(static_cast<size_t>(order) * 4096 + bx) * 8 + 6] = static_cast<long long>(n_deferred) * 1000000ll + static_cast<long long>(tot);
    }
#endif
    unsigned e = 0, chunk_done = 0;
    while (e < n_deferred) {
      __syncthreads();
      if (t2 < kWave) {
        // lanes 0..7 of the first wavefront read the next entries; the group is the longest prefix that fits
        const bool have = t2 < kSegUnits && e + t2 < n_deferred;
        uint4 ent = make_uint4(0u, 0u, 0u, 0u);
        if (have) ent = my_heavy[e + t2];
        unsigned cnt_l = have ? ent.z : 0u, off_l = ent.y;
        if (t2 == 0u) { cnt_l -= chunk_done; off_l += chunk_done; }
        const bool first_chunked = __shfl(static_cast<int>(cnt_l), 0) > static_cast<int>(kHeavyLdsVals);
        if (t2 == 0u && first_chunked) cnt_l = kHeavyLdsVals;
        unsigned incl = cnt_l;
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
          const unsigned t = static_cast<unsigned>(__shfl_up(static_cast<int>(incl), off));
          if (t2 >= static_cast<unsigned>(off)) incl += t;
        }
        const bool in_group = have && incl <= kHeavyLdsVals && !(first_chunked && t2 > 0u);
        const unsigned long long m = __ballot(in_group);
        const unsigned H = static_cast<unsigned>(__builtin_ctzll(~m));  // leading entries that are in
        if (t2 < H) {
          hlist[kSegB0 + t2] = incl - cnt_l;
          hlist[kSegCnt + t2] = cnt_l;
          hlist[kSegVox + t2] = ent.x;
          def_off[t2] = off_l;
          if (!(t2 == 0u && chunk_done)) hlist[kSegCode + t2] = g.voxels[ent.x];  // (a chunked voxel continues from the list)
        }
        if (t2 == 0u) {
          s_hi = H;
          s_m = first_chunked ? 1u : 0u;
        }
      }
      __syncthreads();
#ifdef HG_BIN_STAMPS
#define TAIL_STAMP(k) do { if (threadIdx.x == 0 && e == 0) stamps[(static_cast<size_t>(order) * 4096 + 3072 + (bx & 1023u)) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TAIL_STAMP(k) do {} while (0)
#endif
      TAIL_STAMP(1);
      const unsigned H = s_hi;
      for (unsigned h = 0; h < H; ++h) {
        const unsigned c = hlist[kSegCnt + h], o = def_off[h], b = hlist[kSegB0 + h];
        for (unsigned i = t2; i < c; i += kBinThreads) hvals[b + i] = L.heavy_vals[o + i];
      }
      __syncthreads();
      TAIL_STAMP(2);
      __builtin_amdgcn_s_setprio(3);
      seg_chains(chain_codec(g), L.p.maximum_weight, hvals, hlist, hscratch, g.voxels, H, 0u, t2);
      __builtin_amdgcn_s_setprio(0);
      __syncthreads();
      TAIL_STAMP(3);
      if (s_m) {  // (uniform) the first entry is larger than the LDS: next chunk, or done with it
        const unsigned total = my_heavy[e].z;
        chunk_done += kHeavyLdsVals;
        if (chunk_done >= total) { chunk_done = 0; ++e; }
      } else {
        e += H;
        chunk_done = 0;
      }
    }
    __syncthreads();
#ifdef HG_BIN_STAMPS
    if (threadIdx.x == 0 && bx < nwork) stamps[(static_cast<size_t>(order) * 4096 + bx) * 8 + 4] = __builtin_amdgcn_s_memrealtime();
#endif
  }
#endif  // HG_DEFER_LONG_CHAINS
}

__global__ __launch_bounds__(kBinThreads, HG_APPLY_WAVES) void k_bin_apply(PyramidIns P, const uint32_t* __restrict__ rec_keys,
                                                          const uint32_t* __restrict__ rec_vals
#ifdef HG_BIN_STAMPS
                                                          , long long* stamps
#endif
                                                          ) {
  // Workgroups are dispatched in index order and a third of the grid is resident at a time. Grid (G, levels):
  // the last (coarsest) level has the longest per-voxel chains, so it goes first. Grid (G * levels): the levels
  // take turns, workgroup b works on level b mod levels -- every level's work list starts with its heaviest
  // bins, so the heads of all lists start at once and no level waits for the one before it to drain.
  const bool turns = gridDim.y == 1u && P.levels > 1;
  const unsigned order = turns ? blockIdx.x % static_cast<unsigned>(P.levels) : blockIdx.y;
  const unsigned bx = turns ? blockIdx.x / static_cast<unsigned>(P.levels) : blockIdx.x;
  const unsigned gs = turns ? gridDim.x / static_cast<unsigned>(P.levels) : gridDim.x;
  bin_apply_body(P.lv[P.levels - 1 - order], order, bx, gs, rec_keys, rec_vals,
                 P.slice_records <= 0
#ifdef HG_BIN_STAMPS
                 , stamps
#endif
                 );
}
#ifndef HG_BIN_STAMPS
// grid (G, levels * jobs): y = level order * jobs + job, so the coarse levels of ALL jobs go first
__global__ __launch_bounds__(kBinThreads, 6) void k_bin_apply_jobs(const InsertJob* __restrict__ jobs, int njobs) {
  const InsertJob& J = jobs[blockIdx.y % njobs];
  const unsigned order = blockIdx.y / njobs;
  const LevelIns L = level_as_device(J.P.lv[J.P.levels - 1 - order]);  // a copy, see k_bin_apply_small_jobs
  bin_apply_body(L, order, blockIdx.x, gridDim.x, as_device(J.rec_keys), as_device(J.rec_vals), J.P.slice_records <= 0);
}
#endif

#ifndef HG_BIN_STAMPS
// ==========================================================================================
// Merged apply of a scan stream's group (round 6). Scan after scan, the apply launches of a stream were 67 % of its
// time: every launch drains three levels of ~2400 work items over 1024 workgroup slots and ends on its slowest
// item, 32 times per call. The scans of a group cannot share a launch item by item -- a voxel's updates of scan
// k + 1 must follow those of scan k -- but blocks (and voxel slices of a block) are independent of each other:
//   k_stream_offsets_jobs   per (scan, level): bin offsets as k_bin_offsets_jobs, bin counts left in place
//   k_stream_union          per level: the union of the group's touched blocks (a claim word per block, tagged with
//                           the group's epoch; the entry that claims a block appends it to the level's list)
//   k_stream_units          a wavefront per block of the union, lane = scan: every scan's count and offset for it, the
//                           block cut into S voxel slices from the LARGEST of those bins, per slice one UNIT = the
//                           slice's work items of the scans that touch the block, in scan order
//   k_bin_apply_stream      one launch for the group: a workgroup (or, for blocks whose bins all hold <= 256 records, a
//                           wavefront) takes a unit and applies its items one after the other (bin_apply_body)
// Every voxel belongs to exactly one unit, so it receives the updates of the group's scans in scan order and, inside
// a scan, in seq order: bit for bit the result of scan-by-scan insertion.
// ==========================================================================================
struct StreamGroup {
  uint32_t* claim[kMaxInsLevels];      // per level: epoch of the group that last claimed the block slot
  uint2* wg_units[kMaxInsLevels];      // per level: unit tables and their counters {workgroup units, wavefront units,
  uint2* wave_units[kMaxInsLevels];    //   items} (counts[level][0..2]); the items go to lv[].g.work of job 0
  uint32_t* counts;                    // kMaxInsLevels x 8 words (ApplyUnits::counts + the item cursor + the union's size), zero between groups
  uint32_t* union_list[kMaxInsLevels]; // per level: the block slots the group touches, in claim order (k_stream_union)
  uint32_t unit_capacity, item_capacity;
  uint32_t epoch;
  int slice_records;                   // as PyramidIns::slice_records of the stream (< 0: -records per slice of large bins)
};

__global__ __launch_bounds__(1024) void k_stream_offsets_jobs(const InsertJob* __restrict__ jobs, int levels) {
  const InsertJob& J = jobs[blockIdx.x / levels];
  const int level = blockIdx.x % levels;
  const LevelIns L = level_as_device(J.P.lv[level]);
  __shared__ unsigned s_scan[16];
  __shared__ unsigned s_base;
  const unsigned nt = L.g.call[0];
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (unsigned c0 = 0; c0 < nt; c0 += 1024u) {
    const unsigned i = c0 + threadIdx.x;
    const unsigned slot = i < nt ? L.g.touched[i] : 0u;
    const unsigned cnt = i < nt ? L.g.bin_count[slot] : 0u;
    unsigned chunk_total = 0;
    const unsigned excl = block_exclusive_scan(cnt, s_scan, &chunk_total);
    if (i < nt) L.g.bin_offset[slot] = static_cast<unsigned>(level) * J.records_per_level + s_base + excl;
    __syncthreads();
    if (threadIdx.x == 0) s_base += chunk_total;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    unsigned long long* upd = reinterpret_cast<unsigned long long*>(&L.g.counters[4]);
    if (J.P.shared) atomicAdd(upd, static_cast<unsigned long long>(s_base));
    else *upd = s_base + (J.P.accumulate ? *upd : 0ull);  // U of this call
    publish_flags(J.P, level);
  }
}

// The union of the group's touched blocks, per level: grid (kStreamUnitWgs, levels), 256 threads; jobs = the group's
// scans in order (at most kStreamGroupMax). The entries of all scans are one flat list in chunks of 256; whichever
// entry claims a block first appends it to the level's union list (one device atomic per wavefront).
// (Rounds of this step, docs/EXPERIMENTS.md: the claimer used to BE the block's owner and emit its units. Ownership
// is first come, first served, and the workgroups dispatched first -- the ones holding scan 0's entries -- won almost
// every claim: nine workgroups did a level's owner work while 247 waited, 72 - 93 us per group of 32 scans. Now the
// claim only builds the list, and k_stream_units spreads the list over the chip.)
constexpr int kStreamGroupMax = 32;
constexpr unsigned kStreamUnitWgs = 256;  // (a chunk of 256 entries each for a group of 32 scans: one pass per workgroup)
constexpr unsigned kUnionCount = 6;        // word of a level's counters that holds the size of its union list
typedef __attribute__((address_space(1))) uint32_t hg_gu32;  // (device-memory addresses: through generic pointers the
                                                              // loads below are FLAT operations, which also count against the LDS / scalar counter)
__global__ __launch_bounds__(256) void k_stream_union(const InsertJob* __restrict__ jobs, int njobs, int levels, StreamGroup G) {
  const int level = blockIdx.y;
  uint32_t* const claim = G.claim[level];
  uint32_t* const counts = G.counts + 8 * level;
  hg_gu32* const list = (hg_gu32*)G.union_list[level];
  __shared__ unsigned s_nt[kStreamGroupMax + 1];  // prefix of the scans' touched counts
  __shared__ unsigned s_cnt[kStreamGroupMax];
  __shared__ const hg_gu32* s_touched[kStreamGroupMax];
  __shared__ unsigned s_won[5];
  if (threadIdx.x < static_cast<unsigned>(kStreamGroupMax)) {
    const bool in = static_cast<int>(threadIdx.x) < njobs;
    const GridView& gq = jobs[in ? threadIdx.x : 0u].P.lv[level].g;
    s_cnt[threadIdx.x] = in ? gq.call[0] : 0u;
    s_touched[threadIdx.x] = (const hg_gu32*)gq.touched;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned acc = 0;
    for (int j = 0; j < kStreamGroupMax; ++j) {
      s_nt[j] = acc;
      acc += s_cnt[j];
    }
    s_nt[kStreamGroupMax] = acc;
  }
  __syncthreads();
  const unsigned total = s_nt[kStreamGroupMax];
  const unsigned lane = threadIdx.x & 63u;
  for (unsigned c0 = blockIdx.x * 256u; c0 < total; c0 += gridDim.x * 256u) {
    const unsigned e = c0 + threadIdx.x;
    bool won = false;
    uint32_t slot = 0;
    if (e < total) {
      int j = 0;
#pragma unroll
      for (int q = 1; q < kStreamGroupMax; ++q) j = (e >= s_nt[q]) ? q : j;
      slot = s_touched[j][e - s_nt[j]];
      won = atomicExch(&claim[slot], G.epoch) != G.epoch;
    }
    // (one device atomic per workgroup and chunk: per wavefront, the 1 500 atomics of scans that share no block queued
    // on the one word for 18 us)
    const unsigned long long m = __ballot(won);
    const unsigned wave = threadIdx.x >> 6;
    if (lane == 0) s_won[wave] = static_cast<unsigned>(__popcll(m));
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned n = s_won[0] + s_won[1] + s_won[2] + s_won[3];
      s_won[4] = n ? atomicAdd(&counts[kUnionCount], n) : 0u;
    }
    __syncthreads();
    if (won) {
      unsigned at = s_won[4];
      for (unsigned w = 0; w < wave; ++w) at += s_won[w];
      list[at + static_cast<unsigned>(__popcll(m & ((1ull << lane) - 1ull)))] = slot;
    }
    __syncthreads();  // s_won is reused by the next chunk
  }
}

// Units and items of the union's blocks: grid (G, levels), 512 threads; HALF A WAVEFRONT PER BLOCK, lane = scan. A
// lane loads its scan's count and offset for the block (one round trip each), the half reduces the largest bin and the
// scans that touch the block, cuts the block into S voxel slices from the LARGEST bin, and emits per slice one UNIT =
// the slice's work items of the touching scans in scan order. The 16 R blocks of a batch (R per half) reserve their
// units and items through LDS with one device atomic per counter.
__global__ __launch_bounds__(512) void k_stream_units(const InsertJob* __restrict__ jobs, int njobs, int levels, StreamGroup G,
                                                      const uint32_t* rec_base /* the group's record buffer: jobs[j].rec_keys - rec_base = job j's offset in it */) {
  const int level = blockIdx.y;
  uint32_t* const counts = G.counts + 8 * level;
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(1))) u4v gu4;
  typedef __attribute__((address_space(1))) u2v gu2;
  gu4* const work = (gu4*)jobs[0].P.lv[level].g.work;
  const hg_gu32* const list = (const hg_gu32*)G.union_list[level];
  const unsigned slice_above = G.slice_records < 0 ? static_cast<unsigned>(-G.slice_records) : HG_SLICE_ABOVE;
  __shared__ unsigned s_need[kUnitTiers + 2], s_base[kUnitTiers + 2];
  __shared__ hg_gu32* s_bin_count[kStreamGroupMax];
  __shared__ const hg_gu32* s_bin_offset[kStreamGroupMax];
  __shared__ unsigned s_rec_off[kStreamGroupMax], s_seq_bits[kStreamGroupMax];
  if (threadIdx.x < static_cast<unsigned>(kStreamGroupMax)) {
    const bool in = static_cast<int>(threadIdx.x) < njobs;
    const InsertJob& Jq = jobs[in ? threadIdx.x : 0u];
    const GridView& gq = Jq.P.lv[level].g;
    s_bin_count[threadIdx.x] = (hg_gu32*)gq.bin_count;
    s_bin_offset[threadIdx.x] = (const hg_gu32*)gq.bin_offset;
    s_rec_off[threadIdx.x] = static_cast<unsigned>(Jq.rec_keys - rec_base);
    const unsigned rpl = Jq.records_per_level;
    s_seq_bits[threadIdx.x] = 32u - static_cast<unsigned>(__builtin_clz((rpl > 2u ? rpl : 2u) - 1u));
  }
  __syncthreads();
  const unsigned n_union = counts[kUnionCount];
  const unsigned lane = threadIdx.x & 63u, q = lane & 31u, half = threadIdx.x >> 5;  // half: 0 .. 15
  const int head = static_cast<int>(lane & 32u);                                        // first lane of this half
  hg_gu32* const my_count = s_bin_count[q];
  const hg_gu32* const my_offset = s_bin_offset[q];
  const unsigned my_rec_off = s_rec_off[q], my_seq_bits = s_seq_bits[q];
  const bool my_scan = static_cast<int>(q) < njobs;
  // A batch = 16 halves x R blocks each (two barriers and one device atomic per counter per batch): R = 1 while the
  // union is short (one room: 2 100 blocks on the finest level), R = 4 when it is long (scans that share no block:
  // 100 000 blocks per group -- at one block per wavefront and batch the barriers made that 123 us)
  auto batches = [&](auto r_tag) {
    constexpr unsigned R = decltype(r_tag)::value;
    for (unsigned b0 = blockIdx.x * 16u * R; b0 < n_union; b0 += gridDim.x * 16u * R) {  // (uniform)
      if (threadIdx.x < kUnitTiers + 2u) s_need[threadIdx.x] = 0u;
      __syncthreads();
      uint32_t slot[R];
      unsigned c[R], o[R], touch_mask[R], slices[R], cls[R], lu[R], lw[R];
#pragma unroll
      for (unsigned r = 0; r < R; ++r) {
        const unsigned b = b0 + half * R + r;
        slot[r] = b < n_union ? list[b] : 0xFFFFFFFFu;  // (uniform over the half)
      }
#pragma unroll
      for (unsigned r = 0; r < R; ++r) c[r] = (slot[r] != 0xFFFFFFFFu && my_scan) ? my_count[slot[r]] : 0u;
#pragma unroll
      for (unsigned r = 0; r < R; ++r) o[r] = c[r] ? my_offset[slot[r]] : 0u;
#pragma unroll
      for (unsigned r = 0; r < R; ++r) {
        o[r] = c[r] ? o[r] + my_rec_off : 0u;
        unsigned maxc = c[r];
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) maxc = max(maxc, static_cast<unsigned>(__shfl_xor(static_cast<int>(maxc), off)));
        touch_mask[r] = static_cast<unsigned>((__ballot(c[r] != 0u) >> head) & 0xFFFFFFFFull);
        const unsigned touching = static_cast<unsigned>(__popc(touch_mask[r]));
        const bool small = maxc <= kSmallBinInKernel;
        slices[r] = touching ? 1u : 0u;
        if (touching && !small) {
          const unsigned per_slice = maxc < HG_SLICE_THRESH ? HG_SLICE_BELOW : slice_above;
          while (slices[r] < 128u && maxc > slices[r] * per_slice) slices[r] <<= 1;
        }
        cls[r] = small ? kUnitTiers : (maxc >= 16384u ? 0u : maxc >= 4096u ? 1u : maxc >= 1024u ? 2u : 3u);
        lu[r] = lw[r] = 0u;
        if (q == 0u && touching) {
          lu[r] = atomicAdd(&s_need[cls[r]], slices[r]);
          lw[r] = atomicAdd(&s_need[kUnitTiers + 1u], slices[r] * touching);
        }
        lu[r] = static_cast<unsigned>(__shfl(static_cast<int>(lu[r]), head));
        lw[r] = static_cast<unsigned>(__shfl(static_cast<int>(lw[r]), head));
      }
      __syncthreads();
      if (threadIdx.x < kUnitTiers + 2u && s_need[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&counts[threadIdx.x], s_need[threadIdx.x]);
      __syncthreads();
#pragma unroll
      for (unsigned r = 0; r < R; ++r) {
        const unsigned touching = static_cast<unsigned>(__popc(touch_mask[r]));
        if (touching) {  // (uniform over the half)
          const unsigned u0 = s_base[cls[r]] + lu[r], w0 = s_base[kUnitTiers + 1u] + lw[r];
          if (u0 + slices[r] > G.unit_capacity || w0 + slices[r] * touching > G.item_capacity) {
            if (q == 0u) atomicOr(&jobs[0].P.lv[level].g.counters[1], kFlagWorkOverflow);  // cannot happen: the host sizes both from the group's records
          } else {
            gu2* const table = (gu2*)(cls[r] == kUnitTiers ? G.wave_units[level] : G.wg_units[level] + static_cast<size_t>(cls[r]) * G.unit_capacity);
            const unsigned rank = static_cast<unsigned>(__popc(touch_mask[r] & ((1u << q) - 1u)));  // scan order
            const unsigned step = 512u / slices[r];
            for (unsigned k = q; k < slices[r]; k += 32u) table[u0 + k] = u2v{w0 + k * touching, touching};
            if (c[r])
              for (unsigned k = 0; k < slices[r]; ++k)
                work[w0 + k * touching + rank] = u4v{slot[r], (k * step) | (((k + 1u) * step) << 10) | (my_seq_bits << 20), c[r], o[r]};
            }
          if (c[r]) my_count[slot[r]] = 0u;  // ready for the next call
        }
      }
    }
  };
  if (n_union >= 32u * gridDim.x) batches(std::integral_constant<unsigned, 4>{});
  else batches(std::integral_constant<unsigned, 1>{});
}

// grid (G, levels): one launch for the group; the pyramid of the group's first scan stands for all (same grids).
__global__ __launch_bounds__(kBinThreads, HG_APPLY_WAVES) void k_bin_apply_stream(PyramidIns P, const uint32_t* __restrict__ rec_keys,
                                                                 const uint32_t* __restrict__ rec_vals, StreamGroup G) {
  const unsigned order = blockIdx.y;
  const int level = P.levels - 1 - static_cast<int>(order);
  ApplyUnits units;
  units.wg_units = G.wg_units[level];
  units.wave_units = G.wave_units[level];
  units.counts = G.counts + 8 * level;
  units.tier_stride = G.unit_capacity;
  bin_apply_body(P.lv[level], order, blockIdx.x, gridDim.x, rec_keys, rec_vals, true, &units);
}
// <<<1, 64>>> behind the apply launch: the level counters and the scans' touched counts start from zero again
__global__ void k_stream_reset(StreamGroup G, int levels, const InsertJob* __restrict__ jobs, int njobs) {
  if (threadIdx.x < static_cast<unsigned>(levels) * 8u) G.counts[threadIdx.x] = 0u;
  for (unsigned t = threadIdx.x; t < static_cast<unsigned>(levels * njobs); t += blockDim.x)
    jobs[t / levels].P.lv[t % levels].g.call[0] = 0u;
}
#endif  // !HG_BIN_STAMPS

}  // namespace hg

using namespace hg;

namespace {

// Launch shape of k_bin_apply (see there): one grid row per level; HG_APPLY_TURNS=1 lets the levels take turns
// (measured: slower, kept as a switch for the diagnostics).
dim3 apply_grid(const hg_ctx* c, int levels) {
  const bool turns = c->opt(OPT_APPLY_TURNS) != 0;
  return turns ? dim3(1024u * static_cast<unsigned>(levels), 1u) : dim3(1024u, static_cast<unsigned>(levels));
}
// Slice size of the large bins of a scan stream (PyramidIns::slice_records < 0). HG_STREAM_SLICE overrides.
int stream_slice_records(const hg_ctx* c) {
  // 1024: measured 16.4k scans/s at B = 32 against 14.9k with the single chain's 512 (768: 16.0k, 1280: 15.9k,
  // 1536: 15.7k) -- applies of consecutive scans queue behind each other, so total work counts for more than
  // the latency of the heaviest slice
  const int v = static_cast<int>(c->opt(OPT_STREAM_SLICE));
  return v > 0 ? -v : 0;
}

// Sequential decimation of Insert (:703-710): depends only on the index and the ratio.
void build_gate(double ratio, size_t n, uint8_t* out) {
  size_t inserted = 0, omitted = 0;
  for (size_t i = 0; i < n; ++i) {
    if (double(inserted) <= ratio * double(inserted + omitted)) {
      ++inserted;
      out[i] = 1;
    } else {
      ++omitted;
      out[i] = 0;
    }
  }
}

InsertParams make_params(const hg_insert_opts& o, const hg_grid* grid, bool has_pose, size_t width) {
  InsertParams p;
  p.project_normals = o.project_sdf_distance_to_scan_normal ? 1 : 0;
  p.vertical_stride = static_cast<unsigned>(std::max(0, o.normal_computation_vertical_stride));
  p.horizontal_stride = static_cast<unsigned>(std::max(0, o.normal_computation_horizontal_stride)) * static_cast<unsigned>(width);
  p.width = static_cast<unsigned>(width);
  p.min_range = o.min_range;
  p.max_range = o.max_range;
  p.truncation_distance =
      static_cast<float>(o.relative_truncation_distance * static_cast<double>(grid->view.resolution));
  p.maximum_weight = static_cast<float>(o.maximum_weight);
  p.epsilon = static_cast<float>(o.weight_function_epsilon);
  p.sigma = static_cast<float>(o.weight_function_sigma);
  p.free_space = o.num_free_space_voxels > 0 ? 1 : 0;
  p.has_pose = has_pose ? 1 : 0;
  return p;
}

// ---- compaction path (exact record count; any options), one level ---------------------------
int insert_chunk_compact(hg_grid* grid, const InsertParams& p, const ScanTable* d_scans,
                         uint32_t n_scans, const float* d_xyz, unsigned long long n,
                         const uint8_t* d_gate, bool first_chunk) {
  hg_ctx* c = grid->ctx;
  hipStream_t s = c->stream;
  int rc;
  if ((rc = c->ws_counts.reserve(sizeof(uint32_t) * (n + 1))) != HG_OK) return rc;
  if ((rc = c->ws_offsets.reserve(sizeof(unsigned long long) * (n + 1))) != HG_OK) return rc;
  uint32_t* d_counts = c->ws_counts.as<uint32_t>();
  unsigned long long* d_offsets = c->ws_offsets.as<unsigned long long>();
  const unsigned wg = 256;
  const unsigned nwg = static_cast<unsigned>((n + wg - 1) / wg);
  HG_HIP_CHECK(hipMemsetAsync(d_counts + n, 0, sizeof(uint32_t), s));
  if (first_chunk)  // per-call counters (hits, updates) restart; num_blocks and sticky flags stay
    HG_HIP_CHECK(hipMemsetAsync(grid->view.counters + 2, 0, 6 * sizeof(uint32_t), s));
  {
    ProfScope ps(c, HG_K_RAY_COUNT, n);
    hipLaunchKernelGGL(k_ray_count, dim3(nwg), dim3(wg), 0, s, grid->view, p, d_scans, n_scans,
                       d_xyz, n, d_gate, d_counts);
  }
  HG_HIP_CHECK(hipGetLastError());
  size_t temp_bytes = 0;
  HG_HIP_CHECK(rocprim::exclusive_scan(nullptr, temp_bytes, d_counts, d_offsets, 0ull, n + 1,
                                       rocprim::plus<unsigned long long>(), s));
  if ((rc = c->ws_temp.reserve(temp_bytes)) != HG_OK) return rc;
  {
    ProfScope ps(c, HG_K_SCAN, n);
    HG_HIP_CHECK(rocprim::exclusive_scan(c->ws_temp.ptr, temp_bytes, d_counts, d_offsets, 0ull, n + 1,
                                         rocprim::plus<unsigned long long>(), s));
  }
  unsigned long long total = 0;
  HG_HIP_CHECK(hipMemcpyAsync(&total, d_offsets + n, sizeof(total), hipMemcpyDeviceToHost, s));
  HG_HIP_CHECK(hipStreamSynchronize(s));
  if (total == 0) return HG_OK;
  if ((rc = c->ws_keys_a.reserve(8 * total)) != HG_OK) return rc;
  if ((rc = c->ws_keys_b.reserve(8 * total)) != HG_OK) return rc;
  if ((rc = c->ws_vals_a.reserve(8 * total)) != HG_OK) return rc;
  if ((rc = c->ws_vals_b.reserve(8 * total)) != HG_OK) return rc;
  unsigned long long* ka = c->ws_keys_a.as<unsigned long long>();
  unsigned long long* kb = c->ws_keys_b.as<unsigned long long>();
  unsigned long long* va = c->ws_vals_a.as<unsigned long long>();
  unsigned long long* vb = c->ws_vals_b.as<unsigned long long>();
  {
    ProfScope ps(c, HG_K_RAY_EXPAND, n);
    hipLaunchKernelGGL(k_ray_expand, dim3(nwg), dim3(wg), 0, s, grid->view, p, d_scans, n_scans,
                       d_xyz, n, d_gate, d_offsets, ka, va);
  }
  HG_HIP_CHECK(hipGetLastError());
  const unsigned end_bit = 43;  // 33 block + 9 voxel bits; dropped records (~0) differ in bit 42
  temp_bytes = 0;
  HG_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, temp_bytes, ka, kb, va, vb, total, 0u, end_bit, s));
  if ((rc = c->ws_temp.reserve(temp_bytes)) != HG_OK) return rc;
  {
    ProfScope ps(c, HG_K_SORT, total);
    HG_HIP_CHECK(rocprim::radix_sort_pairs(c->ws_temp.ptr, temp_bytes, ka, kb, va, vb, total, 0u,
                                           end_bit, s));
  }
  const unsigned nwg_r = static_cast<unsigned>((total + wg - 1) / wg);
  {
    ProfScope ps(c, HG_K_ALLOC, total);
    hipLaunchKernelGGL(k_alloc_blocks, dim3(nwg_r), dim3(wg), 0, s, grid->view, kb, total);
  }
  HG_HIP_CHECK(hipGetLastError());
  {
    ProfScope ps(c, HG_K_APPLY, total);
    hipLaunchKernelGGL(k_apply_runs, dim3(nwg_r), dim3(wg), 0, s, grid->view, p, kb, vb, total);
  }
  HG_HIP_CHECK(hipGetLastError());
  return HG_OK;
}

// ---- fused fixed-stride path ----------------------------------------------------------------
template <typename K, typename V>
int insert_chunk_fixed(hg_ctx* c, const PyramidIns& P, const ScanTable* d_scans, uint32_t n_scans,
                       const float* d_xyz, unsigned long long n, bool want_stats) {
  hipStream_t s = c->stream;
  const unsigned long long slots = n * static_cast<unsigned long long>(P.levels) * kSlots;
  int rc;
  if ((rc = c->ws_keys_a.reserve(sizeof(K) * slots)) != HG_OK) return rc;
  if ((rc = c->ws_keys_b.reserve(sizeof(K) * slots)) != HG_OK) return rc;
  if ((rc = c->ws_vals_a.reserve(sizeof(V) * slots)) != HG_OK) return rc;
  if ((rc = c->ws_vals_b.reserve(sizeof(V) * slots)) != HG_OK) return rc;
  const unsigned nwg_e = static_cast<unsigned>((n + 255) / 256);
  const unsigned nwg_a = static_cast<unsigned>((slots + 255) / 256);
  if ((rc = c->ws_counts.reserve(sizeof(unsigned) * (static_cast<size_t>(nwg_e) * kMaxInsLevels +
                                                     static_cast<size_t>(nwg_a) * kMaxInsLevels))) != HG_OK)
    return rc;
  unsigned* wg_hits = c->ws_counts.as<unsigned>();
  unsigned* wg_upd = wg_hits + static_cast<size_t>(nwg_e) * kMaxInsLevels;
  K* ka = c->ws_keys_a.as<K>();
  K* kb = c->ws_keys_b.as<K>();
  V* va = c->ws_vals_a.as<V>();
  V* vb = c->ws_vals_b.as<V>();
  {
    ProfScope ps(c, HG_K_RAY_EXPAND, n * P.levels);
    hipLaunchKernelGGL((k_expand_fixed<K, V>), dim3(nwg_e, P.levels), dim3(256), 0, s, P, d_scans,
                       n_scans, d_xyz, n, ka, va, wg_hits);
  }
  HG_HIP_CHECK(hipGetLastError());
  unsigned end_bit = KeyCodec<K>::kBits;
  size_t temp_bytes = 0;
  HG_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, temp_bytes, ka, kb, va, vb, slots, 0u, end_bit, s));
  if ((rc = c->ws_temp.reserve(temp_bytes)) != HG_OK) return rc;
  {
    ProfScope ps(c, HG_K_SORT, slots);
    HG_HIP_CHECK(rocprim::radix_sort_pairs(c->ws_temp.ptr, temp_bytes, ka, kb, va, vb, slots, 0u,
                                           end_bit, s));
  }
  {
    ProfScope ps(c, HG_K_ALLOC, slots);
    hipLaunchKernelGGL((k_alloc_sorted<K>), dim3(nwg_a), dim3(256), 0, s, P, kb, slots);
  }
  HG_HIP_CHECK(hipGetLastError());
  {
    ProfScope ps(c, HG_K_APPLY, slots);
    hipLaunchKernelGGL((k_apply_wave<K, V>), dim3(nwg_a), dim3(256), 0, s, P, kb, vb, slots, wg_upd);
  }
  HG_HIP_CHECK(hipGetLastError());
  if (want_stats) {
    hipLaunchKernelGGL(k_sum_stats, dim3(1), dim3(256), 0, s, P, wg_hits, nwg_e, wg_upd, nwg_a);
    HG_HIP_CHECK(hipGetLastError());
  } else {
    hipLaunchKernelGGL(k_publish_flags, dim3(1), dim3(P.levels), 0, s, P);
    HG_HIP_CHECK(hipGetLastError());
  }
  return HG_OK;
}

// Room for the deferred long chains of one apply launch over `P` (k_bin_apply, "Long chains"): per level one word
// per record the level can hold (a record is deferred at most once, so the values cannot overflow) and the lists
// of `grid_x` workgroups. `val_off` / `list_off`: this launch's share of the context's buffers, in values / entries
// (several jobs in one launch); the caller has reserved ws_heavy / ws_heavy_list. HG_NO_DEFER=1 switches the
// deferral off (every chain applied in its pass by one lane, as before round 5).
bool heavy_enabled(const hg_ctx* c) {
#ifdef HG_DEFER_LONG_CHAINS
  return c->opt(OPT_DEFER_LONG_CHAINS) != 0;
#else
  return false;
#endif
}
void attach_heavy(hg_ctx* c, PyramidIns& P, size_t records_per_level, unsigned grid_x, size_t val_off, size_t list_off) {
  for (int l = 0; l < P.levels; ++l) {
    LevelIns& L = P.lv[l];
    if (!heavy_enabled(c)) {
      L.heavy_vals = nullptr;
      L.heavy_list = nullptr;
      L.heavy_capacity = 0;
      continue;
    }
    L.heavy_vals = c->ws_heavy.as<uint32_t>() + val_off + records_per_level * l;
    L.heavy_list = c->ws_heavy_list.as<uint4>() + list_off + static_cast<size_t>(grid_x) * kHeavyPerWg * l;
    L.heavy_capacity = static_cast<uint32_t>(records_per_level);
  }
}

// ---- binned path (single scan, unit weight) ------------------------------------------------
// `pipe` >= 0: chunk number of a pipelined scan stream (several binned chunks in one call). The front
// end (count, offsets, scatter) of chunk k runs on the context's stream, its apply pass on the apply
// stream: with known poses the front end of scan k + 1 does not depend on the apply pass of scan k,
// which is a few long per-voxel chains on an otherwise idle chip. Records, work list and work
// counters are double buffered (chunk parity); applies stay in scan order on their stream; the caller
// makes the context's stream wait for the last apply before it returns. pipe < 0: everything on the
// context's stream.
int insert_chunk_binned(hg_ctx* c, const PyramidIns& P_in, const ScanTable* d_scans, uint32_t n_scans,
                        const float* d_xyz, unsigned long long n, bool want_stats, int pipe = -1) {
  hipStream_t s = c->stream;
  const unsigned records_per_level = static_cast<unsigned>(n) * kSlots;
  PyramidIns P = P_in;
  const int parity = pipe >= 0 ? (pipe & 1) : 0;
  for (int l = 0; l < P.levels; ++l) P.lv[l].g.call = P.lv[l].g.counters + 16 + 4 * parity;
  DeviceBuffer& buf_work = parity ? c->ws_offsets_b : c->ws_offsets;
  DeviceBuffer& buf_keys = parity ? c->ws_keys_c : c->ws_keys_a;
  DeviceBuffer& buf_vals = parity ? c->ws_vals_c : c->ws_vals_a;
  hipStream_t sa = s;
  if (pipe >= 0) {
    int prc = ensure_apply_stream(c);
    if (prc != HG_OK) return prc;
    sa = c->apply_stream;
    // the buffers of this parity were last read by the apply pass of chunk pipe - 2
    if (pipe >= 2) HG_HIP_CHECK(hipStreamWaitEvent(s, c->ev_apply[parity], 0));
  }
  const size_t slots = static_cast<size_t>(records_per_level) * P.levels;
  int rc;
  // Apply work list, per level: one item per touched bin (a return touches at most kMaxRuns blocks)
  // plus the extra slices of large bins -- a bin of cnt > 512 records is cut into S <= cnt / 256
  // slices, so all bins together add fewer than records / 256 items. Sized from the records of this
  // call, not from the grid's block pool.
  {
    size_t max_pool = 0;
    for (int l = 0; l < P.levels; ++l) max_pool = std::max<size_t>(max_pool, P.lv[l].g.max_blocks);
    const size_t per_level = std::min<size_t>(static_cast<size_t>(n) * kMaxRuns, max_pool) +
                             records_per_level / 256u + 64u;
    if ((rc = buf_work.reserve(sizeof(uint4) * per_level * P.levels)) != HG_OK) return rc;
    for (int l = 0; l < P.levels; ++l) {
      P.lv[l].g.work = buf_work.as<uint4>() + per_level * l;
      P.lv[l].g.work_capacity = static_cast<uint32_t>(per_level);
    }
  }
  if ((rc = buf_keys.reserve(sizeof(uint32_t) * slots)) != HG_OK) return rc;
  if ((rc = buf_vals.reserve(sizeof(uint32_t) * slots)) != HG_OK) return rc;
  // (one buffer for both parities: the apply passes of a pipelined stream run one after the other on their stream)
  if (heavy_enabled(c)) {
    if ((rc = c->ws_heavy.reserve(sizeof(uint32_t) * slots)) != HG_OK) return rc;
    if ((rc = c->ws_heavy_list.reserve(sizeof(uint4) * 1024u * kHeavyPerWg * P.levels)) != HG_OK) return rc;
  }
  attach_heavy(c, P, records_per_level, 1024u, 0, 0);
  const unsigned nwg_e = static_cast<unsigned>((n + 255) / 256);
  if ((rc = c->ws_counts.reserve(sizeof(unsigned) * static_cast<size_t>(nwg_e) * kMaxInsLevels)) != HG_OK) return rc;
  if ((rc = c->ws_keys_b.reserve(sizeof(RunInfo) * run_info_units(n, nwg_e, P.levels))) != HG_OK) return rc;
  unsigned* wg_hits = c->ws_counts.as<unsigned>();
  uint32_t* rk = buf_keys.as<uint32_t>();
  uint32_t* rv = buf_vals.as<uint32_t>();
  RunInfo* runs = c->ws_keys_b.as<RunInfo>();
  {
    ProfScope ps(c, HG_K_RAY_COUNT, n * P.levels);
    hipLaunchKernelGGL(k_bin_count, dim3(nwg_e, P.levels), dim3(256), 0, s, P, d_scans, n_scans, d_xyz,
                       static_cast<unsigned>(n), runs, wg_hits);
  }
  HG_HIP_CHECK(hipGetLastError());
  {
    ProfScope ps(c, HG_K_SCAN, P.levels);
    hipLaunchKernelGGL(k_bin_offsets, dim3(P.levels), dim3(1024), 0, s, P, records_per_level);
  }
  HG_HIP_CHECK(hipGetLastError());
  {
    ProfScope ps(c, HG_K_RAY_EXPAND, n * P.levels);
    hipLaunchKernelGGL(k_bin_scatter, dim3(nwg_e, P.levels), dim3(256), 0, s, P, d_scans, n_scans, d_xyz,
                       static_cast<unsigned>(n), runs, rk, rv);
  }
  HG_HIP_CHECK(hipGetLastError());
  if (pipe >= 0) {
    if (want_stats) {  // wg_hits is reused by the next chunk's front end: sum on the front stream
      hipLaunchKernelGGL(k_sum_stats, dim3(1), dim3(256), 0, s, P, wg_hits, nwg_e, nullptr, 0u);
      HG_HIP_CHECK(hipGetLastError());
    }
    HG_HIP_CHECK(hipEventRecord(c->ev_front[parity], s));
    HG_HIP_CHECK(hipStreamWaitEvent(sa, c->ev_front[parity], 0));
  }
  {
    ProfScope ps(c, HG_K_APPLY, slots, 1, true, sa);
#ifdef HG_BIN_STAMPS
    static long long* d_st = nullptr;
    if (!d_st) hipMalloc(reinterpret_cast<void**>(&d_st), 3 * 4096 * 8 * sizeof(long long));
    hipMemsetAsync(d_st, 0, 3 * 4096 * 8 * sizeof(long long), s);
    hipLaunchKernelGGL(k_bin_apply, apply_grid(c, P.levels), dim3(kBinThreads), 0, s, P, rk, rv, d_st);
    {
      std::vector<long long> h(3 * 4096 * 8);
      hipMemcpy(h.data(), d_st, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
      long long t0 = -1;
      for (size_t i = 0; i < h.size(); i += 8) if (h[i] && (t0 < 0 || h[i] < t0)) t0 = h[i];
      for (int y = 0; y < 3; ++y) {
        // summary: count, mean phase durations, and the latest finishing items
        double sum[5] = {0, 0, 0, 0, 0};
        int cnt = 0;
        long long last_end = 0; int last_i = -1;
        for (int w = 0; w < 3072; ++w) {
          const long long* e = &h[(static_cast<size_t>(y) * 4096 + w) * 8];
          if (!e[0]) continue;
          ++cnt;
          for (int k = 1; k <= 4; ++k) if (e[k]) sum[k] += double(e[k] - e[k - 1]);
          long long end = e[4] ? e[4] : e[1];
          if (end > last_end) { last_end = end; last_i = w; }
        }
        {
          long long first = -1, lastend = 0, longest = 0; int li = -1;
          for (int w = 0; w < 3072; ++w) {
            const long long* e = &h[(static_cast<size_t>(y) * 4096 + w) * 8];
            if (!e[0] || !e[5]) continue;
            if (first < 0 || e[0] < first) first = e[0];
            if (e[5] > lastend) lastend = e[5];
            if (e[5] - e[0] > longest) { longest = e[5] - e[0]; li = w; }
          }
          const long long* e = &h[(static_cast<size_t>(y) * 4096 + (li < 0 ? 0 : li)) * 8];
          fprintf(stderr, "bin y=%d span %lld..%lld (x10ns from kernel start); longest item %d: %lld (n=%lld) starts %lld hist=%lld last-pass[group=%lld rank=%lld chain=%lld]\n", y,
                  first - t0, lastend - t0, li, longest, e[6], e[0] - t0, e[1] - e[0], e[2] - e[1], e[3] - e[2], e[4] - e[3]);
          // the 5 items that end last
          for (int rep = 0; rep < 5; ++rep) {
            long long best = 0; int bi = -1;
            for (int w = 0; w < 3072; ++w) {
              const long long* q = &h[(static_cast<size_t>(y) * 4096 + w) * 8];
              if (q[0] && q[5] > best && q[5] < (rep ? lastend : lastend + 1)) { best = q[5]; bi = w; }
            }
            if (bi < 0) break;
            const long long* q = &h[(static_cast<size_t>(y) * 4096 + bi) * 8];
            fprintf(stderr, "   late item %d: start %lld end %lld n=%lld\n", bi, q[0] - t0, q[5] - t0, q[6]);
            lastend = best;
          }
        }
        {
          // deferred long chains: [7] tail start, [4] tail end (overwrites the item's chain stamp), [6] entries * 1e6 + values
          int tails = 0; double tsum = 0; long long tmax = 0, tend = 0; long long tmax_info = 0;
          const long long* tph = nullptr;
          for (int w = 0; w < 3072; ++w) {
            const long long* q = &h[(static_cast<size_t>(y) * 4096 + w) * 8];
            if (!q[7]) continue;
            ++tails; tsum += double(q[4] - q[7]);
            if (q[4] - q[7] > tmax) { tmax = q[4] - q[7]; tmax_info = q[6]; tph = &h[(static_cast<size_t>(y) * 4096 + 3072 + (w & 1023)) * 8]; }
            if (q[4] > tend) tend = q[4];
          }
          if (tails) fprintf(stderr, "   tails y=%d: %d workgroups, mean %.0f, longest %lld (entries %lld values %lld), last tail ends at %lld; longest's first group: entries %lld copy %lld seg %lld\n", y, tails,
                             tsum / tails, tmax, tmax_info / 1000000ll, tmax_info % 1000000ll, tend - t0, tph ? tph[1] - tph[0] : 0, tph ? tph[2] - tph[1] : 0, tph ? tph[3] - tph[2] : 0);
        }
#ifdef HG_SEG_STATS
        if (y == 0) {
          unsigned st[8] = {0}, zero[8] = {0};
          hipMemcpyFromSymbol(st, HIP_SYMBOL(hg::g_seg_stats), sizeof(st));
          hipMemcpyToSymbol(HIP_SYMBOL(hg::g_seg_stats), zero, sizeof(zero));
          fprintf(stderr, "   seg stats: lookups %u misses %u failed weight checks %u max |c - p| %u mean %.2f\n", st[0], st[1], st[2], st[3],
                  st[0] ? double(st[4]) / st[0] : 0.0);
          long long sp[8]; unsigned long long z = 0;
          hipMemcpyFromSymbol(sp, HIP_SYMBOL(hg::g_seg_stamps), sizeof(sp));
          hipMemcpyToSymbol(HIP_SYMBOL(hg::g_seg_longest), &z, sizeof(z));
          fprintf(stderr, "   longest seg_chains call (x10ns): assign %lld affine+barrier %lld predict+chain(wave 0) %lld barrier %lld walk %lld; first voxel %lld values, %lld voxels\n",
                  sp[1] - sp[0], sp[2] - sp[1], sp[3] - sp[2], sp[4] - sp[3], sp[5] - sp[4], sp[6], sp[7]);
        }
#endif
        fprintf(stderr, "bin y=%d items=%d mean cycles hist=%.0f group=%.0f rank=%.0f chain=%.0f; last item %d ends at %lld",
                y, cnt, cnt ? sum[1] / cnt : 0, cnt ? sum[2] / cnt : 0, cnt ? sum[3] / cnt : 0, cnt ? sum[4] / cnt : 0, last_i, last_end - t0);
        if (last_i >= 0) {
          const long long* e = &h[(static_cast<size_t>(y) * 4096 + last_i) * 8];
          fprintf(stderr, " [n=%lld start=%lld hist=%lld group=%lld rank=%lld chain=%lld]", e[6], e[0] - t0, e[1] - e[0], e[2] - e[1], e[3] - e[2], e[4] - e[3]);
        }
        fprintf(stderr, "\n");
      }
    }
#else
    hipLaunchKernelGGL(k_bin_apply, apply_grid(c, P.levels), dim3(kBinThreads), 0, sa, P, rk, rv);
#endif
  }
  HG_HIP_CHECK(hipGetLastError());
  if (pipe >= 0) {
    HG_HIP_CHECK(hipEventRecord(c->ev_apply[parity], sa));
    return HG_OK;
  }
  if (want_stats) {
    // hits from the per-workgroup counts; updates were written by k_bin_offsets
    hipLaunchKernelGGL(k_sum_stats, dim3(1), dim3(256), 0, s, P, wg_hits, nwg_e, nullptr, 0u);
    HG_HIP_CHECK(hipGetLastError());
  }
  return HG_OK;
}

// ---- tolerance path (HG_INSERT_FAST, unit weight) -------------------------------------------
// Tolerance path on the bins (k_fast_offsets ... k_fast_bin_apply). One chunk = up to 2^20 - 1 returns of one or
// several scans: the scans of a chunk share their bins, so a voxel's updates of the whole chunk are applied once.
int insert_chunk_fast(hg_ctx* c, const PyramidIns& P_in, const ScanTable* d_scans, uint32_t n_scans,
                      const float* d_xyz, unsigned long long n, bool want_stats) {
  hipStream_t s = c->stream;
  const unsigned records_per_level = static_cast<unsigned>(n) * kSlots;
  PyramidIns P = P_in;
  for (int l = 0; l < P.levels; ++l) P.lv[l].g.call = P.lv[l].g.counters + 16;
  int rc;
  {
    size_t max_pool = 0;
    for (int l = 0; l < P.levels; ++l) max_pool = std::max<size_t>(max_pool, P.lv[l].g.max_blocks);
    // one item per touched bin plus the extra parts of cut bins (fewer than records / kFastItemRecords)
    const size_t per_level = std::min<size_t>(static_cast<size_t>(n) * kMaxRuns, max_pool) +
                             records_per_level / kFastItemRecords + 64u;
    if ((rc = c->ws_offsets.reserve(sizeof(uint4) * per_level * P.levels)) != HG_OK) return rc;
    for (int l = 0; l < P.levels; ++l) {
      P.lv[l].g.work = c->ws_offsets.as<uint4>() + per_level * l;
      P.lv[l].g.work_capacity = static_cast<uint32_t>(per_level);
    }
  }
  // a cut bin of cnt records has ceil(cnt / kFastItemRecords) <= cnt / kFastItemRecords + 1 < 2 cnt / kFastItemRecords parts
  FastScratch fs;
  fs.tile_capacity = 2u * (records_per_level / kFastItemRecords) + 64u;
  if ((rc = c->ws_vals_a.reserve(sizeof(unsigned long long) * kVoxelsPerBlock * fs.tile_capacity * P.levels)) != HG_OK) return rc;
  fs.tiles = c->ws_vals_a.as<unsigned long long>();
  const size_t slots = static_cast<size_t>(records_per_level) * P.levels;
  if ((rc = c->ws_keys_a.reserve(sizeof(uint32_t) * slots)) != HG_OK) return rc;
  const unsigned nwg_e = static_cast<unsigned>((n + 255) / 256);
  if ((rc = c->ws_counts.reserve(sizeof(unsigned) * static_cast<size_t>(nwg_e) * kMaxInsLevels)) != HG_OK) return rc;
  if ((rc = c->ws_keys_b.reserve(sizeof(RunInfo) * run_info_units(n, nwg_e, P.levels))) != HG_OK) return rc;
  unsigned* wg_hits = c->ws_counts.as<unsigned>();
  uint32_t* recs = c->ws_keys_a.as<uint32_t>();
  RunInfo* runs = c->ws_keys_b.as<RunInfo>();
  {
    ProfScope ps(c, HG_K_RAY_COUNT, n * P.levels);
    hipLaunchKernelGGL(k_bin_count, dim3(nwg_e, P.levels), dim3(256), 0, s, P, d_scans, n_scans, d_xyz,
                       static_cast<unsigned>(n), runs, wg_hits);
  }
  {
    ProfScope ps(c, HG_K_SCAN, P.levels);
    hipLaunchKernelGGL(k_fast_offsets, dim3(P.levels), dim3(1024), 0, s, P, records_per_level, fs);
  }
  {
    ProfScope ps(c, HG_K_RAY_EXPAND, n * P.levels);
    hipLaunchKernelGGL(k_fast_scatter, dim3(nwg_e, P.levels), dim3(256), 0, s, P, d_scans, n_scans, d_xyz,
                       static_cast<unsigned>(n), runs, recs);
  }
  {
    ProfScope ps(c, HG_K_APPLY, slots);
    hipLaunchKernelGGL(k_fast_bin_apply, dim3(1024, P.levels), dim3(kFastApplyThreads), 0, s, P, recs, fs);
  }
  HG_HIP_CHECK(hipGetLastError());
  if (want_stats) {
    // hits from the per-workgroup counts; updates were written by k_fast_offsets
    hipLaunchKernelGGL(k_sum_stats, dim3(1), dim3(256), 0, s, P, wg_hits, nwg_e, nullptr, 0u);
    HG_HIP_CHECK(hipGetLastError());
  }
  return HG_OK;
}

}  // namespace
// Sticky error flags of a grid -> status code + message.
int hg::async_status_grids(hg_grid* const* grids, int count) {
  uint32_t f = 0;
  for (int i = 0; i < count; ++i)
    if (grids[i] && grids[i]->ctx->flag_words) f |= grids[i]->ctx->flag_words[grids[i]->flag_slot];
  return f ? flags_to_status(f) : HG_OK;
}

int hg::flags_to_status(uint32_t flags) {
  if (flags & kFlagCapacity) {
    set_last_error("block pool exhausted: raise max_blocks");
    return HG_ERR_CAPACITY;
  }
  if (flags & kFlagRange) {
    set_last_error("cell index outside +-8192 (or outside the 32-bit key window)");
    return HG_ERR_RANGE;
  }
  if (flags & kFlagStride) {
    set_last_error("ray produced more than 8 samples on the fixed-stride path");
    return HG_ERR_UNSUPPORTED;
  }
  if (flags & kFlagBinOverflow) {
    set_last_error("a voxel received more updates in one scan than the binned path supports");
    return HG_ERR_UNSUPPORTED;
  }
  if (flags & kFlagWorkOverflow) {
    set_last_error("apply work list overflow (internal sizing error)");
    return HG_ERR_CAPACITY;
  }
  if (flags & kFlagTime) {
    set_last_error("per-point unwarping: a return's time lies outside the control points");
    return HG_ERR_TIME;
  }
  return HG_OK;
}
namespace {

int read_stats(hg_grid* grid, hg_insert_stats* out) {
  uint32_t cnt[8];
  hipStream_t s = grid->ctx->stream;
  HG_HIP_CHECK(hipMemcpyAsync(cnt, grid->view.counters, sizeof(cnt), hipMemcpyDeviceToHost, s));
  HG_HIP_CHECK(hipStreamSynchronize(s));
  hg_insert_stats st;
  st.num_hits = cnt[2];
  st.num_updates = static_cast<uint64_t>(cnt[4]) | (static_cast<uint64_t>(cnt[5]) << 32);
  st.num_blocks = std::min(cnt[0], grid->view.max_blocks);
  st.flags = cnt[1];
  grid->last_stats = st;
  if (out) *out = st;
  return flags_to_status(cnt[1]);
}

void host_transform(const float* pose, const float* in, float* out) {
  const float qw = pose[3], qx = pose[4], qy = pose[5], qz = pose[6];
  const float x = in[0], y = in[1], z = in[2];
  float ux = qy * z - qz * y, uy = qz * x - qx * z, uz = qx * y - qy * x;
  ux += ux; uy += uy; uz += uz;
  out[0] = x + qw * ux + (qy * uz - qz * uy) + pose[0];
  out[1] = y + qw * uy + (qz * ux - qx * uz) + pose[1];
  out[2] = z + qw * uz + (qx * uy - qy * ux) + pose[2];
}

}  // namespace

// Exact insertion of one scan into each of `count` pyramids with shared launches (the insertion half
// of hg_register_scan_batch): job j inserts xyz[j] (n[j] returns, device memory, tracking frame) into
// grids[j * levels .. ] at the fp64 pose d_poses[j] (device memory; cast to float on the device as
// Rigid3d::cast<float>() does). Binned path only: HG_ERR_UNSUPPORTED for options outside it (the
// caller then inserts pyramid by pyramid). No stats read-back: errors reach the host through the
// context's mailbox (async_status).
int hg::pyramid_insert_jobs(hg_ctx* c, int count, hg_grid* const* grids, const hg_insert_opts* opts, int levels,
                            const float* origins, const float* const* xyz, const size_t* n, size_t width,
                            const double* const* d_poses) {
  if (!c || count < 1 || !grids || !opts || levels < 1 || levels > kMaxInsLevels || !origins || !xyz || !n || !d_poses)
    return HG_ERR_INVALID;
  for (int l = 0; l < levels; ++l) {
    if (opts[l].num_free_space_voxels > 0 || !(opts[l].relative_truncation_distance <= 3.0) ||
        !(static_cast<float>(opts[l].weight_function_epsilon) >= 1.0f) || opts[l].insertion_ratio < 1.0)
      return HG_ERR_UNSUPPORTED;
    if (opts[l].project_sdf_distance_to_scan_normal && opts[l].normal_computation_method != 1) return HG_ERR_UNSUPPORTED;
  }
  hipStream_t s = c->stream;
  HG_HIP_CHECK(hipSetDevice(c->device));
  // pinned staging of the job table (the previous batch's copy has completed: its poses were fetched)
  const size_t table_bytes = static_cast<size_t>(count) * sizeof(InsertJob);
  if (c->ijobs_capacity < table_bytes) {
    if (c->pinned_ijobs) (void)hipHostFree(c->pinned_ijobs);
    c->pinned_ijobs = nullptr;
    c->ijobs_capacity = 0;
    const size_t cap = std::max<size_t>(16, count) * sizeof(InsertJob);
    HG_HIP_CHECK(hipHostMalloc(&c->pinned_ijobs, cap));
    c->ijobs_capacity = cap;
  }
  InsertJob* jobs = static_cast<InsertJob*>(c->pinned_ijobs);
  // workspace layout
  size_t rec_words = 0, run_items = 0, hit_words = 0, work_items = 0;
  unsigned max_nwg = 0;
  for (int j = 0; j < count; ++j) {
    if (n[j] == 0 || n[j] >= (1ull << 20) || !xyz[j] || !d_poses[j]) return HG_ERR_UNSUPPORTED;
    const unsigned nj = static_cast<unsigned>(n[j]);
    const unsigned nwg = (nj + 255u) / 256u;
    max_nwg = std::max(max_nwg, nwg);
    rec_words += static_cast<size_t>(nj) * kSlots * levels;
    run_items += run_info_units(nj, nwg, levels);
    hit_words += static_cast<size_t>(nwg) * kMaxInsLevels;
    size_t max_pool = 0;
    for (int l = 0; l < levels; ++l) {
      hg_grid* g = grids[j * levels + l];
      if (!g || g->ctx != c) return HG_ERR_INVALID;
      max_pool = std::max<size_t>(max_pool, g->view.max_blocks);
    }
    work_items += (std::min<size_t>(static_cast<size_t>(nj) * kMaxRuns, max_pool) + static_cast<size_t>(nj) * kSlots / 256u + 64u) * levels;
  }
  int rc;
  if ((rc = c->ws_keys_a.reserve(sizeof(uint32_t) * rec_words)) != HG_OK) return rc;
  if ((rc = c->ws_vals_a.reserve(sizeof(uint32_t) * rec_words)) != HG_OK) return rc;
  if ((rc = c->ws_keys_b.reserve(sizeof(RunInfo) * run_items)) != HG_OK) return rc;
  if ((rc = c->ws_counts.reserve(sizeof(unsigned) * hit_words)) != HG_OK) return rc;
  if ((rc = c->ws_offsets.reserve(sizeof(uint4) * work_items)) != HG_OK) return rc;
  if ((rc = c->ws_jobs.reserve(table_bytes)) != HG_OK) return rc;
  const unsigned apply_gx = count <= 2 ? 1024u : 512u;
  if (heavy_enabled(c)) {
    if ((rc = c->ws_heavy.reserve(sizeof(uint32_t) * rec_words)) != HG_OK) return rc;
    if ((rc = c->ws_heavy_list.reserve(sizeof(uint4) * apply_gx * kHeavyPerWg * levels * count)) != HG_OK) return rc;
  }
  size_t rec_off = 0, run_off = 0, hit_off = 0, work_off = 0;
  for (int j = 0; j < count; ++j) {
    InsertJob& J = jobs[j];
    std::memset(&J, 0, sizeof(J));
    const unsigned nj = static_cast<unsigned>(n[j]);
    J.n = nj;
    J.nwg = (nj + 255u) / 256u;
    J.records_per_level = nj * kSlots;
    J.xyz = xyz[j];
    J.rec_keys = c->ws_keys_a.as<uint32_t>() + rec_off;
    J.rec_vals = c->ws_vals_a.as<uint32_t>() + rec_off;
    J.runs = c->ws_keys_b.as<RunInfo>() + run_off;
    J.wg_hits = c->ws_counts.as<unsigned>() + hit_off;
    const size_t heavy_off = rec_off;
    rec_off += static_cast<size_t>(nj) * kSlots * levels;
    run_off += run_info_units(nj, J.nwg, levels);
    hit_off += static_cast<size_t>(J.nwg) * kMaxInsLevels;
    PyramidIns& P = J.P;
    P.levels = levels;
    P.d_pose = d_poses[j];
    P.accumulate = 0;
    P.host_flags = c->async_flags;
    P.slice_records = count >= 4 ? 2048 : 0;
    P.shared = 0;
    P.scan0.begin = 0;
    P.scan0.count = nj;
    std::memcpy(P.scan0.origin, origins + 3 * j, sizeof(P.scan0.origin));
    size_t max_pool = 0;
    for (int l = 0; l < levels; ++l) max_pool = std::max<size_t>(max_pool, grids[j * levels + l]->view.max_blocks);
    const size_t per_level = std::min<size_t>(static_cast<size_t>(nj) * kMaxRuns, max_pool) + static_cast<size_t>(nj) * kSlots / 256u + 64u;
    for (int l = 0; l < levels; ++l) {
      LevelIns& L = P.lv[l];
      L.g = grids[j * levels + l]->view;
      P.flag_slot[l] = static_cast<uint16_t>(grids[j * levels + l]->flag_slot);
      L.p = make_params(opts[l], grids[j * levels + l], true, width);
      L.gate = nullptr;
      L.g.work = c->ws_offsets.as<uint4>() + work_off + per_level * l;
      L.g.work_capacity = static_cast<uint32_t>(per_level);
    }
    attach_heavy(c, P, static_cast<size_t>(nj) * kSlots, apply_gx, heavy_off, static_cast<size_t>(apply_gx) * kHeavyPerWg * levels * j);
    work_off += per_level * levels;
  }
  const InsertJob* d_jobs = c->ws_jobs.as<InsertJob>();
  HG_HIP_CHECK(hipMemcpyAsync(c->ws_jobs.ptr, jobs, table_bytes, hipMemcpyHostToDevice, s));
  unsigned long long units = 0;
  for (int j = 0; j < count; ++j) units += n[j] * static_cast<unsigned long long>(levels);
  {
    ProfScope ps(c, HG_K_RAY_COUNT, units);
    hipLaunchKernelGGL(k_bin_count_jobs, dim3(max_nwg, count * levels), dim3(256), 0, s, d_jobs, levels);
  }
  {
    ProfScope ps(c, HG_K_SCAN, count * levels);
    hipLaunchKernelGGL(k_bin_offsets_jobs, dim3(count * levels), dim3(1024), 0, s, d_jobs, levels);
  }
  {
    ProfScope ps(c, HG_K_RAY_EXPAND, units);
    hipLaunchKernelGGL(k_bin_scatter_jobs, dim3(max_nwg, count * levels), dim3(256), 0, s, d_jobs, levels);
  }
  {
    ProfScope ps(c, HG_K_APPLY, units * kSlots);
#ifndef HG_BIN_STAMPS
    hipLaunchKernelGGL(k_bin_apply_jobs, dim3(apply_gx, levels * count), dim3(kBinThreads), 0, s, d_jobs, count);
#endif
    if (count >= 4)  // as P.slice_records: throughput mode
      hipLaunchKernelGGL(k_bin_apply_small_jobs, dim3(512, count * levels), dim3(kSmallThreads), 0, s, d_jobs, levels);
  }
  HG_HIP_CHECK(hipGetLastError());
  return HG_OK;
}

namespace {

// A stream of scans with known poses into ONE pyramid (hg_pyramid_insert_batch, exact binned path).
// Scan after scan, the four insert kernels are each a short dependent chain that leaves most of the
// chip idle. Here the FRONT ENDS (count, offsets, scatter) of a group of scans share their launches
// (k_bin_*_jobs): they do not depend on the map's voxels, only the apply passes do. Every scan of a
// group has its own bin arrays, touched list, work list and records, so the per-scan bins -- and with
// them every voxel's update order -- are exactly those of scan-by-scan insertion. The apply passes
// then run one launch per scan, in scan order. Per 100k-point scan the front end drops from 72 to
// 29 us (groups of 8). Running the next group's front end NEXT TO the apply passes (second stream,
// also with the front end confined to half of the CUs by a CU mask) was measured and dropped: the
// apply pass slows down 2.5x under the front end's atomics and scattered writes, no net gain.
// Pinned staging of a call's table (job table of a grouped stream, scan table of a multi-scan call): the previous
// call's copy must have left it; the caller records ev_sjobs behind its own copy (stage_commit).
int stage_table(hg_ctx* c, size_t table_bytes, void** out) {
  if (c->sjobs_pending) {
    HG_HIP_CHECK(hipEventSynchronize(c->ev_sjobs));  // the previous call's table copy has left the staging
    c->sjobs_pending = false;
  }
  if (!c->ev_sjobs) HG_HIP_CHECK(hipEventCreateWithFlags(&c->ev_sjobs, hipEventDisableTiming));
  if (c->sjobs_capacity < table_bytes) {
    if (c->pinned_sjobs) (void)hipHostFree(c->pinned_sjobs);
    c->pinned_sjobs = nullptr;
    c->sjobs_capacity = 0;
    HG_HIP_CHECK(hipHostMalloc(&c->pinned_sjobs, table_bytes + table_bytes / 2));
    c->sjobs_capacity = table_bytes + table_bytes / 2;
  }
  *out = c->pinned_sjobs;
  return HG_OK;
}
int stage_commit(hg_ctx* c) {
  HG_HIP_CHECK(hipEventRecord(c->ev_sjobs, c->stream));
  c->sjobs_pending = true;
  return HG_OK;
}

int insert_stream_grouped(hg_ctx* c, const PyramidIns& P0, const float* origins, const float* d_xyz,
                          const uint64_t* scan_offsets, size_t n_scans, const float* poses_tq,
                          bool want_stats) {
  hipStream_t s = c->stream;
  const int levels = P0.levels;
  const int group = std::max(1, std::min(32, static_cast<int>(c->opt(OPT_STREAM_GROUP))));
  std::vector<size_t> scans;  // the non-empty scans
  for (size_t i = 0; i < n_scans; ++i)
    if (scan_offsets[i + 1] > scan_offsets[i]) scans.push_back(i);
  const int count = static_cast<int>(scans.size());
  if (count == 0) return HG_OK;
  int rc;
  // job table staging
  const size_t table_bytes = static_cast<size_t>(count) * sizeof(InsertJob);
  void* staged = nullptr;
  if ((rc = stage_table(c, table_bytes, &staged)) != HG_OK) return rc;
  InsertJob* jobs = static_cast<InsertJob*>(staged);
  // workspace sizes: the largest group decides
  size_t max_pool = 0, max_blocks = 0;
  for (int l = 0; l < levels; ++l) {
    max_pool = std::max<size_t>(max_pool, P0.lv[l].g.pool_blocks);
    max_blocks = std::max<size_t>(max_blocks, P0.lv[l].g.max_blocks);
  }
  size_t rec_words = 0, run_items = 0, hit_words = 0, work_items = 0;
  unsigned n_max = 0;
  for (int g0 = 0; g0 < count; g0 += group) {
    size_t rw = 0, ri = 0, hw = 0, wi = 0;
    for (int j = g0; j < std::min(count, g0 + group); ++j) {
      const unsigned long long nj = scan_offsets[scans[j] + 1] - scan_offsets[scans[j]];
      n_max = std::max(n_max, static_cast<unsigned>(nj));
      rw += nj * kSlots * levels;
      ri += run_info_units(nj, (nj + 255u) / 256u, levels);
      hw += ((nj + 255u) / 256u) * kMaxInsLevels;
      wi += (std::min<size_t>(nj * kMaxRuns, max_blocks) + nj * kSlots / 256u + 64u) * levels;
    }
    rec_words = std::max(rec_words, rw);
    run_items = std::max(run_items, ri);
    hit_words = std::max(hit_words, hw);
    work_items = std::max(work_items, wi);
  }
  const size_t touched_cap = std::min<size_t>(static_cast<size_t>(n_max) * kMaxRuns, max_blocks) + 64u;
  // shadow layout: [group][levels] call counters (4 words), then per (job slot, level)
  // bin_count[pool], bin_offset[pool], touched[touched_cap]; merged apply: + per level claim[pool], + 8 counter words per
  // level, + per level union[pool]
  // stream_merge (default 1): ONE apply launch per group of scans (k_stream_units / k_bin_apply_stream) instead of one per
  // scan. Measured on one box, interleaved (scripts/r06_stream_ab.sh, groups of 8): 32 scans of one room 18.3k -> 18.8k
  // scans/s, 32 scans over 32 rooms 16.4k -> 22.1k, 64 over 64 rooms 9.2k -> 17.7k, 500 over 400 rooms (2 GB of voxels)
  // 10.2k -> 18.8k: where the scans of a group touch different blocks their launches now run side by side. Groups of
  // 16 / 32 (the default since): one room 19.2k / 20.7k, 64 rooms 19.3k / 19.9k. What is left in one room is the unit of
  // the busiest (block, slice): its voxels' updates of all scans of the group in one serial chain.
#ifdef HG_BIN_STAMPS
  const bool merged = false;  // (the stamp build times k_bin_apply's items: a launch per scan)
#else
  const bool merged = group >= 2 && group <= kStreamGroupMax && !heavy_enabled(c) && c->opt(OPT_STREAM_MERGE) != 0;
#endif
  std::vector<char> group_merged((count + group - 1) / group, merged ? 1 : 0);
  const size_t call_words = static_cast<size_t>(group) * levels * 4u;
  const size_t per_slot_level = 2u * max_pool + touched_cap;
  const size_t claim_words = merged ? 2u * max_pool * levels + 8u * kMaxInsLevels : 0u;  // claim words, counters, union lists
  const size_t shadow_words = call_words + per_slot_level * levels * group + claim_words;
  // merged apply: unit tables and items of a group, per level (k_stream_units)
  size_t unit_cap = 0, item_cap = 0;
  if (merged) {
    size_t group_records = 0, group_runs = 0;
    for (int g0 = 0; g0 < count; g0 += group) {
      size_t r = 0, u = 0;
      for (int j = g0; j < std::min(count, g0 + group); ++j) {
        const unsigned long long nj = scan_offsets[scans[j] + 1] - scan_offsets[scans[j]];
        r += nj * kSlots;
        u += nj * kMaxRuns;
      }
      group_records = std::max(group_records, r);
      group_runs = std::max(group_runs, u);
    }
    unit_cap = std::min<size_t>(group_runs, max_blocks) + 2u * (group_records / 1024u) + 64u;
    item_cap = unit_cap * static_cast<size_t>(group);
    // [levels][2 unit tables of unit_cap uint2] behind [levels][item_cap uint4]
    work_items = std::max(work_items, (item_cap + ((kUnitTiers + 1u) * unit_cap * sizeof(uint2) + sizeof(uint4) - 1u) / sizeof(uint4)) * levels);
  }
  if ((rc = c->ws_keys_a.reserve(sizeof(uint32_t) * rec_words)) != HG_OK) return rc;
  if ((rc = c->ws_vals_a.reserve(sizeof(uint32_t) * rec_words)) != HG_OK) return rc;
  if ((rc = c->ws_offsets.reserve(sizeof(uint4) * work_items)) != HG_OK) return rc;
  if ((rc = c->ws_keys_b.reserve(sizeof(RunInfo) * run_items)) != HG_OK) return rc;
  if ((rc = c->ws_counts.reserve(sizeof(unsigned) * hit_words)) != HG_OK) return rc;
  if ((rc = c->ws_sjobs.reserve(table_bytes)) != HG_OK) return rc;
  if ((rc = c->ws_shadow.reserve(sizeof(uint32_t) * shadow_words)) != HG_OK) return rc;
  // (the apply passes run one after the other: they share one set of deferral buffers)
  if (heavy_enabled(c)) {
    if ((rc = c->ws_heavy.reserve(sizeof(uint32_t) * static_cast<size_t>(n_max) * kSlots * levels)) != HG_OK) return rc;
    if ((rc = c->ws_heavy_list.reserve(sizeof(uint4) * 1024u * kHeavyPerWg * levels)) != HG_OK) return rc;
  }
  {
    // bin counts and call counters are all-zero between calls of ONE layout (k_bin_offsets restores
    // that); bin offsets and touched lists keep their last values, so a call that lays the buffer out
    // differently (other scan sizes, pyramid, pool, group) must start from zeroes again
    const unsigned long long layout[7] = {reinterpret_cast<unsigned long long>(c->ws_shadow.ptr), c->ws_shadow.bytes,
                                          call_words, per_slot_level + (merged ? (1ull << 40) : 0ull), max_pool,
                                          static_cast<unsigned long long>(levels), static_cast<unsigned long long>(group)};
    if (std::memcmp(layout, c->shadow_layout, sizeof(layout)) != 0) {
      HG_HIP_CHECK(hipMemsetAsync(c->ws_shadow.ptr, 0, c->ws_shadow.bytes, s));
      std::memcpy(c->shadow_layout, layout, sizeof(layout));
    }
  }
  uint32_t* shadow = c->ws_shadow.as<uint32_t>();
  unsigned max_nwg_all = 0;
  for (int j = 0; j < count; ++j) {
    const int g = j / group, q = j % group;
    const size_t i = scans[j];
    const unsigned long long first = scan_offsets[i] - scan_offsets[0];
    const unsigned nj = static_cast<unsigned>(scan_offsets[i + 1] - scan_offsets[i]);
    InsertJob& J = jobs[j];
    std::memset(&J, 0, sizeof(J));
    J.n = nj;
    J.nwg = (nj + 255u) / 256u;
    max_nwg_all = std::max(max_nwg_all, J.nwg);
    J.records_per_level = nj * kSlots;
    J.xyz = d_xyz + 3 * first;
    // offsets inside the group's buffers
    size_t rec_off = 0, run_off = 0, hit_off = 0, work_off = 0;
    for (int k = g * group; k < j; ++k) {
      const unsigned long long nk = scan_offsets[scans[k] + 1] - scan_offsets[scans[k]];
      rec_off += nk * kSlots * levels;
      run_off += run_info_units(nk, (nk + 255u) / 256u, levels);
      hit_off += ((nk + 255u) / 256u) * kMaxInsLevels;
      work_off += (std::min<size_t>(nk * kMaxRuns, max_blocks) + nk * kSlots / 256u + 64u) * levels;
    }
    J.rec_keys = c->ws_keys_a.as<uint32_t>() + rec_off;
    J.rec_vals = c->ws_vals_a.as<uint32_t>() + rec_off;
    J.runs = c->ws_keys_b.as<RunInfo>() + run_off;
    J.wg_hits = c->ws_counts.as<unsigned>() + hit_off;
    PyramidIns& P = J.P;
    P = P0;
    P.d_pose = nullptr;
    P.accumulate = 1;
    P.shared = 1;
    P.slice_records = stream_slice_records(c);
    P.scan0.begin = 0;
    P.scan0.count = nj;
    std::memcpy(P.scan0.origin, origins + 3 * i, sizeof(P.scan0.origin));
    if (poses_tq) std::memcpy(P.scan0.pose, poses_tq + 7 * i, sizeof(P.scan0.pose));
    else std::memset(P.scan0.pose, 0, sizeof(P.scan0.pose));
    const size_t per_level = std::min<size_t>(static_cast<size_t>(nj) * kMaxRuns, max_blocks) +
                             static_cast<size_t>(nj) * kSlots / 256u + 64u;
    for (int l = 0; l < levels; ++l) {
      LevelIns& L = P.lv[l];
      if (L.gate) L.gate += first;
      if (group_merged[g]) {  // the level's items of the whole group
        L.g.work = c->ws_offsets.as<uint4>() + item_cap * l;
        L.g.work_capacity = static_cast<uint32_t>(item_cap);
      } else {
        L.g.work = c->ws_offsets.as<uint4>() + work_off + per_level * l;
        L.g.work_capacity = static_cast<uint32_t>(per_level);
      }
      L.g.call = shadow + (static_cast<size_t>(q) * levels + l) * 4u;
      uint32_t* base = shadow + call_words + (static_cast<size_t>(q) * levels + l) * per_slot_level;
      L.g.bin_count = base;
      L.g.bin_offset = base + max_pool;
      L.g.touched = base + 2u * max_pool;
    }
    attach_heavy(c, P, static_cast<size_t>(nj) * kSlots, 1024u, 0, 0);
  }
  if (!P0.accumulate)  // hits and updates of this call start from zero; the jobs add to them
    for (int l = 0; l < levels; ++l)
      HG_HIP_CHECK(hipMemsetAsync(P0.lv[l].g.counters + 2, 0, 4 * sizeof(uint32_t), s));
  const InsertJob* d_jobs = c->ws_sjobs.as<InsertJob>();
  HG_HIP_CHECK(hipMemcpyAsync(c->ws_sjobs.ptr, jobs, table_bytes, hipMemcpyHostToDevice, s));
  if ((rc = stage_commit(c)) != HG_OK) return rc;
  for (int g0 = 0; g0 < count; g0 += group) {
    const int gn = std::min(group, count - g0);
    unsigned max_nwg = 0;
    unsigned long long units = 0;
    for (int j = g0; j < g0 + gn; ++j) {
      max_nwg = std::max(max_nwg, jobs[j].nwg);
      units += static_cast<unsigned long long>(jobs[j].n) * levels;
    }
    {
      ProfScope ps(c, HG_K_RAY_COUNT, units);
      hipLaunchKernelGGL(k_bin_count_jobs, dim3(max_nwg, gn * levels), dim3(256), 0, s, d_jobs + g0, levels);
    }
#ifndef HG_BIN_STAMPS
    StreamGroup SG;
    std::memset(&SG, 0, sizeof(SG));
    const bool merged_g = merged && group_merged[g0 / group] != 0;
    if (merged_g) {
      uint32_t* claim0 = shadow + call_words + per_slot_level * levels * group;
      for (int l = 0; l < levels; ++l) {
        SG.claim[l] = claim0 + max_pool * l;
        uint2* tables = reinterpret_cast<uint2*>(c->ws_offsets.as<uint4>() + item_cap * levels) + (kUnitTiers + 1u) * unit_cap * l;
        SG.wg_units[l] = tables;
        SG.wave_units[l] = tables + kUnitTiers * unit_cap;
      }
      SG.counts = claim0 + max_pool * levels;
      for (int l = 0; l < levels; ++l) SG.union_list[l] = SG.counts + 8u * kMaxInsLevels + max_pool * l;
      SG.unit_capacity = static_cast<uint32_t>(unit_cap);
      SG.item_capacity = static_cast<uint32_t>(item_cap);
      if (++c->stream_epoch == 0u) ++c->stream_epoch;  // (0 = never claimed; a tag of 2^32 groups ago cannot be met again: the layout is re-zeroed long before)
      SG.epoch = c->stream_epoch;
      SG.slice_records = stream_slice_records(c);
    }
#endif
    {
      ProfScope ps(c, HG_K_SCAN, gn * levels);
#ifndef HG_BIN_STAMPS
      if (merged_g) hipLaunchKernelGGL(k_stream_offsets_jobs, dim3(gn * levels), dim3(1024), 0, s, d_jobs + g0, levels);
      else
#endif
        hipLaunchKernelGGL(k_bin_offsets_jobs, dim3(gn * levels), dim3(1024), 0, s, d_jobs + g0, levels);
    }
    {
      ProfScope ps(c, HG_K_RAY_EXPAND, units);
      hipLaunchKernelGGL(k_bin_scatter_jobs, dim3(max_nwg, gn * levels), dim3(256), 0, s, d_jobs + g0, levels);
    }
    if (want_stats)
      for (int j = g0; j < g0 + gn; ++j)
        hipLaunchKernelGGL(k_sum_stats, dim3(1), dim3(256), 0, s, jobs[j].P, jobs[j].wg_hits, jobs[j].nwg, nullptr, 0u);
    HG_HIP_CHECK(hipGetLastError());
    {
      ProfScope ps(c, HG_K_APPLY, units * kSlots, static_cast<unsigned>(gn));
#ifndef HG_BIN_STAMPS
      if (merged_g) {
        // one apply launch for the group: units of (block, voxel slice) x scans (k_stream_units)
        hipLaunchKernelGGL(k_stream_union, dim3(kStreamUnitWgs, static_cast<unsigned>(levels)), dim3(256), 0, s, d_jobs + g0, gn, levels, SG);
        hipLaunchKernelGGL(k_stream_units, dim3(kStreamUnitWgs, static_cast<unsigned>(levels)), dim3(512), 0, s, d_jobs + g0, gn,
                           levels, SG, static_cast<const uint32_t*>(jobs[g0].rec_keys));
        hipLaunchKernelGGL(k_bin_apply_stream, dim3(1024u, static_cast<unsigned>(levels)), dim3(kBinThreads), 0, s, jobs[g0].P,
                           jobs[g0].rec_keys, jobs[g0].rec_vals, SG);
        hipLaunchKernelGGL(k_stream_reset, dim3(1), dim3(64), 0, s, SG, levels, d_jobs + g0, gn);
      } else {
        for (int j = g0; j < g0 + gn; ++j)  // pyramid as kernel argument (scalar registers), as scan by scan
          hipLaunchKernelGGL(k_bin_apply, apply_grid(c, levels), dim3(kBinThreads), 0, s, jobs[j].P, jobs[j].rec_keys,
                             jobs[j].rec_vals);
      }
#endif
    }
    HG_HIP_CHECK(hipGetLastError());
  }
#ifdef HG_COUNT_STAMPS
  {
    HG_HIP_CHECK(hipStreamSynchronize(s));
    unsigned long long st[16], zero[16] = {0};
    HG_HIP_CHECK(hipMemcpyFromSymbol(st, HIP_SYMBOL(hg::g_count_stamps), sizeof(st)));
    HG_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(hg::g_count_stamps), zero, sizeof(zero)));
    const double tot = double(st[0] + st[1] + st[2] + st[3] + st[4] + st[5]);
    if (tot > 0)
      fprintf(stderr, "k_bin_count_jobs phases, share of wavefront wall time: ray %.1f%%  key loads %.1f%%  hash path %.1f%%  "
              "LDS aggregation %.1f%%  bin atomics %.1f%%  tables %.1f%%; %.2f us per wavefront; %llu wavefronts, %.1f%% entered the hash path, "
              "%llu runs through it\n", 100 * st[0] / tot, 100 * st[1] / tot, 100 * st[2] / tot, 100 * st[3] / tot, 100 * st[4] / tot,
              100 * st[5] / tot, tot * 0.01 / double(st[6] ? st[6] : 1), st[6], 100.0 * st[7] / double(st[6] ? st[6] : 1), st[8]);
  }
#endif
  return HG_OK;
}

}  // namespace

extern "C" {

int hg_pyramid_insert_batch(hg_grid* const* grids, const hg_insert_opts* opts, int levels,
                            const float* origins, const float* xyz, const uint64_t* scan_offsets,
                            size_t n_scans, size_t width, const float* poses_tq, int mode,
                            int memspace, hg_insert_stats* stats) {
  return pyramid_insert_impl(grids, opts, levels, origins, xyz, scan_offsets, n_scans, width,
                             poses_tq, nullptr, mode, memspace, stats);
}

}  // extern "C"

// poses_tq: host float poses per scan (also used for the key window when d_pose_tq is given, where
// it must hold an approximation of the device pose); d_pose_tq: fp64 pose in device memory that
// the kernels read instead (single-scan calls only).
int hg::pyramid_insert_impl(hg_grid* const* grids, const hg_insert_opts* opts, int levels,
                            const float* origins, const float* xyz, const uint64_t* scan_offsets,
                            size_t n_scans, size_t width, const float* poses_tq,
                            const double* d_pose_tq, int mode, int memspace, hg_insert_stats* stats,
                            const float* d_origin) {
  if (d_pose_tq && (n_scans != 1 || !poses_tq)) return HG_ERR_INVALID;
  if (d_origin && (n_scans != 1 || d_pose_tq)) return HG_ERR_INVALID;
  if (!grids || !opts || levels < 1 || levels > kMaxInsLevels || !origins || !scan_offsets || n_scans == 0)
    return HG_ERR_INVALID;
  if (mode != HG_INSERT_EXACT && mode != HG_INSERT_FAST) return HG_ERR_INVALID;
  hg_ctx* c = grids[0] ? grids[0]->ctx : nullptr;
  if (!c) {
    if (grids[0]) set_last_error("the handle's context has been destroyed");
    return HG_ERR_INVALID;
  }
  for (int l = 0; l < levels; ++l) {
    if (!grids[l] || grids[l]->ctx != c) return HG_ERR_INVALID;
    if (opts[l].project_sdf_distance_to_scan_normal && opts[l].normal_computation_method != 1) {
      set_last_error("only normal_computation_method = CLOUD_STRUCTURE is implemented on the device path");
      return HG_ERR_UNSUPPORTED;
    }
  }
  hipStream_t s = c->stream;
  HG_HIP_CHECK(hipSetDevice(c->device));
  if (!stats) {  // asynchronous call: report what earlier asynchronous calls on these grids have left behind
    const int arc = async_status_grids(grids, levels);
    if (arc != HG_OK) return arc;
  }
  const unsigned long long n_total = scan_offsets[n_scans] - scan_offsets[0];
  if (n_total && !xyz) return HG_ERR_INVALID;
  for (size_t i = 0; i < n_scans; ++i)
    if (scan_offsets[i + 1] < scan_offsets[i]) return HG_ERR_INVALID;


  const float* d_xyz = nullptr;
  if (n_total) {
    if (memspace == HG_HOST) {
      int rc = c->ws_points.reserve(n_total * 3 * sizeof(float));
      if (rc != HG_OK) return rc;
      HG_HIP_CHECK(hipMemcpyAsync(c->ws_points.ptr, xyz + 3 * scan_offsets[0],
                                  n_total * 3 * sizeof(float), hipMemcpyHostToDevice, s));
      d_xyz = c->ws_points.as<float>();
    } else {
      d_xyz = xyz + 3 * scan_offsets[0];
    }
  }

  // insertion_ratio decimation masks (host, sequential recurrence on the index only)
  const uint8_t* d_gate[kMaxInsLevels] = {nullptr, nullptr, nullptr, nullptr};
  {
    size_t need = 0;
    for (int l = 0; l < levels; ++l)
      if (opts[l].insertion_ratio < 1.0) need += n_total;
    if (need) {
      int rc = c->ws_gate.reserve(need);
      if (rc != HG_OK) return rc;
      std::vector<uint8_t> gate(n_total);
      size_t off = 0;
      for (int l = 0; l < levels; ++l) {
        if (!(opts[l].insertion_ratio < 1.0)) continue;
        for (size_t i = 0; i < n_scans; ++i)
          build_gate(opts[l].insertion_ratio, scan_offsets[i + 1] - scan_offsets[i],
                     gate.data() + (scan_offsets[i] - scan_offsets[0]));
        HG_HIP_CHECK(hipMemcpyAsync(c->ws_gate.as<uint8_t>() + off, gate.data(), n_total,
                                    hipMemcpyHostToDevice, s));
        HG_HIP_CHECK(hipStreamSynchronize(s));
        d_gate[l] = c->ws_gate.as<uint8_t>() + off;
        off += n_total;
      }
    }
  }

  // choose the path
  bool fixed_ok = true, unit_weight = true;
  for (int l = 0; l < levels; ++l) {
    if (opts[l].num_free_space_voxels > 0 || !(opts[l].relative_truncation_distance <= 3.0)) fixed_ok = false;
    if (!(static_cast<float>(opts[l].weight_function_epsilon) >= 1.0f)) unit_weight = false;
  }
  if (mode == HG_INSERT_FAST) {
    if (!fixed_ok || !unit_weight) {
      set_last_error("HG_INSERT_FAST needs unit update weights (weight_function_epsilon >= 1), no free-space "
                     "voxels and relative_truncation_distance <= 3");
      return HG_ERR_UNSUPPORTED;
    }
  }
  if (!fixed_ok && levels > 1) {
    // general options: fall back to one compaction pass per level
    int rc = HG_OK;
    if (d_pose_tq) return HG_ERR_UNSUPPORTED;
    for (int l = 0; l < levels && rc == HG_OK; ++l)
      rc = pyramid_insert_impl(grids + l, opts + l, 1, origins, xyz, scan_offsets, n_scans, width,
                               poses_tq, nullptr, mode, memspace, stats ? stats + l : nullptr, d_origin);
    return rc;
  }

  // grid-frame origins (for the 32-bit key window)
  std::vector<float> go(3 * n_scans);
  for (size_t i = 0; i < n_scans; ++i) {
    if (poses_tq) host_transform(poses_tq + 7 * i, origins + 3 * i, go.data() + 3 * i);
    else std::memcpy(go.data() + 3 * i, origins + 3 * i, 3 * sizeof(float));
  }
  bool key32 = fixed_ok && levels <= 3;
  PyramidIns P;
  std::memset(&P, 0, sizeof(P));
  P.levels = levels;
  P.d_pose = d_pose_tq;
  P.d_origin = d_origin;
  P.host_flags = stats ? nullptr : c->async_flags;
  if (d_pose_tq && !fixed_ok) return HG_ERR_UNSUPPORTED;
  for (int l = 0; l < levels; ++l) {
    LevelIns& L = P.lv[l];
    L.g = grids[l]->view;
    P.flag_slot[l] = static_cast<uint16_t>(grids[l]->flag_slot);
    L.p = make_params(opts[l], grids[l], poses_tq != nullptr, width);
    L.gate = d_gate[l];
    const float res = L.g.resolution;
    for (int a = 0; a < 3; ++a) {
      const long ci = std::lround(go[a] / res);
      L.base[a] = static_cast<int>((ci + kIndexOffset) >> 3);
    }
    // every sample lies within max_range + tau (+1 cell) of its scan origin
    double reach = 0.0;
    for (size_t i = 0; i < n_scans; ++i)
      for (int a = 0; a < 3; ++a)
        reach = std::max(reach, static_cast<double>(std::fabs(go[3 * i + a] - go[a])));
    reach += opts[l].max_range + L.p.truncation_distance + 2.0 * res + ((d_pose_tq || d_origin) ? 2.0 : 0.0);
    if (!(reach / (8.0 * res) < 62.0)) key32 = false;
  }

  // chunk by scans so the record workspace stays bounded; when the binned path applies every scan
  // is its own chunk (same per-voxel chain length as one sorted batch, without the global sort)
  const bool binned_ok = fixed_ok && unit_weight && c->opt(OPT_INSERT_SORT) == 0;
  // a voxel receives at most one update per return: FAST chunks stay below the 20-bit update count
  const unsigned long long kMaxChunkPoints = mode == HG_INSERT_FAST ? (1ull << 20) - 1ull : 4ull << 20;
  const bool fast = mode == HG_INSERT_FAST;
  std::vector<ScanTable> table;
  size_t s0 = 0;
  int rc = HG_OK;
  bool chunk_launched = false;
  // A call that carries a stream of scans (known poses) on the exact binned path:
  //  - scans of >= 2^14 returns: front ends grouped, one apply launch per scan (insert_stream_grouped;
  //    HG_STREAM_GROUP = scans per group, 0 = off);
  //  - many small scans: multi-scan chunks of up to 2^17 returns, pipelined over two streams: front
  //    end of chunk k + 1 next to the apply pass of chunk k (HG_INSERT_PIPELINE=0: one stream).
  bool pipelined = false;
  if (binned_ok && !fast && n_scans > 1 && n_total > (1ull << 17)) {
    bool fits = true;  // every scan within the 23-bit seq of the binned records
    for (size_t i = 0; i < n_scans; ++i)
      if (scan_offsets[i + 1] - scan_offsets[i] >= (1ull << 20)) fits = false;
    if (fits && n_total / n_scans >= (1ull << 14) && c->opt(OPT_STREAM_GROUP) > 0) {
      rc = insert_stream_grouped(c, P, origins, d_xyz, scan_offsets, n_scans, poses_tq, stats != nullptr);
      s0 = n_scans;  // done (or failed)
    } else {
      pipelined = fits && c->opt(OPT_INSERT_PIPELINE) != 0;
    }
  }
  int pipe_chunks = 0;
  // The chunks of the call (whole scans up to chunk_cap returns) and ONE table of all scans, every entry relative to its
  // chunk's first return. Exact binned path: one pass takes whole scans up to 2^17 returns together (a scan of up to
  // 2^20 - 1 returns on its own: the 23-bit seq of its records). Several small scans share their launches; larger
  // chunks were measured slower (ten 100k-point scans per pass: 187 us per scan in the apply kernel against 100 us
  // scan by scan) because every slice of a bin reads the whole bin and voxels beyond one LDS pass are applied in
  // rounds, both of which grow with the chunk. The tolerance mode has neither: its chunks are as large as the
  // 20-bit update count of a voxel allows. The table travels through the context's pinned staging once per call
  // (round 6; an upload and a stream synchronisation per chunk before).
  const unsigned long long chunk_cap = (binned_ok && !fast) ? (1ull << 17) : kMaxChunkPoints;
  struct Chunk { size_t s0, s1; unsigned long long pts; };
  std::vector<Chunk> chunks;
  const ScanTable* d_table = nullptr;
  if (s0 < n_scans && rc == HG_OK) {
    void* staged = nullptr;
    if ((rc = stage_table(c, sizeof(ScanTable) * n_scans, &staged)) != HG_OK) return rc;
    ScanTable* tab = static_cast<ScanTable*>(staged);
    size_t a = s0;
    bool multi = false;
    // (tolerance mode: chunks of about equal size -- 32 scans of 100k returns go as 4 x 8, not 10 + 10 + 10 + 2:
    // the launches of a chunk of two cost nearly those of ten)
    unsigned long long cap = chunk_cap;
    if (fast && n_total > chunk_cap) {
      const unsigned long long k = (n_total + chunk_cap - 1) / chunk_cap;
      cap = std::min(chunk_cap, (n_total + k - 1) / k + n_total / std::max<unsigned long long>(1, n_scans) / 2);
    }
    while (a < n_scans) {
      size_t b = a;
      unsigned long long pts = 0;
      while (b < n_scans && (b == a || pts + (scan_offsets[b + 1] - scan_offsets[b]) <= cap)) {
        ScanTable& t = tab[b];
        t.begin = scan_offsets[b] - scan_offsets[a];
        t.count = scan_offsets[b + 1] - scan_offsets[b];
        std::memcpy(t.origin, origins + 3 * b, sizeof(t.origin));
        if (poses_tq) std::memcpy(t.pose, poses_tq + 7 * b, sizeof(t.pose));
        else std::memset(t.pose, 0, sizeof(t.pose));
        pts += scan_offsets[b + 1] - scan_offsets[b];
        ++b;
      }
      chunks.push_back({a, b, pts});
      multi = multi || b - a > 1;
      a = b;
    }
    table.assign(tab, tab + n_scans);
    if (multi || !fixed_ok) {
      if ((rc = c->ws_scan_table.reserve(sizeof(ScanTable) * n_scans)) != HG_OK) return rc;
      HG_HIP_CHECK(hipMemcpyAsync(c->ws_scan_table.ptr, tab, sizeof(ScanTable) * n_scans, hipMemcpyHostToDevice, s));
      if ((rc = stage_commit(c)) != HG_OK) return rc;
      d_table = c->ws_scan_table.as<ScanTable>();
      if (d_origin)  // (single-scan calls only) the origin the unwarping left in device memory replaces the host's guess
        HG_HIP_CHECK(hipMemcpyAsync(reinterpret_cast<char*>(c->ws_scan_table.ptr) + offsetof(ScanTable, origin), d_origin,
                                    3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
  }
  for (size_t ci = 0; ci < chunks.size() && rc == HG_OK; ++ci) {
    s0 = chunks[ci].s0;
    const size_t s1 = chunks[ci].s1;
    const unsigned long long pts = chunks[ci].pts;
    const size_t tsize = s1 - s0;
    if (pts > 0) {
      const ScanTable* d_scans = (tsize > 1 || !fixed_ok) ? d_table + s0 : nullptr;
      const unsigned long long first = scan_offsets[s0] - scan_offsets[0];
      PyramidIns Pc = P;
      Pc.scan0 = table[s0];
      Pc.accumulate = chunk_launched ? 1 : 0;  // counters restart with the first chunk that runs
      chunk_launched = true;
      for (int l = 0; l < levels; ++l)
        if (Pc.lv[l].gate) Pc.lv[l].gate += first;
      if (fast) {
        if (pts > kMaxChunkPoints) {
          set_last_error("HG_INSERT_FAST: a single scan is limited to 2^20 - 1 returns");
          return HG_ERR_UNSUPPORTED;
        }
        rc = insert_chunk_fast(c, Pc, d_scans, static_cast<uint32_t>(tsize), d_xyz + 3 * first, pts, stats != nullptr);
      } else if (binned_ok && pts < (1ull << 20)) {
        rc = insert_chunk_binned(c, Pc, tsize > 1 ? d_scans : nullptr, static_cast<uint32_t>(tsize),
                                 d_xyz + 3 * first, pts, stats != nullptr, pipelined ? pipe_chunks : -1);
        if (pipelined && rc == HG_OK) ++pipe_chunks;
      } else if (fixed_ok) {
        const bool ws = stats != nullptr;
        if (key32 && unit_weight)
          rc = insert_chunk_fixed<uint32_t, uint32_t>(c, Pc, d_scans, static_cast<uint32_t>(tsize), d_xyz + 3 * first, pts, ws);
        else if (key32)
          rc = insert_chunk_fixed<uint32_t, unsigned long long>(c, Pc, d_scans, static_cast<uint32_t>(tsize), d_xyz + 3 * first, pts, ws);
        else if (unit_weight)
          rc = insert_chunk_fixed<unsigned long long, uint32_t>(c, Pc, d_scans, static_cast<uint32_t>(tsize), d_xyz + 3 * first, pts, ws);
        else
          rc = insert_chunk_fixed<unsigned long long, unsigned long long>(c, Pc, d_scans, static_cast<uint32_t>(tsize), d_xyz + 3 * first, pts, ws);
      } else {
        rc = insert_chunk_compact(grids[0], Pc.lv[0].p, d_scans, static_cast<uint32_t>(tsize),
                                  d_xyz + 3 * first, pts, Pc.lv[0].gate, Pc.accumulate == 0);
      }
    }
  }
  if (pipe_chunks > 0)  // later work on the context's stream is ordered after the last apply pass
    HG_HIP_CHECK(hipStreamWaitEvent(s, c->ev_apply[(pipe_chunks - 1) & 1], 0));
  if (rc != HG_OK) return rc;
  if (stats) {
    for (int l = 0; l < levels; ++l) {
      const int r2 = read_stats(grids[l], stats + l);
      if (r2 != HG_OK) rc = r2;
    }
  }
  return rc;
}

extern "C" {

int hg_grid_insert_batch(hg_grid* grid, const hg_insert_opts* opts, const float* origins,
                         const float* xyz, const uint64_t* scan_offsets, size_t n_scans,
                         size_t width, const float* poses_tq, int mode, int memspace,
                         hg_insert_stats* stats) {
  if (!grid || !opts) return HG_ERR_INVALID;
  hg_grid* g[1] = {grid};
  return hg_pyramid_insert_batch(g, opts, 1, origins, xyz, scan_offsets, n_scans, width, poses_tq,
                                 mode, memspace, stats);
}

int hg_grid_insert(hg_grid* grid, const hg_insert_opts* opts, const float origin[3],
                   const float* xyz, size_t n, size_t width, const float* pose_tq, int mode,
                   int memspace, hg_insert_stats* stats) {
  if (!origin) return HG_ERR_INVALID;
  const uint64_t offsets[2] = {0, n};
  return hg_grid_insert_batch(grid, opts, origin, xyz, offsets, 1, width, pose_tq, mode, memspace,
                              stats);
}

int hg_pyramid_insert(hg_grid* const* grids, const hg_insert_opts* opts, int levels,
                      const float origin[3], const float* xyz, size_t n, size_t width,
                      const float* pose_tq, int mode, int memspace, hg_insert_stats* stats) {
  if (!origin) return HG_ERR_INVALID;
  const uint64_t offsets[2] = {0, n};
  return hg_pyramid_insert_batch(grids, opts, levels, origin, xyz, offsets, 1, width, pose_tq, mode,
                                 memspace, stats);
}

int hg_grid_status(hg_grid* grid, hg_insert_stats* stats) {
  HG_REQUIRE_CTX(grid);
  return read_stats(grid, stats);
}

}  // extern "C"
