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
#include <cstring>
#include <string.h>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "hg_internal.h"

namespace hg {

struct InsertParams {
  double min_range, max_range;
  float truncation_distance;  // float(rel_trunc(double) * resolution(float))   (:298-299)
  float maximum_weight;       // static_cast<float>(options_.maximum_weight())   (:735)
  float epsilon, sigma;
  int free_space;             // num_free_space_voxels > 0                        (:303)
  int has_pose;
};

struct ScanTable {        // per scan of a batch
  unsigned long long begin;  // first point index
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
};

__device__ inline float norm3(float x, float y, float z) {
  // Eigen fixed-size reduction order: x0 + (x1 + x2)
  return sqrtf(x * x + (y * y + z * z));
}

// Eigen Quaternion<float>::_transformVector followed by + translation
// (transform/rigid_transform.h:193-197, sensor/range_data.cc:25-39).
__device__ inline void transform_point(const float* pose, float& x, float& y, float& z) {
  const float qw = pose[3], qx = pose[4], qy = pose[5], qz = pose[6];
  float ux = qy * z - qz * y, uy = qz * x - qx * z, uz = qx * y - qy * x;
  ux = ux + ux; uy = uy + uy; uz = uz + uz;
  const float cx = qy * uz - qz * uy, cy = qz * ux - qx * uz, cz = qx * uy - qy * ux;
  const float rx = x + qw * ux + cx, ry = y + qw * uy + cy, rz = z + qw * uz + cz;
  x = rx + pose[0]; y = ry + pose[1]; z = rz + pose[2];
}

__device__ inline uint32_t find_scan(const ScanTable* scans, uint32_t n_scans, unsigned long long i) {
  uint32_t lo = 0, hi = n_scans;  // last scan with begin <= i
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (scans[mid].begin <= i) lo = mid; else hi = mid;
  }
  return lo;
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
  if (range < tau) return r;
  const float ratio = tau / range;
  float b_x, b_y, b_z;
  if (p.free_space) {
    b_x = ox; b_y = oy; b_z = oz;
  } else {
    const float s = 1.0f - ratio;
    b_x = ox + s * rx; b_y = oy + s * ry; b_z = oz + s * rz;
  }
  const float e = 1.0f + ratio;
  const float e_x = ox + e * rx, e_y = oy + e * ry, e_z = oz + e * rz;
  r.bx = cell_index_1d(b_x, g.resolution);
  r.by = cell_index_1d(b_y, g.resolution);
  r.bz = cell_index_1d(b_z, g.resolution);
  r.dx = cell_index_1d(e_x, g.resolution) - r.bx;
  r.dy = cell_index_1d(e_y, g.resolution) - r.by;
  r.dz = cell_index_1d(e_z, g.resolution) - r.bz;
  r.n = max(abs(r.dx), max(abs(r.dy), abs(r.dz)));
  r.range = range;
  r.ox = ox; r.oy = oy; r.oz = oz;
  r.valid = r.n > 0 && r.n < (1 << 15);
  return r;
}

__device__ inline void ray_sample(const GridView& g, const InsertParams& p, const Ray& r, int pos,
                                  int& cx, int& cy, int& cz, float& tsd, float& weight) {
  const float fp = static_cast<float>(pos), fn = static_cast<float>(r.n);
  cx = r.bx + static_cast<int>(roundf(static_cast<float>(r.dx) * fp / fn));
  cy = r.by + static_cast<int>(roundf(static_cast<float>(r.dy) * fp / fn));
  cz = r.bz + static_cast<int>(roundf(static_cast<float>(r.dz) * fp / fn));
  const float ccx = static_cast<float>(cx) * g.resolution;
  const float ccy = static_cast<float>(cy) * g.resolution;
  const float ccz = static_cast<float>(cz) * g.resolution;
  const float dist = norm3(ccx - r.ox, ccy - r.oy, ccz - r.oz);
  const float tau = p.truncation_distance;
  tsd = clampf(r.range - dist, -tau, tau);
  weight = 1.0f;
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
  const unsigned long long m = __ballot(hit);
  if ((threadIdx.x & (kWave - 1)) == 0 && m) atomicAdd(&g.counters[2], static_cast<uint32_t>(__popcll(m)));
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

__global__ void k_apply_runs(GridView g, InsertParams p, const unsigned long long* keys,
                             const unsigned long long* vals, unsigned long long n) {
  const unsigned long long i = blockIdx.x * static_cast<unsigned long long>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  if (k == ~0ull) return;
  if (i > 0 && keys[i - 1] == k) return;  // not a run head
  const uint32_t slot = find_block(g, k >> 9);
  if (slot >= g.max_blocks) return;  // capacity exceeded (flag already set)
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

}  // namespace hg

using namespace hg;

namespace {

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

int insert_chunk(hg_grid* grid, const InsertParams& p, const ScanTable* h_scans, uint32_t n_scans,
                 const float* d_xyz, unsigned long long n, const uint8_t* d_gate) {
  hg_ctx* c = grid->ctx;
  hipStream_t s = c->stream;
  int rc;
  if ((rc = c->ws_scan_table.reserve(sizeof(ScanTable) * n_scans)) != HG_OK) return rc;
  HG_HIP_CHECK(hipMemcpyAsync(c->ws_scan_table.ptr, h_scans, sizeof(ScanTable) * n_scans,
                              hipMemcpyHostToDevice, s));
  if ((rc = c->ws_counts.reserve(sizeof(uint32_t) * (n + 1))) != HG_OK) return rc;
  if ((rc = c->ws_offsets.reserve(sizeof(unsigned long long) * (n + 1))) != HG_OK) return rc;
  const ScanTable* d_scans = c->ws_scan_table.as<ScanTable>();
  uint32_t* d_counts = c->ws_counts.as<uint32_t>();
  unsigned long long* d_offsets = c->ws_offsets.as<unsigned long long>();
  const unsigned wg = 256;
  const unsigned nwg = static_cast<unsigned>((n + wg - 1) / wg);
  HG_HIP_CHECK(hipMemsetAsync(d_counts + n, 0, sizeof(uint32_t), s));
  {
    ProfScope ps(c, HG_K_RAY_COUNT, n);
    hipLaunchKernelGGL(k_ray_count, dim3(nwg), dim3(wg), 0, s, grid->view, p, d_scans, n_scans,
                       d_xyz, n, d_gate, d_counts);
  }
  HG_HIP_CHECK(hipGetLastError());
  // exclusive scan over n+1 counts -> offsets[n] = total number of records
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
  // stable LSD radix sort on the 42 key bits (33 block + 9 voxel); dropped records (~0) need bit 42+
  const unsigned end_bit = 43;
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

}  // namespace

extern "C" {

int hg_grid_insert_batch(hg_grid* grid, const hg_insert_opts* opts, const float* origins,
                         const float* xyz, const uint64_t* scan_offsets, size_t n_scans,
                         size_t width, const float* poses_tq, int mode, int memspace,
                         hg_insert_stats* stats) {
  (void)width;
  if (!grid || !opts || !origins || !scan_offsets || n_scans == 0) return HG_ERR_INVALID;
  if (mode != HG_INSERT_EXACT) return HG_ERR_UNSUPPORTED;
  if (opts->project_sdf_distance_to_scan_normal) {
    set_last_error("project_sdf_distance_to_scan_normal is not implemented on the device path");
    return HG_ERR_UNSUPPORTED;
  }
  hg_ctx* c = grid->ctx;
  hipStream_t s = c->stream;
  HG_HIP_CHECK(hipSetDevice(c->device));
  const unsigned long long n_total = scan_offsets[n_scans] - scan_offsets[0];
  if (n_total && !xyz) return HG_ERR_INVALID;
  for (size_t i = 0; i < n_scans; ++i)
    if (scan_offsets[i + 1] < scan_offsets[i]) return HG_ERR_INVALID;

  InsertParams p;
  p.min_range = opts->min_range;
  p.max_range = opts->max_range;
  p.truncation_distance =
      static_cast<float>(opts->relative_truncation_distance * static_cast<double>(grid->view.resolution));
  p.maximum_weight = static_cast<float>(opts->maximum_weight);
  p.epsilon = static_cast<float>(opts->weight_function_epsilon);
  p.sigma = static_cast<float>(opts->weight_function_sigma);
  p.free_space = opts->num_free_space_voxels > 0 ? 1 : 0;
  p.has_pose = poses_tq ? 1 : 0;

  // reset per-call counters (hits, updates); keep num_blocks and sticky flags
  HG_HIP_CHECK(hipMemsetAsync(grid->view.counters + 2, 0, 6 * sizeof(uint32_t), s));

  const float* d_xyz = xyz;
  if (memspace == HG_HOST && n_total) {
    int rc = c->ws_points.reserve(n_total * 3 * sizeof(float));
    if (rc != HG_OK) return rc;
    HG_HIP_CHECK(hipMemcpyAsync(c->ws_points.ptr, xyz + 3 * scan_offsets[0],
                                n_total * 3 * sizeof(float), hipMemcpyHostToDevice, s));
    d_xyz = c->ws_points.as<float>();
  } else if (n_total) {
    d_xyz = xyz + 3 * scan_offsets[0];
  }

  const uint8_t* d_gate = nullptr;
  if (opts->insertion_ratio < 1.0 && n_total) {
    std::vector<uint8_t> gate(n_total);
    for (size_t i = 0; i < n_scans; ++i)
      build_gate(opts->insertion_ratio, scan_offsets[i + 1] - scan_offsets[i],
                 gate.data() + (scan_offsets[i] - scan_offsets[0]));
    int rc = c->ws_gate.reserve(n_total);
    if (rc != HG_OK) return rc;
    HG_HIP_CHECK(hipMemcpyAsync(c->ws_gate.ptr, gate.data(), n_total, hipMemcpyHostToDevice, s));
    HG_HIP_CHECK(hipStreamSynchronize(s));  // `gate` goes out of scope
    d_gate = c->ws_gate.as<uint8_t>();
  }

  // Chunk by scans so a chunk's record workspace stays bounded.
  const unsigned long long kMaxChunkPoints = 8ull << 20;
  std::vector<ScanTable> table;
  size_t s0 = 0;
  while (s0 < n_scans) {
    size_t s1 = s0;
    unsigned long long pts = 0;
    table.clear();
    while (s1 < n_scans && (s1 == s0 || pts + (scan_offsets[s1 + 1] - scan_offsets[s1]) <= kMaxChunkPoints)) {
      ScanTable t;
      t.begin = scan_offsets[s1] - scan_offsets[s0];
      std::memcpy(t.origin, origins + 3 * s1, sizeof(t.origin));
      if (poses_tq) std::memcpy(t.pose, poses_tq + 7 * s1, sizeof(t.pose));
      else std::memset(t.pose, 0, sizeof(t.pose));
      table.push_back(t);
      pts += scan_offsets[s1 + 1] - scan_offsets[s1];
      ++s1;
    }
    if (pts > 0) {
      const unsigned long long first = scan_offsets[s0] - scan_offsets[0];
      int rc = insert_chunk(grid, p, table.data(), static_cast<uint32_t>(table.size()),
                            d_xyz + 3 * first, pts, d_gate ? d_gate + first : nullptr);
      if (rc != HG_OK) return rc;
    }
    s0 = s1;
  }

  uint32_t cnt[8];
  HG_HIP_CHECK(hipMemcpyAsync(cnt, grid->view.counters, sizeof(cnt), hipMemcpyDeviceToHost, s));
  HG_HIP_CHECK(hipStreamSynchronize(s));
  hg_insert_stats st;
  st.num_hits = cnt[2];
  st.num_updates = static_cast<uint64_t>(cnt[4]) | (static_cast<uint64_t>(cnt[5]) << 32);
  st.num_blocks = std::min(cnt[0], grid->view.max_blocks);
  st.flags = cnt[1];
  grid->last_stats = st;
  if (stats) *stats = st;
  if (cnt[1] & kFlagCapacity) {
    set_last_error("block pool exhausted: raise max_blocks");
    return HG_ERR_CAPACITY;
  }
  if (cnt[1] & kFlagRange) {
    set_last_error("cell index outside +-8192");
    return HG_ERR_RANGE;
  }
  return HG_OK;
}

int hg_grid_insert(hg_grid* grid, const hg_insert_opts* opts, const float origin[3],
                   const float* xyz, size_t n, size_t width, const float* pose_tq, int mode,
                   int memspace, hg_insert_stats* stats) {
  if (!origin) return HG_ERR_INVALID;
  const uint64_t offsets[2] = {0, n};
  return hg_grid_insert_batch(grid, opts, origin, xyz, offsets, 1, width, pose_tq, mode, memspace,
                              stats);
}

}  // extern "C"
