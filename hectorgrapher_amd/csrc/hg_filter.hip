// hg_filter.hip — VoxelFilter / AdaptiveVoxelFilter on the device (the step before the matching
// path: sensor/internal/voxel_filter.cc:26-69, sensor/internal/adaptive_voxel_filter.h:33-110).
//
// VoxelFilter keeps the FIRST point (input order) that falls into each voxel; cell = lround(p/res)
// per axis. Device form: a hash table keyed by the packed cell (3 x 21 bits) holds the minimum
// point index per voxel (atomicMin), a second pass flags the points that are their voxel's
// minimum, and a stable stream compaction (rocPRIM select, library call) emits their indices in
// input order — the same set and order as the reference's sequential hash-set loop.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string.h>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "hg_internal.h"

namespace hg {

constexpr unsigned long long kEmptyKey = ~0ull;

__device__ inline bool cell_key21(const float* p, float res, unsigned long long* key) {
  const int x = round_to_int(p[0] / res), y = round_to_int(p[1] / res), z = round_to_int(p[2] / res);
  const unsigned ux = static_cast<unsigned>(x + (1 << 20)), uy = static_cast<unsigned>(y + (1 << 20)),
                 uz = static_cast<unsigned>(z + (1 << 20));
  if ((ux | uy | uz) >> 21) return false;
  *key = (static_cast<unsigned long long>(uz) << 42) | (static_cast<unsigned long long>(uy) << 21) | ux;
  return true;
}

__device__ inline uint32_t mix64(unsigned long long k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 29;
  return static_cast<uint32_t>(k) ^ static_cast<uint32_t>(k >> 32);
}

// FilterByMaxRange (adaptive_voxel_filter.h:33-44): mask[i] = |p_i| <= max_range
__global__ void k_range_mask(const float* pts, unsigned n, int stride, float max_range, uint8_t* mask) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = pts + static_cast<size_t>(i) * stride;
  const float r = sqrtf(p[0] * p[0] + (p[1] * p[1] + p[2] * p[2]));  // Eigen norm() order
  mask[i] = (r <= max_range) ? 1 : 0;
}

__global__ void k_vf_insert(const float* pts, unsigned n, int stride, float res, const uint8_t* mask,
                            unsigned long long* keys, uint32_t* min_idx, uint32_t table_mask,
                            uint32_t* err) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || (mask && !mask[i])) return;
  unsigned long long key;
  if (!cell_key21(pts + static_cast<size_t>(i) * stride, res, &key)) {
    atomicOr(err, kFlagRange);
    return;
  }
  uint32_t h = mix64(key) & table_mask;
  for (uint32_t probe = 0; probe <= table_mask; ++probe) {
    unsigned long long e = keys[h];
    if (e == kEmptyKey) {
      const unsigned long long prev = atomicCAS(&keys[h], kEmptyKey, key);
      e = (prev == kEmptyKey) ? key : prev;
    }
    if (e == key) {
      atomicMin(&min_idx[h], i);
      return;
    }
    h = (h + 1) & table_mask;
  }
  atomicOr(err, kFlagCapacity);
}

__global__ void k_vf_flags(const float* pts, unsigned n, int stride, float res, const uint8_t* mask,
                           const unsigned long long* keys, const uint32_t* min_idx,
                           uint32_t table_mask, uint8_t* flags) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint8_t keep = 0;
  unsigned long long key;
  if ((!mask || mask[i]) && cell_key21(pts + static_cast<size_t>(i) * stride, res, &key)) {
    uint32_t h = mix64(key) & table_mask;
    for (uint32_t probe = 0; probe <= table_mask; ++probe) {
      const unsigned long long e = keys[h];
      if (e == key) {
        keep = (min_idx[h] == i) ? 1 : 0;
        break;
      }
      if (e == kEmptyKey) break;
      h = (h + 1) & table_mask;
    }
  }
  flags[i] = keep;
}

__global__ void k_gather_xyz(const float* pts, int stride, const uint32_t* idx, const uint32_t* count,
                             float* out) {
  const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= *count) return;
  const float* p = pts + static_cast<size_t>(idx[j]) * stride;
  out[3 * j] = p[0];
  out[3 * j + 1] = p[1];
  out[3 * j + 2] = p[2];
}

struct FilterCall {
  hg_ctx* c;
  const float* d_pts;
  unsigned n;
  int stride;
  const uint8_t* d_mask;  // subset (or nullptr)
  unsigned long long* d_keys;
  uint32_t* d_min;
  uint32_t table_mask;
  uint8_t* d_flags;
  uint32_t* d_idx;    // selected indices (n capacity)
  uint32_t* d_count;  // [0] selected count, [1] error flags
};

// One VoxelFilter(resolution).Filter pass over the (masked) cloud; count read back to the host.
int voxel_filter_pass(FilterCall& f, float res, const uint8_t* select_mask_only, size_t* count) {
  hipStream_t s = f.c->stream;
  const unsigned wg = 256, nwg = (f.n + wg - 1) / wg;
  if (select_mask_only) {
    // no voxel filtering: select the masked points themselves
    size_t tb = 0;
    HG_HIP_CHECK(rocprim::select(nullptr, tb, rocprim::counting_iterator<uint32_t>(0), select_mask_only,
                                 f.d_idx, f.d_count, f.n, s));
    int rc = f.c->ws_temp.reserve(tb);
    if (rc != HG_OK) return rc;
    HG_HIP_CHECK(rocprim::select(f.c->ws_temp.ptr, tb, rocprim::counting_iterator<uint32_t>(0),
                                 select_mask_only, f.d_idx, f.d_count, f.n, s));
  } else {
    const size_t cap = static_cast<size_t>(f.table_mask) + 1;
    HG_HIP_CHECK(hipMemsetAsync(f.d_keys, 0xFF, cap * sizeof(unsigned long long), s));
    HG_HIP_CHECK(hipMemsetAsync(f.d_min, 0xFF, cap * sizeof(uint32_t), s));
    hipLaunchKernelGGL(k_vf_insert, dim3(nwg), dim3(wg), 0, s, f.d_pts, f.n, f.stride, res, f.d_mask,
                       f.d_keys, f.d_min, f.table_mask, f.d_count + 1);
    HG_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_vf_flags, dim3(nwg), dim3(wg), 0, s, f.d_pts, f.n, f.stride, res, f.d_mask,
                       f.d_keys, f.d_min, f.table_mask, f.d_flags);
    HG_HIP_CHECK(hipGetLastError());
    size_t tb = 0;
    HG_HIP_CHECK(rocprim::select(nullptr, tb, rocprim::counting_iterator<uint32_t>(0), f.d_flags,
                                 f.d_idx, f.d_count, f.n, s));
    int rc = f.c->ws_temp.reserve(tb);
    if (rc != HG_OK) return rc;
    HG_HIP_CHECK(rocprim::select(f.c->ws_temp.ptr, tb, rocprim::counting_iterator<uint32_t>(0),
                                 f.d_flags, f.d_idx, f.d_count, f.n, s));
  }
  uint32_t h[2];
  HG_HIP_CHECK(hipMemcpyAsync(h, f.d_count, sizeof(h), hipMemcpyDeviceToHost, s));
  HG_HIP_CHECK(hipStreamSynchronize(s));
  if (h[1] & kFlagRange) {
    set_last_error("voxel filter: cell index outside +-2^20 (device key is 3 x 21 bits)");
    return HG_ERR_RANGE;
  }
  if (h[1] & kFlagCapacity) return HG_ERR_CAPACITY;
  *count = h[0];
  return HG_OK;
}

int filter_setup(hg_ctx* c, const float* pts, size_t n, int stride, int memspace, FilterCall* f) {
  if (!c || (n && !pts) || (stride != 3 && stride != 4) || n > 0x7FFFFFFFull) return HG_ERR_INVALID;
  HG_HIP_CHECK(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  f->c = c;
  f->n = static_cast<unsigned>(n);
  f->stride = stride;
  f->d_mask = nullptr;
  int rc;
  if (memspace == HG_HOST) {
    if ((rc = c->ws_points.reserve(std::max<size_t>(16, n * stride * sizeof(float)))) != HG_OK) return rc;
    if (n) HG_HIP_CHECK(hipMemcpyAsync(c->ws_points.ptr, pts, n * stride * sizeof(float), hipMemcpyHostToDevice, s));
    f->d_pts = c->ws_points.as<float>();
  } else {
    f->d_pts = pts;
  }
  uint32_t cap = 1024;
  while (cap < 2 * n) cap <<= 1;
  f->table_mask = cap - 1;
  if ((rc = c->ws_keys_a.reserve(cap * sizeof(unsigned long long))) != HG_OK) return rc;
  if ((rc = c->ws_vals_a.reserve(cap * sizeof(uint32_t))) != HG_OK) return rc;
  // flags | mask | indices | xyz out | counters
  const size_t nb = std::max<size_t>(n, 1);
  if ((rc = c->ws_filter.reserve(2 * ((nb + 255) / 256 * 256) + nb * sizeof(uint32_t) + nb * 3 * sizeof(float) + 256)) != HG_OK) return rc;
  char* base = c->ws_filter.as<char>();
  const size_t fl = (nb + 255) / 256 * 256;
  f->d_flags = reinterpret_cast<uint8_t*>(base);
  f->d_idx = reinterpret_cast<uint32_t*>(base + 2 * fl);
  f->d_count = reinterpret_cast<uint32_t*>(base + 2 * fl + nb * sizeof(uint32_t) + nb * 3 * sizeof(float));
  f->d_keys = c->ws_keys_a.as<unsigned long long>();
  f->d_min = c->ws_vals_a.as<uint32_t>();
  HG_HIP_CHECK(hipMemsetAsync(f->d_count, 0, 16, s));
  return HG_OK;
}

int filter_finish(FilterCall& f, size_t count, uint32_t* indices_out) {
  hg_ctx* c = f.c;
  hipStream_t s = c->stream;
  const size_t nb = std::max<size_t>(f.n, 1);
  const size_t fl = (nb + 255) / 256 * 256;
  float* d_xyz = reinterpret_cast<float*>(c->ws_filter.as<char>() + 2 * fl + nb * sizeof(uint32_t));
  if (count) {
    hipLaunchKernelGGL(k_gather_xyz, dim3(static_cast<unsigned>((count + 255) / 256)), dim3(256), 0, s,
                       f.d_pts, f.stride, f.d_idx, f.d_count, d_xyz);
    HG_HIP_CHECK(hipGetLastError());
  }
  c->filter_idx = f.d_idx;
  c->filter_xyz = d_xyz;
  c->filter_count = count;
  if (indices_out && count)
    HG_HIP_CHECK(hipMemcpyAsync(indices_out, f.d_idx, count * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  HG_HIP_CHECK(hipStreamSynchronize(s));
  return HG_OK;
}

}  // namespace hg

using namespace hg;

extern "C" {

int hg_voxel_filter(hg_ctx* ctx, float resolution, const float* pts, size_t n, int stride,
                    int memspace, uint32_t* indices_out, size_t* count) {
  if (!count || !(resolution > 0.f)) return HG_ERR_INVALID;
  FilterCall f;
  int rc = filter_setup(ctx, pts, n, stride, memspace, &f);
  if (rc != HG_OK) return rc;
  *count = 0;
  if (n == 0) { ctx->filter_count = 0; return HG_OK; }
  size_t k = 0;
  rc = voxel_filter_pass(f, resolution, nullptr, &k);
  if (rc != HG_OK) return rc;
  *count = k;
  return filter_finish(f, k, indices_out);
}

int hg_adaptive_voxel_filter(hg_ctx* ctx, float max_length, float min_num_points, float max_range,
                             const float* pts, size_t n, int stride, int memspace,
                             uint32_t* indices_out, size_t* count) {
  if (!count || !(max_length > 0.f)) return HG_ERR_INVALID;
  FilterCall f;
  int rc = filter_setup(ctx, pts, n, stride, memspace, &f);
  if (rc != HG_OK) return rc;
  *count = 0;
  if (n == 0) { ctx->filter_count = 0; return HG_OK; }
  hipStream_t s = ctx->stream;
  const size_t fl = (static_cast<size_t>(f.n) + 255) / 256 * 256;
  uint8_t* d_mask = f.d_flags + fl;
  hipLaunchKernelGGL(k_range_mask, dim3((f.n + 255) / 256), dim3(256), 0, s, f.d_pts, f.n, f.stride,
                     max_range, d_mask);
  HG_HIP_CHECK(hipGetLastError());
  f.d_mask = d_mask;
  // AdaptivelyVoxelFiltered (adaptive_voxel_filter.h:46-86), sizes compared as the reference does
  size_t in_count = 0;
  rc = voxel_filter_pass(f, 0.f, d_mask, &in_count);  // the range-filtered cloud itself
  if (rc != HG_OK) return rc;
  size_t result = in_count;
  bool have = true;  // d_idx currently holds `result`
  if (!(static_cast<float>(in_count) <= min_num_points)) {
    rc = voxel_filter_pass(f, max_length, nullptr, &result);
    if (rc != HG_OK) return rc;
    if (!(static_cast<float>(result) >= min_num_points)) {
      bool done = false;
      for (float high_length = max_length; high_length > 1e-2f * max_length && !done; high_length /= 2.f) {
        float low_length = high_length / 2.f;
        rc = voxel_filter_pass(f, low_length, nullptr, &result);
        if (rc != HG_OK) return rc;
        if (static_cast<float>(result) >= min_num_points) {
          float hl = high_length;
          float result_length = low_length;
          while ((hl - low_length) / low_length > 1e-1f) {
            const float mid_length = (low_length + hl) / 2.f;
            size_t cand = 0;
            rc = voxel_filter_pass(f, mid_length, nullptr, &cand);
            if (rc != HG_OK) return rc;
            if (static_cast<float>(cand) >= min_num_points) {
              low_length = mid_length;
              result_length = mid_length;
              result = cand;
              have = true;
            } else {
              hl = mid_length;
              have = false;  // d_idx holds the rejected candidate
            }
          }
          if (!have) {  // re-materialise the accepted result
            rc = voxel_filter_pass(f, result_length, nullptr, &result);
            if (rc != HG_OK) return rc;
          }
          done = true;
        }
      }
    }
  }
  *count = result;
  return filter_finish(f, result, indices_out);
}

int hg_filter_last_device(hg_ctx* ctx, const uint32_t** indices_dev, const float** xyz_dev, size_t* count) {
  if (!ctx) return HG_ERR_INVALID;
  if (indices_dev) *indices_dev = ctx->filter_idx;
  if (xyz_dev) *xyz_dev = ctx->filter_xyz;
  if (count) *count = ctx->filter_count;
  return HG_OK;
}

}  // extern "C"
