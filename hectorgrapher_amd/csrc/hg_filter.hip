// hg_filter.hip — VoxelFilter / AdaptiveVoxelFilter on the device (the step before the matching
// path: sensor/internal/voxel_filter.cc:26-69, sensor/internal/adaptive_voxel_filter.h:33-110).
//
// VoxelFilter keeps the FIRST point (input order) that falls into each voxel; cell = lround(p/res)
// per axis. Device form: a hash table keyed by the packed cell (3 x 21 bits; a pass that meets a cell
// outside that window is redone with the reference's 3 x 32-bit keys) holds the minimum
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

// ---- wide keys: 3 x 32-bit cell indices, the reference's key (voxel_filter.cc:64-69) ------------
// Taken only when a cell leaves the 3 x 21-bit window of the fast path (coordinates beyond
// 2^20 * resolution: voxel_filter_test.cc:40-48 has 1e5 m at 1 cm). A table slot is two 64-bit words,
// A = x | y << 32 and B = z | state << 32 (state 0 empty, 1 being written, 2 ready); every access
// during the insert pass is a 64-bit device-scope atomic, which is coherent across the XCDs' L2s.
struct WideKey {
  unsigned long long a;
  uint32_t z;
};
__device__ inline WideKey cell_key96(const float* p, float res) {
  // RoundToInt of a float beyond the int range is undefined in the reference too; coordinates below
  // 2^31 * resolution are exact here
  const int x = round_to_int(p[0] / res), y = round_to_int(p[1] / res), z = round_to_int(p[2] / res);
  return {static_cast<unsigned long long>(static_cast<uint32_t>(x)) |
              (static_cast<unsigned long long>(static_cast<uint32_t>(y)) << 32),
          static_cast<uint32_t>(z)};
}
__device__ inline uint32_t mix96(const WideKey& k) { return mix64(k.a ^ (0x9E3779B97F4A7C15ull * (k.z + 1ull))); }

__global__ void k_vf_insert_wide(const float* pts, unsigned n, int stride, float res, const uint8_t* mask,
                                 unsigned long long* wa, unsigned long long* wb, uint32_t* min_idx,
                                 uint32_t table_mask, uint32_t* err) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || (mask && !mask[i])) return;
  const WideKey key = cell_key96(pts + static_cast<size_t>(i) * stride, res);
  const unsigned long long ready = static_cast<unsigned long long>(key.z) | (2ull << 32);
  uint32_t h = mix96(key) & table_mask;
  uint32_t probes = 0;
  bool done = false;
  while (!done) {
    unsigned long long b = atomicAdd(&wb[h], 0ull);
    if ((b >> 32) == 0ull) {
      const unsigned long long prev = atomicCAS(&wb[h], 0ull, 1ull << 32);
      if (prev == 0ull) {  // this lane owns the slot: publish the key inside the iteration it won in
        atomicExch(&wa[h], key.a);
        __threadfence();
        atomicExch(&wb[h], ready);
        b = ready;
      } else {
        b = prev;
      }
    }
    if ((b >> 32) == 1ull) {
      __builtin_amdgcn_s_sleep(1);  // the owner is publishing
      continue;
    }
    if (b == ready && atomicAdd(&wa[h], 0ull) == key.a) {
      atomicMin(&min_idx[h], i);
      done = true;
    } else {
      h = (h + 1) & table_mask;
      if (++probes > table_mask) {
        atomicOr(err, kFlagCapacity);
        done = true;
      }
    }
  }
}

__global__ void k_vf_flags_wide(const float* pts, unsigned n, int stride, float res, const uint8_t* mask,
                                const unsigned long long* wa, const unsigned long long* wb,
                                const uint32_t* min_idx, uint32_t table_mask, uint8_t* flags) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint8_t keep = 0;
  if (!mask || mask[i]) {
    const WideKey key = cell_key96(pts + static_cast<size_t>(i) * stride, res);
    const unsigned long long ready = static_cast<unsigned long long>(key.z) | (2ull << 32);
    uint32_t h = mix96(key) & table_mask;
    for (uint32_t probe = 0; probe <= table_mask; ++probe) {
      const unsigned long long b = wb[h];
      if (b == 0ull) break;
      if (b == ready && wa[h] == key.a) {
        keep = (min_idx[h] == i) ? 1 : 0;
        break;
      }
      h = (h + 1) & table_mask;
    }
  }
  flags[i] = keep;
}

// True on the lowest active lane of every distinct key of the wavefront. Neighbouring returns of a
// scan share voxels, so letting only the leaders touch the table removes most same-address atomics.
__device__ inline bool wave_key_leader(unsigned long long key, bool active) {
  unsigned long long todo = __ballot(active);
  const unsigned lane = threadIdx.x & 63u;
  bool leader = false;
  while (todo) {
    const int l = __ffsll(static_cast<long long>(todo)) - 1;
    const unsigned lo = __builtin_amdgcn_readlane(static_cast<unsigned>(key), l);
    const unsigned hi = __builtin_amdgcn_readlane(static_cast<unsigned>(key >> 32), l);
    const unsigned long long k = (static_cast<unsigned long long>(hi) << 32) | lo;
    const unsigned long long same = __ballot(active && key == k);
    leader = leader || (lane == static_cast<unsigned>(l));
    todo &= ~same;
  }
  return leader;
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
  unsigned long long key = 0;
  bool valid = i < n && (!mask || mask[i]);
  if (valid && !cell_key21(pts + static_cast<size_t>(i) * stride, res, &key)) {
    atomicOr(err, kFlagRange);
    valid = false;
  }
  // lanes are in input order: the leader of a wavefront's key group holds its smallest index
  if (!wave_key_leader(key, valid)) return;
  uint32_t h = mix64(key) & table_mask;
  for (uint32_t probe = 0; probe <= table_mask; ++probe) {
    unsigned long long e = keys[h];
    if (e == kEmptyKey) {
      const unsigned long long prev = atomicCAS(&keys[h], kEmptyKey, key);
      e = (prev == kEmptyKey) ? key : prev;
    }
    if (e == key) {
      if (__hip_atomic_load(&min_idx[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > i) atomicMin(&min_idx[h], i);
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


__global__ void k_fill_u64(unsigned long long* p, size_t n, unsigned long long v) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

// ---- AdaptivelyVoxelFiltered without host round trips ------------------------------------------
// The reference's search (adaptive_voxel_filter.h:46-86) is a chain of VoxelFilter passes whose
// edge lengths depend only on the point counts of the passes before. Every pass is enqueued up
// front; each workgroup replays the decision chain from the counts already written, so no pass waits
// for the host. kAvfPasses bounds the chain: 1 (max_length) + 7 halvings + 4 bisection steps.
constexpr int kAvfPasses = 12;

struct AvfState {
  uint32_t in_count;            // points inside max_range
  uint32_t counts[kAvfPasses];  // voxels of pass i
  // written by k_avf_final
  float final_length;
  uint32_t final_is_mask;  // 1: the range-filtered cloud itself is the result
  uint32_t unfinished;     // chain longer than kAvfPasses (cannot happen; reported as an error)
};

struct AvfDecision {
  bool done;
  bool is_mask;
  float length;  // edge length of the next pass, or of the result when done
};

// Replays the reference control flow over the first `have` pass counts.
__device__ inline AvfDecision avf_replay(const AvfState* st, int have, float max_length, float min_num_points) {
  AvfDecision d{false, false, max_length};
  if (static_cast<float>(st->in_count) <= min_num_points) { d.done = true; d.is_mask = true; return d; }
  int i = 0;
  if (i >= have) return d;  // pass 0 runs at max_length
  if (static_cast<float>(st->counts[i++]) >= min_num_points) { d.done = true; return d; }
  float last = max_length;
  for (float high_length = max_length; high_length > 1e-2f * max_length; high_length /= 2.f) {
    float low_length = high_length / 2.f;
    d.length = low_length;
    if (i >= have) return d;
    last = low_length;
    if (static_cast<float>(st->counts[i++]) >= min_num_points) {
      float hl = high_length;
      float result_length = low_length;
      while ((hl - low_length) / low_length > 1e-1f) {
        const float mid_length = (low_length + hl) / 2.f;
        d.length = mid_length;
        if (i >= have) return d;
        if (static_cast<float>(st->counts[i++]) >= min_num_points) {
          low_length = mid_length;
          result_length = mid_length;
        } else {
          hl = mid_length;
        }
      }
      d.done = true;
      d.length = result_length;
      return d;
    }
  }
  d.done = true;  // no length reached min_num_points: the last (finest) pass is the result
  d.length = last;
  return d;
}

// Pass `pass` of the chain: counts the voxels the masked cloud occupies at the length the replay
// asks for (keys only; the first thread to claim a slot counts it).
__global__ void k_avf_count(const float* pts, unsigned n, int stride, const uint8_t* mask,
                            unsigned long long* keys, uint32_t table_mask, AvfState* st, int pass,
                            float max_length, float min_num_points, uint32_t* err) {
  __shared__ float s_len;
  __shared__ int s_run;
  __shared__ unsigned s_fresh;
  if (threadIdx.x == 0) {
    s_fresh = 0;
    const AvfDecision d = avf_replay(st, pass, max_length, min_num_points);
    s_run = d.done ? 0 : 1;
    s_len = d.length;
  }
  __syncthreads();
  if (!s_run) return;
  const float res = s_len;
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned fresh = 0;
  unsigned long long key = 0;
  bool valid = i < n && mask[i];
  if (valid && !cell_key21(pts + static_cast<size_t>(i) * stride, res, &key)) {
    atomicOr(err, kFlagRange);
    valid = false;
  }
  if (wave_key_leader(key, valid)) {
    {
      uint32_t h = mix64(key) & table_mask;
      bool placed = false;
      for (uint32_t probe = 0; probe <= table_mask; ++probe) {
        unsigned long long e = keys[h];
        if (e == kEmptyKey) {
          const unsigned long long prev = atomicCAS(&keys[h], kEmptyKey, key);
          if (prev == kEmptyKey) { fresh = 1; placed = true; break; }
          e = prev;
        }
        if (e == key) { placed = true; break; }
        h = (h + 1) & table_mask;
      }
      if (!placed) atomicOr(err, kFlagCapacity);
    }
  }
  const unsigned long long b = __ballot(fresh != 0);
  if (b && (threadIdx.x & 63u) == 0) atomicAdd(&s_fresh, static_cast<unsigned>(__popcll(b)));
  __syncthreads();
  if (threadIdx.x == 0 && s_fresh) atomicAdd(&st->counts[pass], s_fresh);
}

// FilterByMaxRange + the size of its result (one atomic per workgroup)
__global__ void k_avf_range_mask(const float* pts, unsigned n, int stride, float max_range, uint8_t* mask,
                                 AvfState* st) {
  __shared__ unsigned s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  unsigned local = 0;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float* p = pts + static_cast<size_t>(i) * stride;
    const float r = sqrtf(p[0] * p[0] + (p[1] * p[1] + p[2] * p[2]));  // Eigen norm() order
    const bool in = r <= max_range;
    mask[i] = in ? 1 : 0;
    local += in ? 1u : 0u;
  }
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
  if ((threadIdx.x & 63u) == 0 && local) atomicAdd(&s_cnt, local);
  __syncthreads();
  if (threadIdx.x == 0 && s_cnt) atomicAdd(&st->in_count, s_cnt);
}

__global__ void k_avf_final(AvfState* st, float max_length, float min_num_points) {
  const AvfDecision d = avf_replay(st, kAvfPasses, max_length, min_num_points);
  st->final_length = d.length;
  st->final_is_mask = d.is_mask ? 1u : 0u;
  st->unfinished = d.done ? 0u : 1u;
}

// The result pass at the length the chain settled on (read from the device state).
__global__ void k_avf_insert(const float* pts, unsigned n, int stride, const uint8_t* mask,
                             unsigned long long* keys, uint32_t* min_idx, uint32_t table_mask,
                             const AvfState* st, uint32_t* err) {
  if (st->final_is_mask) return;
  const float res = st->final_length;
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long key = 0;
  bool valid = i < n && mask[i];
  if (valid && !cell_key21(pts + static_cast<size_t>(i) * stride, res, &key)) {
    atomicOr(err, kFlagRange);
    valid = false;
  }
  // lanes are in input order, so the leader of a wavefront's key group holds its smallest index
  if (!wave_key_leader(key, valid)) return;
  uint32_t h = mix64(key) & table_mask;
  for (uint32_t probe = 0; probe <= table_mask; ++probe) {
    unsigned long long e = keys[h];
    if (e == kEmptyKey) {
      const unsigned long long prev = atomicCAS(&keys[h], kEmptyKey, key);
      e = (prev == kEmptyKey) ? key : prev;
    }
    if (e == key) {
      if (__hip_atomic_load(&min_idx[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > i) atomicMin(&min_idx[h], i);
      return;
    }
    h = (h + 1) & table_mask;
  }
  atomicOr(err, kFlagCapacity);
}

__global__ void k_avf_flags(const float* pts, unsigned n, int stride, const uint8_t* mask,
                            const unsigned long long* keys, const uint32_t* min_idx,
                            uint32_t table_mask, const AvfState* st, uint8_t* flags) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (st->final_is_mask) { flags[i] = mask[i]; return; }
  const float res = st->final_length;
  uint8_t keep = 0;
  unsigned long long key;
  if (mask[i] && cell_key21(pts + static_cast<size_t>(i) * stride, res, &key)) {
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
    {
      // a cell outside the 3 x 21-bit window: redo the pass with the reference's 3 x 32-bit keys
      uint32_t flags_now[2];
      HG_HIP_CHECK(hipMemcpyAsync(flags_now, f.d_count, sizeof(flags_now), hipMemcpyDeviceToHost, s));
      HG_HIP_CHECK(hipStreamSynchronize(s));
      if (flags_now[1] & kFlagRange) {
        int rc = f.c->ws_keys_b.reserve(cap * sizeof(unsigned long long));
        if (rc != HG_OK) return rc;
        unsigned long long* wb = f.c->ws_keys_b.as<unsigned long long>();
        HG_HIP_CHECK(hipMemsetAsync(f.d_count + 1, 0, sizeof(uint32_t), s));
        HG_HIP_CHECK(hipMemsetAsync(f.d_keys, 0, cap * sizeof(unsigned long long), s));
        HG_HIP_CHECK(hipMemsetAsync(wb, 0, cap * sizeof(unsigned long long), s));
        HG_HIP_CHECK(hipMemsetAsync(f.d_min, 0xFF, cap * sizeof(uint32_t), s));
        hipLaunchKernelGGL(k_vf_insert_wide, dim3(nwg), dim3(wg), 0, s, f.d_pts, f.n, f.stride, res, f.d_mask,
                           f.d_keys, wb, f.d_min, f.table_mask, f.d_count + 1);
        HG_HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(k_vf_flags_wide, dim3(nwg), dim3(wg), 0, s, f.d_pts, f.n, f.stride, res, f.d_mask,
                           f.d_keys, wb, f.d_min, f.table_mask, f.d_flags);
        HG_HIP_CHECK(hipGetLastError());
      }
    }
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
  const unsigned wg = 256, nwg = (f.n + wg - 1) / wg;
  // one key table per pass of the chain + the result pass, cleared by a single memset
  const size_t cap = static_cast<size_t>(f.table_mask) + 1;
  if ((rc = ctx->ws_keys_a.reserve((kAvfPasses + 1) * cap * sizeof(unsigned long long))) != HG_OK) return rc;
  if ((rc = ctx->ws_misc.reserve(sizeof(AvfState))) != HG_OK) return rc;
  unsigned long long* d_keys = ctx->ws_keys_a.as<unsigned long long>();
  AvfState* d_st = ctx->ws_misc.as<AvfState>();
  HG_HIP_CHECK(hipMemsetAsync(d_st, 0, sizeof(AvfState), s));
  hipLaunchKernelGGL(k_fill_u64, dim3(2048), dim3(256), 0, s, d_keys, (kAvfPasses + 1) * cap, kEmptyKey);
  HG_HIP_CHECK(hipMemsetAsync(f.d_min, 0xFF, cap * sizeof(uint32_t), s));
  hipLaunchKernelGGL(k_avf_range_mask, dim3(std::min(nwg, 256u)), dim3(wg), 0, s, f.d_pts, f.n, f.stride,
                     max_range, d_mask, d_st);
  for (int pass = 0; pass < kAvfPasses; ++pass)
    hipLaunchKernelGGL(k_avf_count, dim3(nwg), dim3(wg), 0, s, f.d_pts, f.n, f.stride, d_mask,
                       d_keys + static_cast<size_t>(pass) * cap, f.table_mask, d_st, pass, max_length,
                       min_num_points, f.d_count + 1);
  hipLaunchKernelGGL(k_avf_final, dim3(1), dim3(1), 0, s, d_st, max_length, min_num_points);
  unsigned long long* d_keys_final = d_keys + static_cast<size_t>(kAvfPasses) * cap;
  hipLaunchKernelGGL(k_avf_insert, dim3(nwg), dim3(wg), 0, s, f.d_pts, f.n, f.stride, d_mask, d_keys_final,
                     f.d_min, f.table_mask, d_st, f.d_count + 1);
  hipLaunchKernelGGL(k_avf_flags, dim3(nwg), dim3(wg), 0, s, f.d_pts, f.n, f.stride, d_mask, d_keys_final,
                     f.d_min, f.table_mask, d_st, f.d_flags);
  HG_HIP_CHECK(hipGetLastError());
  size_t tb = 0;
  HG_HIP_CHECK(rocprim::select(nullptr, tb, rocprim::counting_iterator<uint32_t>(0), f.d_flags, f.d_idx,
                               f.d_count, f.n, s));
  if ((rc = ctx->ws_temp.reserve(tb)) != HG_OK) return rc;
  HG_HIP_CHECK(rocprim::select(ctx->ws_temp.ptr, tb, rocprim::counting_iterator<uint32_t>(0), f.d_flags,
                               f.d_idx, f.d_count, f.n, s));
  uint32_t h[2];
  AvfState h_st;
  HG_HIP_CHECK(hipMemcpyAsync(h, f.d_count, sizeof(h), hipMemcpyDeviceToHost, s));
  HG_HIP_CHECK(hipMemcpyAsync(&h_st, d_st, sizeof(h_st), hipMemcpyDeviceToHost, s));
  HG_HIP_CHECK(hipStreamSynchronize(s));
  if (h[1] & kFlagRange) {
    set_last_error("voxel filter: cell index outside +-2^20 (device key is 3 x 21 bits)");
    return HG_ERR_RANGE;
  }
  if (h[1] & kFlagCapacity) return HG_ERR_CAPACITY;
  if (h_st.unfinished) {
    set_last_error("adaptive voxel filter: search chain longer than the enqueued passes");
    return HG_ERR_CAPACITY;
  }
  *count = h[0];
  return filter_finish(f, h[0], indices_out);
}

int hg_filter_last_device(hg_ctx* ctx, const uint32_t** indices_dev, const float** xyz_dev, size_t* count) {
  if (!ctx) return HG_ERR_INVALID;
  if (indices_dev) *indices_dev = ctx->filter_idx;
  if (xyz_dev) *xyz_dev = ctx->filter_xyz;
  if (count) *count = ctx->filter_count;
  return HG_OK;
}

}  // extern "C"
