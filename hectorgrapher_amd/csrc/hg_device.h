// hg_device.h — shared host/device definitions for the gfx950 TSDF path.
// Written for CDNA4 (wave64) only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hg_mi355x.h"

namespace hg {

constexpr int kWave = 64;
constexpr uint32_t kVoxelsPerBlock = 512;  // 8^3, one FlatGrid<TSDFVoxel,3> leaf (2 KiB)
constexpr int kIndexOffset = 8192;         // cell index range [-8192, 8191] (hybrid_grid_tsdf.h:58)
constexpr uint32_t kSlotPending = 0xFFFFFFu;
constexpr uint16_t kUpdateMarker = 1u << 15;

// Device view of one HybridGridTSDF. Passed to kernels by value.
//
// Block pool = dir_blocks "direct" slots followed by max_blocks overflow slots. A new block takes
// the direct slot its coordinates map to (the low bits of each block coordinate: a toroidal window
// of 2^dir_bits blocks per axis) when that slot is free, else the next overflow slot. While no block
// has gone to the overflow area and the blocks' bounding box fits the window, the address of a voxel
// is a function of its coordinates alone: lookups of the matcher then need no hash probe (one memory
// round trip instead of two, see pool_is_direct / DirectWindow). The hash table always holds every
// block and is what inserts, exports and the general lookup use.
struct GridView {
  unsigned long long* table;  // open-addressing hash: 0 = empty, else ((key + 1) << 24) | slot
  uint32_t table_mask;        // capacity - 1 (power of two)
  uint32_t max_blocks;        // blocks the grid may hold (capacity of the overflow area)
  uint32_t pool_blocks;       // dir_blocks + max_blocks slots
  uint32_t dir_blocks;        // 1 << (dir_bits[0] + dir_bits[1] + dir_bits[2])
  uint32_t dir_bits[3];       // window bits per axis (x, y, z)
  uint32_t* voxels;           // pool_blocks * 512, voxel = tsd_code | weight_code << 16
  unsigned long long* block_keys;  // per slot: key + 1 of the block it holds, 0 = free
  uint32_t* block_list;       // slots in allocation order, [0, num_blocks)
  uint32_t* counters;         // [0] num_blocks, [1] error flags, [2] hits, [4..5] updates (u64),
                              // [6] touched blocks being collected, [7] touched blocks to apply,
                              // [8] blocks in the overflow area, [9..11] min / [12..14] max block
                              // coordinate (x, y, z) over all blocks
  uint32_t* call;             // counters of the running binned insert: [0] touched blocks being collected,
                              // [1] apply work items, [2] of which slices of large bins (by default words
                              // of `counters`; a scan stream gives every scan in flight its own, as well
                              // as its own bin_count / bin_offset / touched arrays)
  uint32_t* bin_count;        // per block slot: records of the current insert call
  uint32_t* bin_offset;       // per block slot: first record of its bin
  uint32_t* touched;          // slots touched by the current insert call
  uint4* work;                // apply work items {slot, v_lo | v_hi << 16, records of the bin, 0}: per-call
  uint32_t work_capacity;     // context workspace sized from the call's record count (set by the host per call)
  float resolution;
  float max_tsd, min_tsd, max_weight;
  float tsd_resolution, weight_resolution;  // encode scales (tsd_value_converter.cc:27-28)
  float tsd_scale, tsd_offset;              // decode: code * scale + offset (value_conversion_tables.cc:35-36)
  float weight_scale, weight_offset;
};

enum : uint32_t { kFlagCapacity = 1u, kFlagRange = 2u, kFlagTime = 32u };  // (4, 8, 16: hg_insert.hip)

// Workgroups are dispatched round-robin over the 8 XCDs (workgroup b runs on XCD b % 8), and each
// XCD has its own L2. Consecutive returns of a scan touch the same voxel blocks, so workgroup b is
// given the returns of a contiguous eighth of the block per XCD instead of chunk b: the lines an
// XCD's workgroups share then live in ONE L2 instead of being replicated in all eight. A bijection
// on [0, num_wg); the partial sums keep the order of the returns, so results do not change.
__device__ inline unsigned xcd_chunk(unsigned b, unsigned num_wg) {
  const unsigned x = b & 7u, k = b >> 3;
  // workgroups with b % 8 == y: (num_wg - y + 7) / 8
  unsigned start = 0;
#pragma unroll
  for (unsigned y = 0; y < 8; ++y) start += (y < x) ? (num_wg + 7u - y) / 8u : 0u;
  return start + k;
}


// ---- codec (ref mapping/2d/tsd_value_converter.h:39-67, value_conversion_tables.cc:29-37) ----
__host__ __device__ inline float clampf(float v, float lo, float hi) {
  if (v > hi) return hi;
  if (v < lo) return lo;
  return v;
}
__device__ inline int round_to_int(float x) { return static_cast<int>(roundf(x)); }  // lround

__device__ inline uint32_t tsd_to_value(const GridView& g, float tsd) {
  return static_cast<uint32_t>(round_to_int((clampf(tsd, g.min_tsd, g.max_tsd) - g.min_tsd) *
                                            g.tsd_resolution) + 1);
}
__device__ inline uint32_t weight_to_value(const GridView& g, float w) {
  return static_cast<uint32_t>(round_to_int((clampf(w, 0.f, g.max_weight) - 0.f) *
                                            g.weight_resolution) + 1);
}
__device__ inline float value_to_tsd(const GridView& g, uint32_t code) {
  const uint32_t v = code & 0x7FFFu;
  if (v == 0) return g.min_tsd;
  return static_cast<float>(v) * g.tsd_scale + g.tsd_offset;
}
__device__ inline float value_to_weight(const GridView& g, uint32_t code) {
  const uint32_t v = code & 0x7FFFu;
  if (v == 0) return 0.f;
  return static_cast<float>(v) * g.weight_scale + g.weight_offset;
}

// ---- indexing (ref mapping/3d/hybrid_grid_base.h:40-43,428-446) ----
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

__device__ inline int cell_index_1d(float p, float resolution) { return round_to_int(p / resolution); }

// 1 / den refined as the IEEE division sequence refines it (see div_in_range in hg_insert.hip).
__device__ inline float refined_rcp(float den) {
  const float r0 = __builtin_amdgcn_rcpf(den);
  const float e0 = __builtin_fmaf(-den, r0, 1.0f);
  return __builtin_fmaf(e0, r0, r0);
}
// lround(p / res) with the division spelled out: the Newton-Raphson sequence the compiler emits for
// an IEEE fdiv, without the v_div_scale / v_div_fixup wrappers (same FMAs on the same operands, hence
// the same bits) -- those only act on operands with extreme exponents, where the index is 0 or out
// of range whichever way the last bits fall (callers reject |p| >= 1e30 beforehand). res and r are
// uniform, so the reciprocal is refined once per wavefront instead of once per division.
__device__ inline int cell_index_fast(float p, float res, float r) {
  const float q0 = p * r;
  const float rem0 = __builtin_fmaf(-res, q0, p);
  const float q1 = __builtin_fmaf(rem0, r, q0);
  const float rem1 = __builtin_fmaf(-res, q1, p);
  return round_to_int(__builtin_fmaf(rem1, r, q1));
}


// Block key of a cell index: 11 bits per axis of (index + 8192) >> 3.
__device__ inline bool cell_in_range(int x, int y, int z) {
  return (static_cast<unsigned>(x + kIndexOffset) < 16384u) &&
         (static_cast<unsigned>(y + kIndexOffset) < 16384u) &&
         (static_cast<unsigned>(z + kIndexOffset) < 16384u);
}
__device__ inline unsigned long long block_key(int x, int y, int z) {
  const unsigned long long bx = static_cast<unsigned>(x + kIndexOffset) >> 3;
  const unsigned long long by = static_cast<unsigned>(y + kIndexOffset) >> 3;
  const unsigned long long bz = static_cast<unsigned>(z + kIndexOffset) >> 3;
  return (bz << 22) | (by << 11) | bx;
}
__device__ inline uint32_t voxel_in_block(int x, int y, int z) {
  return ((static_cast<unsigned>(z + kIndexOffset) & 7u) << 6) |
         ((static_cast<unsigned>(y + kIndexOffset) & 7u) << 3) |
         (static_cast<unsigned>(x + kIndexOffset) & 7u);
}
__host__ __device__ inline void key_to_block_origin(unsigned long long key, int* x, int* y, int* z) {
  *x = (static_cast<int>(key & 2047u) << 3) - kIndexOffset;
  *y = (static_cast<int>((key >> 11) & 2047u) << 3) - kIndexOffset;
  *z = (static_cast<int>((key >> 22) & 2047u) << 3) - kIndexOffset;
}

// ---- hash table ----
// Spatial hash, separable per axis so the 8 corners of a 2x2x2 lookup share the per-axis products.
__host__ __device__ inline uint32_t hash_mix(uint32_t h) {
  h ^= h >> 15;
  h *= 0x2c1b3c6du;
  h ^= h >> 12;
  return h;
}
__host__ __device__ inline uint32_t hash_x(uint32_t bx) { return bx * 73856093u; }
__host__ __device__ inline uint32_t hash_y(uint32_t by) { return by * 19349663u; }
__host__ __device__ inline uint32_t hash_z(uint32_t bz) { return bz * 83492791u; }
__host__ __device__ inline uint32_t hash_key(unsigned long long k) {
  return hash_mix(hash_x(static_cast<uint32_t>(k) & 2047u) ^ hash_y(static_cast<uint32_t>(k >> 11) & 2047u) ^
                  hash_z(static_cast<uint32_t>(k >> 22) & 2047u));
}

// Read-only lookup. Returns the slot or 0xFFFFFFFF if the block does not exist.
__device__ inline uint32_t find_block(const GridView& g, unsigned long long key) {
  uint32_t h = hash_key(key) & g.table_mask;
  const unsigned long long tag = key + 1ull;
  for (uint32_t probe = 0; probe <= g.table_mask; ++probe) {
    const unsigned long long e = g.table[h];
    if (e == 0ull) return 0xFFFFFFFFu;
    if ((e >> 24) == tag) return static_cast<uint32_t>(e & 0xFFFFFFu);
    h = (h + 1) & g.table_mask;
  }
  return 0xFFFFFFFFu;
}

// Direct slot of a block key: the low dir_bits of each block coordinate.
__host__ __device__ inline uint32_t direct_slot(const GridView& g, unsigned long long key) {
  const uint32_t bx = static_cast<uint32_t>(key) & 2047u, by = static_cast<uint32_t>(key >> 11) & 2047u,
                 bz = static_cast<uint32_t>(key >> 22) & 2047u;
  return ((bz & ((1u << g.dir_bits[2]) - 1u)) << (g.dir_bits[0] + g.dir_bits[1])) |
         ((by & ((1u << g.dir_bits[1]) - 1u)) << g.dir_bits[0]) | (bx & ((1u << g.dir_bits[0]) - 1u));
}

// Pool slot for a new block (called by the one thread that won the block's hash entry): the direct
// slot if it is free, else the next overflow slot. 0xFFFFFFFF (capacity flag set) when the grid
// already holds max_blocks blocks.
__device__ inline uint32_t alloc_slot(const GridView& g, unsigned long long key) {
  const uint32_t n = atomicAdd(&g.counters[0], 1u);
  if (n >= g.max_blocks) {
    atomicSub(&g.counters[0], 1u);
    atomicOr(&g.counters[1], kFlagCapacity);
    return 0xFFFFFFFFu;
  }
  uint32_t slot = direct_slot(g, key);
  if (atomicCAS(&g.block_keys[slot], 0ull, key + 1ull) != 0ull) {
    slot = g.dir_blocks + atomicAdd(&g.counters[8], 1u);  // < pool_blocks: n < max_blocks
    g.block_keys[slot] = key + 1ull;
  }
  g.block_list[n] = slot;
  const uint32_t bx = static_cast<uint32_t>(key) & 2047u, by = static_cast<uint32_t>(key >> 11) & 2047u,
                 bz = static_cast<uint32_t>(key >> 22) & 2047u;
  if (bx < g.counters[9]) atomicMin(&g.counters[9], bx);
  if (by < g.counters[10]) atomicMin(&g.counters[10], by);
  if (bz < g.counters[11]) atomicMin(&g.counters[11], bz);
  if (bx > g.counters[12]) atomicMax(&g.counters[12], bx);
  if (by > g.counters[13]) atomicMax(&g.counters[13], by);
  if (bz > g.counters[14]) atomicMax(&g.counters[14], bz);
  return slot;
}

// Insert-or-get for kernels in which each distinct key is inserted by at most one thread.
// Returns the slot, or 0xFFFFFFFF when the pool/table is full (flag set).
__device__ inline uint32_t insert_block_unique(const GridView& g, unsigned long long key) {
  uint32_t h = hash_key(key) & g.table_mask;
  const unsigned long long tag = key + 1ull;
  for (uint32_t probe = 0; probe <= g.table_mask; ++probe) {
    unsigned long long e = g.table[h];
    if (e == 0ull) {
      const unsigned long long pending = (tag << 24) | kSlotPending;
      const unsigned long long prev = atomicCAS(&g.table[h], 0ull, pending);
      if (prev == 0ull) {
        const uint32_t slot = alloc_slot(g, key);
        if (slot == 0xFFFFFFFFu) return slot;  // the entry stays pending: lookups treat it as missing
        atomicExch(&g.table[h], (tag << 24) | slot);
        return slot;
      }
      e = prev;
    }
    if ((e >> 24) == tag) {
      const uint32_t s = static_cast<uint32_t>(e & 0xFFFFFFu);
      return s == kSlotPending ? 0xFFFFFFFFu : s;
    }
    h = (h + 1) & g.table_mask;
  }
  atomicOr(&g.counters[1], kFlagCapacity);
  return 0xFFFFFFFFu;
}

// Insert-or-get when several threads may insert the same key concurrently. A thread that loses
// the CAS on its own key waits for the winner to publish the slot; the winner publishes inside the
// same loop iteration it won in, so lanes of one wavefront cannot wait on each other forever.
__device__ inline uint32_t insert_block_shared(const GridView& g, unsigned long long key) {
  uint32_t h = hash_key(key) & g.table_mask;
  const unsigned long long tag = key + 1ull;
  uint32_t result = 0xFFFFFFFFu;
  bool done = false;
  uint32_t probes = 0;
  while (!done) {
    unsigned long long e = __hip_atomic_load(&g.table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (e == 0ull) {
      const unsigned long long pending = (tag << 24) | kSlotPending;
      const unsigned long long prev = atomicCAS(&g.table[h], 0ull, pending);
      if (prev == 0ull) {
        const uint32_t slot = alloc_slot(g, key);
        if (slot == 0xFFFFFFFFu) {
          // publish a poisoned entry so waiters stop: slot field stays pending but tag is cleared
          atomicExch(&g.table[h], (0x1FFFFFFFFFull << 24) | kSlotPending);
          done = true;
        } else {
          atomicExch(&g.table[h], (tag << 24) | slot);
          result = slot;
          done = true;
        }
      } else {
        e = prev;
      }
    }
    if (!done) {
      if ((e >> 24) == tag) {
        const uint32_t s = static_cast<uint32_t>(e & 0xFFFFFFu);
        if (s != kSlotPending) {
          result = s;
          done = true;
        } else {
          __builtin_amdgcn_s_sleep(1);  // winner is publishing
        }
      } else {
        h = (h + 1) & g.table_mask;
        if (++probes > g.table_mask) {
          atomicOr(&g.counters[1], kFlagCapacity);
          done = true;
        }
      }
    }
  }
  return result;
}

// Voxel fetch for lookups: unknown block -> 0 (default TSDFVoxel).
__device__ inline uint32_t load_voxel(const GridView& g, int x, int y, int z) {
  if (!cell_in_range(x, y, z)) return 0u;
  const uint32_t slot = find_block(g, block_key(x, y, z));
  if (slot >= g.pool_blocks) return 0u;
  return g.voxels[static_cast<size_t>(slot) * kVoxelsPerBlock + voxel_in_block(x, y, z)];
}

}  // namespace hg
