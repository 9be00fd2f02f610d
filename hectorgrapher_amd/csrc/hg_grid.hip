// hg_grid.hip — context and HybridGridTSDF storage on the device.
//
// Storage layout (HBM): a pool of 8x8x8-voxel blocks, 2 KiB each, voxel = u32
// {u16 tsd code | u16 weight code << 16} in the reference leaf's z-major order
// (ref mapping/3d/hybrid_grid_base.h:40-43,69-141, hybrid_grid_tsdf.h:41-51),
// addressed through an open-addressing hash keyed by the block coordinate.
// The reference's DynamicGrid/NestedGrid pointer tree (:144-407) is replaced
// by this flat pool; export restores the tree's iteration order.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <mutex>
#include <utility>
#include <vector>

#include "hg_internal.h"

namespace hg {

static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }

// SetCell for m cells, one thread per cell (blocks shared by several cells are inserted once:
// insert_block_shared). Duplicate cells in one call: last writer wins in no particular order.
__global__ void k_set_cells(GridView g, const int* ijk, size_t m, const float* tsd, const float* weight) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= m) return;
  const int x = ijk[3 * i], y = ijk[3 * i + 1], z = ijk[3 * i + 2];
  if (!cell_in_range(x, y, z)) {
    atomicOr(&g.counters[1], kFlagRange);
    return;
  }
  const uint32_t slot = insert_block_shared(g, block_key(x, y, z));
  if (slot >= g.pool_blocks) return;
  // SetCell: hybrid_grid_tsdf.h:87-92
  const uint32_t code = (tsd_to_value(g, tsd[i]) + kUpdateMarker) | (weight_to_value(g, weight[i]) << 16);
  g.voxels[static_cast<size_t>(slot) * kVoxelsPerBlock + voxel_in_block(x, y, z)] = code;
}

__global__ void k_read_cells(GridView g, const int* ijk, size_t m, uint16_t* tsd, uint16_t* weight) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= m) return;
  const uint32_t v = load_voxel(g, ijk[3 * i], ijk[3 * i + 1], ijk[3 * i + 2]);
  tsd[i] = static_cast<uint16_t>(v & 0xFFFFu);
  weight[i] = static_cast<uint16_t>(v >> 16);
}

// One wave per block (in export order): number of non-default voxels.
__global__ void k_export_count(GridView g, const uint32_t* order, uint32_t nblocks, uint32_t* counts) {
  const uint32_t b = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  const uint32_t lane = threadIdx.x % kWave;
  if (b >= nblocks) return;
  const uint32_t* vox = g.voxels + static_cast<size_t>(order[b]) * kVoxelsPerBlock;
  uint32_t c = 0;
  for (int it = 0; it < 8; ++it) c += (vox[it * kWave + lane] != 0u) ? 1u : 0u;
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
  if (lane == 0) counts[b] = c;
}

// One wave per block: ordered compaction of non-default voxels to (ijk, tsd, weight).
__global__ void k_export_write(GridView g, const uint32_t* order, const uint64_t* offsets,
                               uint32_t nblocks, int* ijk, uint16_t* tsd, uint16_t* weight,
                               uint64_t cap) {
  const uint32_t b = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  const uint32_t lane = threadIdx.x % kWave;
  if (b >= nblocks) return;
  const uint32_t slot = order[b];
  const uint32_t* vox = g.voxels + static_cast<size_t>(slot) * kVoxelsPerBlock;
  int ox, oy, oz;
  key_to_block_origin(g.block_keys[slot] - 1ull, &ox, &oy, &oz);
  uint64_t base = offsets[b];
  for (int it = 0; it < 8; ++it) {
    const uint32_t idx = it * kWave + lane;
    const uint32_t v = vox[idx];
    const unsigned long long mask = __ballot(v != 0u);
    if (v != 0u) {
      const uint64_t pos = base + __popcll(mask & ((1ull << lane) - 1ull));
      if (pos < cap) {
        ijk[3 * pos] = ox + static_cast<int>(idx & 7u);
        ijk[3 * pos + 1] = oy + static_cast<int>((idx >> 3) & 7u);
        ijk[3 * pos + 2] = oz + static_cast<int>(idx >> 6);
        tsd[pos] = static_cast<uint16_t>(v & 0xFFFFu);
        weight[pos] = static_cast<uint16_t>(v >> 16);
      }
    }
    base += __popcll(mask);
  }
}

// Thread per imported block: insert key, copy 2 KiB (one wave per block).
__global__ void k_import_blocks(GridView g, const unsigned long long* keys, const uint32_t* voxels,
                                uint32_t nblocks) {
  const uint32_t b = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  const uint32_t lane = threadIdx.x % kWave;
  if (b >= nblocks) return;
  uint32_t slot = 0;
  if (lane == 0) slot = insert_block_unique(g, keys[b]);
  slot = __shfl(slot, 0);
  if (slot >= g.pool_blocks) return;
  const uint4* src = reinterpret_cast<const uint4*>(voxels + static_cast<size_t>(b) * kVoxelsPerBlock);
  uint4* dst = reinterpret_cast<uint4*>(g.voxels + static_cast<size_t>(slot) * kVoxelsPerBlock);
  dst[lane] = src[lane];
  dst[lane + kWave] = src[lane + kWave];
}

// One wave per block of the block list: key and 2 KiB of voxels into contiguous arrays (the form the
// multi-GPU gather ships and hg_grid_import_blocks takes).
__global__ void k_pack_blocks(GridView g, uint32_t nblocks, unsigned long long* keys, uint32_t* voxels) {
  const uint32_t b = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  const uint32_t lane = threadIdx.x % kWave;
  if (b >= nblocks) return;
  const uint32_t slot = g.block_list[b];
  if (lane == 0) keys[b] = g.block_keys[slot] - 1ull;
  const uint4* src = reinterpret_cast<const uint4*>(g.voxels + static_cast<size_t>(slot) * kVoxelsPerBlock);
  uint4* dst = reinterpret_cast<uint4*>(voxels + static_cast<size_t>(b) * kVoxelsPerBlock);
  dst[lane] = src[lane];
  dst[lane + kWave] = src[lane + kWave];
}

__global__ void k_list_keys(GridView g, uint32_t nblocks, unsigned long long* keys) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nblocks) keys[b] = g.block_keys[g.block_list[b]] - 1ull;
}

static uint32_t next_pow2(uint32_t v) {
  uint32_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

// The second stream of a context -- the apply stream of a pipelined scan stream, the second lane of a split
// batched solve (hg_match.hip) -- created on first use: a stream with a CU mask of all
// CUs, which gets a hardware queue of its own (see hg_ctx_create); ordinary streams may share one
// with the context's stream and then run strictly after it.
int ensure_apply_stream(hg_ctx* c) {
  if (c->apply_stream) return HG_OK;
  hipError_t e = hipErrorUnknown;
  c->apply_stream = acquire_masked_stream(c->device);
  if (c->apply_stream) {
    e = hipSuccess;
    c->apply_stream_pooled = true;
  }
  if (e != hipSuccess) e = hipStreamCreateWithFlags(&c->apply_stream, hipStreamNonBlocking);
  HG_HIP_CHECK(e);
  for (int i = 0; i < 2; ++i) {
    HG_HIP_CHECK(hipEventCreateWithFlags(&c->ev_front[i], hipEventDisableTiming));
    HG_HIP_CHECK(hipEventCreateWithFlags(&c->ev_apply[i], hipEventDisableTiming));
    HG_HIP_CHECK(hipEventCreateWithFlags(&c->ev_batch[i], hipEventDisableTiming));
  }
  return HG_OK;
}


}  // namespace hg

using namespace hg;

namespace {
std::atomic<int> g_live_contexts{0};
}
int hg::live_contexts() { return g_live_contexts.load(); }
namespace {
// CU-mask streams of destroyed contexts, kept for the next context of the same device (see hg_ctx_destroy).
std::mutex g_stream_pool_mutex;
std::vector<std::pair<int, hipStream_t>> g_stream_pool;
}  // namespace

hipStream_t hg::acquire_masked_stream(int device) {
  {
    std::lock_guard<std::mutex> lock(g_stream_pool_mutex);
    for (size_t i = 0; i < g_stream_pool.size(); ++i)
      if (g_stream_pool[i].first == device) {
        hipStream_t s = g_stream_pool[i].second;
        g_stream_pool.erase(g_stream_pool.begin() + static_cast<long>(i));
        return s;
      }
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess || prop.multiProcessorCount <= 0) return nullptr;
  std::vector<uint32_t> mask((prop.multiProcessorCount + 31) / 32, 0u);
  for (int cu = 0; cu < prop.multiProcessorCount; ++cu) mask[cu / 32] |= 1u << (cu % 32);
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, static_cast<uint32_t>(mask.size()), mask.data()) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return s;
}
void hg::release_masked_stream(int device, hipStream_t stream) {
  (void)hipStreamSynchronize(stream);
  std::lock_guard<std::mutex> lock(g_stream_pool_mutex);
  g_stream_pool.push_back({device, stream});
}

extern "C" {

const char* hg_last_error(void) { return g_last_error.c_str(); }
const char* hg_version(void) { return "hectorgrapher_amd 0.1 (gfx950)"; }

}  // extern "C"

// Slots of the grid's blocks in the reference's iterator order (hybrid_grid_base.h:304-372).
int hg::grid_block_order(hg_grid* g, std::vector<uint32_t>* order) {
  hipStream_t s = g->ctx->stream;
  uint32_t nb = 0;
  int rc = hg_grid_num_blocks(g, &nb);
  if (rc != HG_OK) return rc;
  order->clear();
  if (nb == 0) return HG_OK;
  DeviceBuffer& b = g->ctx->ws_misc;
  if ((rc = b.reserve(nb * sizeof(unsigned long long))) != HG_OK) return rc;
  hipLaunchKernelGGL(k_list_keys, dim3((nb + 255) / 256), dim3(256), 0, s, g->view, nb, b.as<unsigned long long>());
  HG_HIP_CHECK(hipGetLastError());
  std::vector<unsigned long long> keys(nb);
  std::vector<uint32_t> slots(nb);
  HG_HIP_CHECK(hipMemcpyAsync(keys.data(), b.ptr, nb * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  HG_HIP_CHECK(hipMemcpyAsync(slots.data(), g->view.block_list, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  HG_HIP_CHECK(hipStreamSynchronize(s));
  std::vector<uint32_t> idx(nb);
  std::iota(idx.begin(), idx.end(), 0u);
  std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t c) {
    return export_order_key(keys[a]) < export_order_key(keys[c]);
  });
  order->resize(nb);
  for (uint32_t i = 0; i < nb; ++i) (*order)[i] = slots[idx[i]];
  return HG_OK;
}

extern "C" {

static void prof_resolve(hg_ctx* c) {
  (void)hipStreamSynchronize(c->stream);
  for (ProfRecord& r : c->prof_records) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
      c->prof_ms[r.kernel] += ms;
      c->prof_launches[r.kernel] += r.launches;
      c->prof_units[r.kernel] += r.units;
    }
    c->prof_free_events.push_back(r.start);
    c->prof_free_events.push_back(r.stop);
  }
  c->prof_records.clear();
}

int hg_ctx_create(int device, void* stream, hg_ctx** out) {
  if (!out) return HG_ERR_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    set_last_error("no HIP device visible");
    return HG_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= count) {
    set_last_error("device index out of range");
    return HG_ERR_INVALID;
  }
  HG_HIP_CHECK(hipSetDevice(device));
  hg_ctx* c = new hg_ctx();
  c->device = device;
  for (int o = 0; o < OPT_COUNT; ++o) {  // defaults, or the environment's HG_<KEY>
    std::string env = "HG_";
    for (const char* k = kOptDesc[o].key; *k; ++k) env.push_back(static_cast<char>(std::toupper(static_cast<unsigned char>(*k))));
    const char* e = std::getenv(env.c_str());
    c->opts[o] = (e && *e) ? std::atoll(e) : kOptDesc[o].def;
  }
  (void)hipDeviceGetAttribute(&c->num_cus, hipDeviceAttributeMultiprocessorCount, device);
  g_live_contexts.fetch_add(1);
  if (stream) {
    c->stream = static_cast<hipStream_t>(stream);
  } else {
    // Contexts of one process (a host thread per trajectory, each with its context) must not share a
    // hardware queue: HIP deals its streams onto a small pool of them (GPU_MAX_HW_QUEUES, 4 by default,
    // per priority level) and two streams that land on the same one run strictly one after the other --
    // measured with four C++ host threads: two of them at half the speed of the others, 1.4x of one
    // thread in total, whichever priorities the streams had. A stream created with a CU mask gets a
    // hardware queue of its own, so every context takes one with ALL CUs enabled (four threads: 2.2x,
    // all at the same pace). HG_STREAM_PRIORITY=<n> asks for an ordinary stream of that priority
    // instead; that is also the fallback (successive contexts then take successive priority levels,
    // whose queue pools are separate).
    hipError_t e = hipErrorUnknown;
    {
      static std::atomic<int> counter{0};
      int least = 0, greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
      const int span = least - greatest + 1;
      int prio = greatest + (span > 0 ? counter++ % span : 0);
      const char* pin = std::getenv("HG_STREAM_PRIORITY");
      if (pin) prio = std::atoi(pin);
      if (!pin) {
        c->stream = acquire_masked_stream(device);
        if (c->stream) {
          e = hipSuccess;
          c->pooled_stream = true;
        }
      }
      if (e != hipSuccess) e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio);
      if (e != hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    }
    if (e != hipSuccess) {
      set_last_error(std::string("hipStreamCreate: ") + hipGetErrorString(e));
      delete c;
      return HG_ERR_HIP;
    }
    c->own_stream = true;
  }
  if (hipHostMalloc(&c->pinned, 4096, hipHostMallocMapped) != hipSuccess) c->pinned = nullptr;
  if (c->pinned) std::memset(c->pinned, 0, 4096);
  {
    void* hp = nullptr;
    void* dp = nullptr;
    if (hipHostMalloc(&hp, hg_ctx::kFlagSlots * sizeof(uint32_t), hipHostMallocMapped) == hipSuccess) {
      std::memset(hp, 0, hg_ctx::kFlagSlots * sizeof(uint32_t));
      if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
        c->flag_words = static_cast<volatile uint32_t*>(hp);
        c->async_flags = static_cast<uint32_t*>(dp);
      } else {
        (void)hipHostFree(hp);
      }
    }
    if (!c->flag_words) {
      set_last_error("hipHostMalloc: mapped page of the per-grid error flags");
      hg_ctx_destroy(c);
      return HG_ERR_HIP;
    }
  }
  *out = c;
  return HG_OK;
}

int hg_ctx_set_option(hg_ctx* c, const char* key, long long value) {
  if (!c || !key) return HG_ERR_INVALID;
  for (int o = 0; o < OPT_COUNT; ++o)
    if (std::strcmp(kOptDesc[o].key, key) == 0) {
      c->opts[o] = value;
      return HG_OK;
    }
  set_last_error(std::string("hg_ctx_set_option: unknown key ") + key);
  return HG_ERR_INVALID;
}
int hg_ctx_get_option(hg_ctx* c, const char* key, long long* value) {
  if (!c || !key || !value) return HG_ERR_INVALID;
  for (int o = 0; o < OPT_COUNT; ++o)
    if (std::strcmp(kOptDesc[o].key, key) == 0) {
      *value = c->opts[o];
      return HG_OK;
    }
  set_last_error(std::string("hg_ctx_get_option: unknown key ") + key);
  return HG_ERR_INVALID;
}

int hg_ctx_destroy(hg_ctx* c) {
  if (!c) return HG_ERR_INVALID;
  g_live_contexts.fetch_sub(1);
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  // children that outlive the context keep their device memory and lose the context (see live_grids)
  for (hg_grid* g : c->live_grids) g->ctx = nullptr;
  for (hg_problem* p : c->live_problems) orphan_problem(p);
  c->live_grids.clear();
  c->live_problems.clear();
  for (DeviceBuffer* b : {&c->ws_points, &c->ws_scan_table, &c->ws_gate, &c->ws_counts,
                          &c->ws_offsets, &c->ws_keys_a, &c->ws_keys_b, &c->ws_vals_a,
                          &c->ws_vals_b, &c->ws_temp, &c->ws_misc, &c->ws_filter, &c->ws_jobs,
                          &c->ws_keys_c, &c->ws_vals_c, &c->ws_offsets_b, &c->ws_heavy, &c->ws_heavy_list, &c->ws_sjobs, &c->ws_shadow,
                          &c->ws_unwarp, &c->ws_unwarp_tab, &c->ws_unwarp_in})
    b->release();
  if (c->copy_stream) {
    (void)hipStreamSynchronize(c->copy_stream);
    (void)hipStreamDestroy(c->copy_stream);
    for (int i = 0; i < 2; ++i) {
      if (c->ev_up[i]) (void)hipEventDestroy(c->ev_up[i]);
      if (c->ev_used[i]) (void)hipEventDestroy(c->ev_used[i]);
      c->ws_seq[i].release();
      if (c->pin_seq[i]) (void)hipHostFree(c->pin_seq[i]);
    }
  }
  if (c->apply_stream) {
    (void)hipStreamSynchronize(c->apply_stream);
    if (c->apply_stream_pooled) release_masked_stream(c->device, c->apply_stream);
    else (void)hipStreamDestroy(c->apply_stream);
    for (int i = 0; i < 2; ++i) {
      if (c->ev_front[i]) (void)hipEventDestroy(c->ev_front[i]);
      if (c->ev_apply[i]) (void)hipEventDestroy(c->ev_apply[i]);
      if (c->ev_batch[i]) (void)hipEventDestroy(c->ev_batch[i]);
    }
  }
  prof_resolve(c);
  for (hipEvent_t e : c->prof_free_events) (void)hipEventDestroy(e);
  if (c->pinned) (void)hipHostFree(c->pinned);
  if (c->flag_words) (void)hipHostFree(const_cast<uint32_t*>(c->flag_words));
  if (c->pinned_jobs) (void)hipHostFree(c->pinned_jobs);
  for (hipEvent_t e : c->ev_jobs)
    if (e) (void)hipEventDestroy(e);
  if (c->ev_lazy_batch) (void)hipEventDestroy(c->ev_lazy_batch);
  if (c->pinned_ijobs) (void)hipHostFree(c->pinned_ijobs);
  if (c->pinned_sjobs) (void)hipHostFree(c->pinned_sjobs);
  if (c->ev_sjobs) (void)hipEventDestroy(c->ev_sjobs);
  if (c->own_stream) {
    if (c->pooled_stream) {
      // A stream with a CU mask goes back to the process-wide pool instead of being destroyed: destroying one
      // shortly before the process exits deadlocks inside the runtime (ROCm 7.2: the completion thread that
      // tears the hardware queue down against the module destructor's lock; one run in six of
      // cpp/example_parity hung in exit(), none of 40 with an ordinary stream or with the stream kept).
      release_masked_stream(c->device, c->stream);
    } else {
      (void)hipStreamDestroy(c->stream);
    }
  }
  delete c;
  return HG_OK;
}

int hg_ctx_synchronize(hg_ctx* c) {
  if (!c) return HG_ERR_INVALID;
  HG_HIP_CHECK(hipStreamSynchronize(c->stream));
  return async_status(c);  // sticky errors of inserts that ran without a stats read-back
}

void* hg_ctx_stream(hg_ctx* c) { return c ? static_cast<void*>(c->stream) : nullptr; }

int hg_prof_enable(hg_ctx* c, int on) {
  if (!c) return HG_ERR_INVALID;
  // only a flag: pending event pairs are resolved (stream synchronisation) by read / reset, so
  // sampling can be switched on and off between steps without stalling the stream
  c->prof_on = on == 2 ? 2 : (on != 0 ? 1 : 0);
  return HG_OK;
}

int hg_prof_reset(hg_ctx* c) {
  if (!c) return HG_ERR_INVALID;
  prof_resolve(c);
  for (int i = 0; i < 16; ++i) {
    c->prof_ms[i] = 0;
    c->prof_launches[i] = 0;
    c->prof_units[i] = 0;
  }
  return HG_OK;
}

int hg_prof_read(hg_ctx* c, int kernel, uint64_t* launches, double* total_ms, uint64_t* units) {
  if (!c || kernel < 0 || kernel >= HG_K_COUNT) return HG_ERR_INVALID;
  prof_resolve(c);
  if (launches) *launches = c->prof_launches[kernel];
  if (total_ms) *total_ms = c->prof_ms[kernel];
  if (units) *units = c->prof_units[kernel];
  return HG_OK;
}

int hg_grid_create(hg_ctx* ctx, float resolution, float relative_truncation_distance,
                   float max_weight, uint32_t max_blocks, hg_grid** out) {
  if (!ctx || !out || !(resolution > 0.f) || !(max_weight > 0.f) ||
      !(relative_truncation_distance > 0.f) || max_blocks == 0 || max_blocks >= (1u << 22))
    return HG_ERR_INVALID;
  *out = nullptr;
  HG_HIP_CHECK(hipSetDevice(ctx->device));
  hg_grid* g = new hg_grid();
  g->ctx = ctx;
  g->relative_truncation_distance = relative_truncation_distance;
  GridView& v = g->view;
  v.max_blocks = max_blocks;
  // direct window: as many slots as the grid may hold blocks (rounded up to a power of two, at most
  // 2^21 = a 128^3-block window), z gets the smallest share of the bits
  {
    uint32_t bits = 0;
    while ((1u << bits) < max_blocks && bits < 21) ++bits;
    if (bits < 3) bits = 3;
    v.dir_bits[2] = bits / 3;
    v.dir_bits[0] = (bits - v.dir_bits[2] + 1) / 2;
    v.dir_bits[1] = bits - v.dir_bits[2] - v.dir_bits[0];
    v.dir_blocks = 1u << bits;
    v.pool_blocks = v.dir_blocks + max_blocks;
  }
  if (v.pool_blocks >= kSlotPending) {  // (unreachable: max_blocks < 2^22 gives at most 2^21 + 2^22 slots)
    delete g;
    return HG_ERR_INVALID;
  }
  // every check has passed: only now does the context learn about the grid and hand out its flag word
  if (!ctx->flag_free.empty()) {
    g->flag_slot = ctx->flag_free.back();
    ctx->flag_free.pop_back();
  } else if (ctx->flag_next < hg_ctx::kFlagSlots) {
    g->flag_slot = ctx->flag_next++;
  } else {
    set_last_error("too many grids on one context (4096 sticky-error words)");
    delete g;
    return HG_ERR_CAPACITY;
  }
  ctx->live_grids.push_back(g);
  g->table_capacity = next_pow2(std::max<uint32_t>(1024u, max_blocks * 2u));
  v.table_mask = g->table_capacity - 1;
  v.resolution = resolution;
  // HybridGridTSDF ctor (hybrid_grid_tsdf.h:61-67) + TSDValueConverter ctor
  // (tsd_value_converter.cc:22-32) + SlowValueToBoundedFloat scale.
  v.max_tsd = relative_truncation_distance * resolution;
  v.min_tsd = -v.max_tsd;
  v.max_weight = max_weight;
  v.tsd_resolution = 32766.f / (v.max_tsd - v.min_tsd);
  v.weight_resolution = 32766.f / (v.max_weight - 0.f);
  v.tsd_scale = (v.max_tsd - v.min_tsd) / 32766.f;
  v.tsd_offset = v.min_tsd - v.tsd_scale;
  v.weight_scale = (v.max_weight - 0.f) / 32766.f;
  v.weight_offset = 0.f - v.weight_scale;
  hipError_t e;
  e = hipMalloc(reinterpret_cast<void**>(&v.table), sizeof(unsigned long long) * g->table_capacity);
  if (e == hipSuccess)
    e = hipMalloc(reinterpret_cast<void**>(&v.voxels), sizeof(uint32_t) * kVoxelsPerBlock * static_cast<size_t>(v.pool_blocks));
  if (e == hipSuccess)
    e = hipMalloc(reinterpret_cast<void**>(&v.block_keys), sizeof(unsigned long long) * v.pool_blocks);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&v.counters), 256);
  // per slot: bin_count | bin_offset; per block: touched | block_list (u32 each); the apply work list
  // lives in the context workspace
  if (e == hipSuccess)
    e = hipMalloc(reinterpret_cast<void**>(&v.bin_count),
                  sizeof(uint32_t) * (2 * static_cast<size_t>(v.pool_blocks) + 2 * static_cast<size_t>(max_blocks)));
  if (e == hipSuccess) {
    v.call = v.counters + 16;
    v.bin_offset = v.bin_count + v.pool_blocks;
    v.touched = v.bin_offset + v.pool_blocks;
    v.block_list = v.touched + max_blocks;
    v.work = nullptr;
    v.work_capacity = 0;
  }
  if (e != hipSuccess) {
    set_last_error(std::string("hipMalloc grid: ") + hipGetErrorString(e));
    hg_grid_destroy(g);
    return HG_ERR_HIP;
  }
  *out = g;
  return hg_grid_clear(g);
}

int hg_grid_destroy(hg_grid* g) {
  if (!g) return HG_ERR_INVALID;
  if (g->ctx) {
    (void)hipSetDevice(g->ctx->device);
    (void)hipStreamSynchronize(g->ctx->stream);
  }
  if (g->view.table) (void)hipFree(g->view.table);
  if (g->view.voxels) (void)hipFree(g->view.voxels);
  if (g->view.block_keys) (void)hipFree(g->view.block_keys);
  if (g->view.counters) (void)hipFree(g->view.counters);
  if (g->view.bin_count) (void)hipFree(g->view.bin_count);
  g->pack.release();
  if (g->ctx) {
    if (g->flag_slot < hg_ctx::kFlagSlots) {  // the stream has drained: nothing writes the word any more
      g->ctx->flag_words[g->flag_slot] = 0u;
      g->ctx->flag_free.push_back(static_cast<uint16_t>(g->flag_slot));
    }
    auto& live = g->ctx->live_grids;
    live.erase(std::remove(live.begin(), live.end(), g), live.end());
  }
  delete g;
  return HG_OK;
}

int hg_grid_clear(hg_grid* g) {
  HG_REQUIRE_CTX(g);
  hipStream_t s = g->ctx->stream;
  const GridView& v = g->view;
  // a cleared grid starts without sticky errors: its own word only. An insertion still in flight could
  // publish into it after the host has cleared it, so a busy stream is drained first; an idle one (the usual
  // case: grids are created and cleared between maps, not between scans) costs one query. The memsets
  // below then run asynchronously.
  if (hipStreamQuery(s) != hipSuccess) {
    (void)hipGetLastError();
    HG_HIP_CHECK(hipStreamSynchronize(s));
  }
  g->ctx->flag_words[g->flag_slot] = 0u;
  HG_HIP_CHECK(hipMemsetAsync(v.table, 0, sizeof(unsigned long long) * g->table_capacity, s));
  HG_HIP_CHECK(hipMemsetAsync(v.voxels, 0, sizeof(uint32_t) * kVoxelsPerBlock * static_cast<size_t>(v.pool_blocks), s));
  HG_HIP_CHECK(hipMemsetAsync(v.block_keys, 0, sizeof(unsigned long long) * v.pool_blocks, s));
  // counters: all zero except the running minimum of the block coordinates
  static const uint32_t kInit[64] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  HG_HIP_CHECK(hipMemcpyAsync(v.counters, kInit, 256, hipMemcpyHostToDevice, s));
  HG_HIP_CHECK(hipMemsetAsync(v.bin_count, 0, sizeof(uint32_t) * 2 * static_cast<size_t>(v.pool_blocks), s));
  return HG_OK;
}

float hg_grid_resolution(const hg_grid* g) { return g ? g->view.resolution : 0.f; }

int hg_grid_params(const hg_grid* g, float* resolution, float* max_tsd, float* max_weight,
                   uint32_t* max_blocks) {
  if (!g) return HG_ERR_INVALID;
  if (resolution) *resolution = g->view.resolution;
  if (max_tsd) *max_tsd = g->view.max_tsd;
  if (max_weight) *max_weight = g->view.max_weight;
  if (max_blocks) *max_blocks = g->view.max_blocks;
  return HG_OK;
}

static int read_counters(hg_grid* g, uint32_t* out8) {
  HG_HIP_CHECK(hipMemcpyAsync(out8, g->view.counters, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost,
                              g->ctx->stream));
  HG_HIP_CHECK(hipStreamSynchronize(g->ctx->stream));
  return HG_OK;
}

int hg_grid_num_blocks(hg_grid* g, uint32_t* num_blocks) {
  HG_REQUIRE_CTX(g);
  if (!g || !num_blocks) return HG_ERR_INVALID;
  uint32_t c[8];
  int rc = read_counters(g, c);
  if (rc != HG_OK) return rc;
  *num_blocks = std::min(c[0], g->view.max_blocks);
  return HG_OK;
}

int hg_grid_window_status(hg_grid* g, uint32_t out[9]) {
  HG_REQUIRE_CTX(g);
  if (!g || !out) return HG_ERR_INVALID;
  uint32_t c[16];
  HG_HIP_CHECK(hipMemcpyAsync(c, g->view.counters, sizeof(c), hipMemcpyDeviceToHost, g->ctx->stream));
  HG_HIP_CHECK(hipStreamSynchronize(g->ctx->stream));
  out[0] = std::min(c[0], g->view.max_blocks);
  out[1] = c[8];
  bool ok = c[8] == 0;
  for (int a = 0; a < 3; ++a) {
    out[2 + a] = 1u << g->view.dir_bits[a];
    out[5 + a] = c[0] ? c[12 + a] - c[9 + a] + 1u : 0u;
    ok = ok && out[5 + a] <= out[2 + a];
  }
  out[8] = ok ? 1u : 0u;
  return HG_OK;
}

int hg_grid_set_cells(hg_grid* g, const int32_t* ijk, size_t m, const float* tsd,
                      const float* weight) {
  HG_REQUIRE_CTX(g);
  if (!g || (m && (!ijk || !tsd || !weight))) return HG_ERR_INVALID;
  if (m == 0) return HG_OK;
  hipStream_t s = g->ctx->stream;
  DeviceBuffer& b = g->ctx->ws_misc;
  const size_t bytes_ijk = m * 3 * sizeof(int), bytes_f = m * sizeof(float);
  int rc = b.reserve(bytes_ijk + 2 * bytes_f + 64);
  if (rc != HG_OK) return rc;
  char* base = b.as<char>();
  HG_HIP_CHECK(hipMemcpyAsync(base, ijk, bytes_ijk, hipMemcpyHostToDevice, s));
  HG_HIP_CHECK(hipMemcpyAsync(base + bytes_ijk, tsd, bytes_f, hipMemcpyHostToDevice, s));
  HG_HIP_CHECK(hipMemcpyAsync(base + bytes_ijk + bytes_f, weight, bytes_f, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_set_cells, dim3(static_cast<unsigned>((m + 255) / 256)), dim3(256), 0, s, g->view,
                     reinterpret_cast<const int*>(base), m,
                     reinterpret_cast<const float*>(base + bytes_ijk),
                     reinterpret_cast<const float*>(base + bytes_ijk + bytes_f));
  HG_HIP_CHECK(hipGetLastError());
  uint32_t c[8];
  rc = read_counters(g, c);
  if (rc != HG_OK) return rc;
  if (c[1] & kFlagCapacity) return HG_ERR_CAPACITY;
  if (c[1] & kFlagRange) return HG_ERR_RANGE;
  return HG_OK;
}

int hg_grid_read_cells(hg_grid* g, const int32_t* ijk, size_t m, uint16_t* tsd, uint16_t* weight) {
  HG_REQUIRE_CTX(g);
  if (!g || (m && (!ijk || !tsd || !weight))) return HG_ERR_INVALID;
  if (m == 0) return HG_OK;
  hipStream_t s = g->ctx->stream;
  DeviceBuffer& b = g->ctx->ws_misc;
  const size_t bytes_ijk = m * 3 * sizeof(int), bytes_h = ((m * sizeof(uint16_t) + 15) / 16) * 16;
  int rc = b.reserve(bytes_ijk + 2 * bytes_h + 64);
  if (rc != HG_OK) return rc;
  char* base = b.as<char>();
  HG_HIP_CHECK(hipMemcpyAsync(base, ijk, bytes_ijk, hipMemcpyHostToDevice, s));
  uint16_t* d_t = reinterpret_cast<uint16_t*>(base + bytes_ijk);
  uint16_t* d_w = reinterpret_cast<uint16_t*>(base + bytes_ijk + bytes_h);
  hipLaunchKernelGGL(k_read_cells, dim3(static_cast<unsigned>((m + 255) / 256)), dim3(256), 0, s,
                     g->view, reinterpret_cast<const int*>(base), m, d_t, d_w);
  HG_HIP_CHECK(hipGetLastError());
  HG_HIP_CHECK(hipMemcpyAsync(tsd, d_t, m * sizeof(uint16_t), hipMemcpyDeviceToHost, s));
  HG_HIP_CHECK(hipMemcpyAsync(weight, d_w, m * sizeof(uint16_t), hipMemcpyDeviceToHost, s));
  HG_HIP_CHECK(hipStreamSynchronize(s));
  return HG_OK;
}


static int export_impl(hg_grid* g, int32_t* ijk, uint16_t* tsd, uint16_t* weight, size_t cap,
                       size_t* count) {
  hipStream_t s = g->ctx->stream;
  std::vector<uint32_t> order;
  int rc = grid_block_order(g, &order);
  if (rc != HG_OK) return rc;
  const uint32_t nb = static_cast<uint32_t>(order.size());
  *count = 0;
  if (nb == 0) return HG_OK;
  DeviceBuffer& b = g->ctx->ws_misc;
  const size_t bytes_order = nb * sizeof(uint32_t);
  const size_t bytes_off = nb * sizeof(uint64_t);
  rc = b.reserve(2 * bytes_order + bytes_off + 256);
  if (rc != HG_OK) return rc;
  char* base = b.as<char>();
  uint64_t* d_off = reinterpret_cast<uint64_t*>(base);
  uint32_t* d_order = reinterpret_cast<uint32_t*>(base + bytes_off);
  uint32_t* d_counts = reinterpret_cast<uint32_t*>(base + bytes_off + bytes_order);
  HG_HIP_CHECK(hipMemcpyAsync(d_order, order.data(), bytes_order, hipMemcpyHostToDevice, s));
  const unsigned wg = 256, per = wg / kWave;
  hipLaunchKernelGGL(k_export_count, dim3((nb + per - 1) / per), dim3(wg), 0, s, g->view, d_order,
                     nb, d_counts);
  HG_HIP_CHECK(hipGetLastError());
  std::vector<uint32_t> counts(nb);
  HG_HIP_CHECK(hipMemcpyAsync(counts.data(), d_counts, bytes_order, hipMemcpyDeviceToHost, s));
  HG_HIP_CHECK(hipStreamSynchronize(s));
  std::vector<uint64_t> offsets(nb);
  uint64_t total = 0;
  for (uint32_t i = 0; i < nb; ++i) {
    offsets[i] = total;
    total += counts[i];
  }
  *count = static_cast<size_t>(total);
  if (!ijk || cap == 0 || total == 0) return HG_OK;
  const size_t n_out = std::min<size_t>(cap, total);
  DeviceBuffer& o = g->ctx->ws_temp;
  const size_t b_ijk = n_out * 3 * sizeof(int), b_h = ((n_out * sizeof(uint16_t) + 15) / 16) * 16;
  rc = o.reserve(b_ijk + 2 * b_h + 64);
  if (rc != HG_OK) return rc;
  char* ob = o.as<char>();
  HG_HIP_CHECK(hipMemcpyAsync(d_off, offsets.data(), bytes_off, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_export_write, dim3((nb + per - 1) / per), dim3(wg), 0, s, g->view, d_order,
                     d_off, nb, reinterpret_cast<int*>(ob),
                     reinterpret_cast<uint16_t*>(ob + b_ijk),
                     reinterpret_cast<uint16_t*>(ob + b_ijk + b_h), static_cast<uint64_t>(n_out));
  HG_HIP_CHECK(hipGetLastError());
  HG_HIP_CHECK(hipMemcpyAsync(ijk, ob, b_ijk, hipMemcpyDeviceToHost, s));
  HG_HIP_CHECK(hipMemcpyAsync(tsd, ob + b_ijk, n_out * sizeof(uint16_t), hipMemcpyDeviceToHost, s));
  HG_HIP_CHECK(hipMemcpyAsync(weight, ob + b_ijk + b_h, n_out * sizeof(uint16_t), hipMemcpyDeviceToHost, s));
  HG_HIP_CHECK(hipStreamSynchronize(s));
  return HG_OK;
}

int hg_grid_count(hg_grid* g, size_t* count) {
  HG_REQUIRE_CTX(g);
  if (!g || !count) return HG_ERR_INVALID;
  return export_impl(g, nullptr, nullptr, nullptr, 0, count);
}

int hg_grid_export(hg_grid* g, int32_t* ijk, uint16_t* tsd, uint16_t* weight, size_t cap,
                   size_t* count) {
  HG_REQUIRE_CTX(g);
  if (!g || !count || (cap && (!ijk || !tsd || !weight))) return HG_ERR_INVALID;
  return export_impl(g, ijk, tsd, weight, cap, count);
}

int hg_grid_block_arrays(hg_grid* g, void** keys_dev, void** voxels_dev, uint32_t* num_blocks) {
  HG_REQUIRE_CTX(g);
  if (!g || !keys_dev || !voxels_dev || !num_blocks) return HG_ERR_INVALID;
  *keys_dev = *voxels_dev = nullptr;
  uint32_t nb = 0;
  int rc = hg_grid_num_blocks(g, &nb);
  if (rc != HG_OK) return rc;
  *num_blocks = nb;
  if (nb == 0) return HG_OK;
  // packed copy in the grid's staging buffer (the pool slots of a grid's blocks are not contiguous)
  const size_t bv = static_cast<size_t>(nb) * 2048, bk = static_cast<size_t>(nb) * sizeof(unsigned long long);
  if ((rc = g->pack.reserve(bv + bk)) != HG_OK) return rc;
  uint32_t* d_vox = g->pack.as<uint32_t>();
  unsigned long long* d_keys = reinterpret_cast<unsigned long long*>(g->pack.as<char>() + bv);
  const unsigned wg = 256, per = wg / kWave;
  hipLaunchKernelGGL(k_pack_blocks, dim3((nb + per - 1) / per), dim3(wg), 0, g->ctx->stream, g->view, nb, d_keys, d_vox);
  HG_HIP_CHECK(hipGetLastError());
  HG_HIP_CHECK(hipStreamSynchronize(g->ctx->stream));
  *keys_dev = d_keys;
  *voxels_dev = d_vox;
  return HG_OK;
}

int hg_grid_import_blocks(hg_grid* g, const void* keys, const void* voxels, uint32_t nb,
                          int memspace) {
  HG_REQUIRE_CTX(g);
  if (!g || (nb && (!keys || !voxels))) return HG_ERR_INVALID;
  if (nb == 0) return HG_OK;
  hipStream_t s = g->ctx->stream;
  const unsigned long long* d_keys = static_cast<const unsigned long long*>(keys);
  const uint32_t* d_vox = static_cast<const uint32_t*>(voxels);
  if (memspace == HG_HOST) {
    DeviceBuffer& b = g->ctx->ws_temp;
    const size_t bk = nb * sizeof(unsigned long long), bv = static_cast<size_t>(nb) * 2048;
    int rc = b.reserve(bk + bv);
    if (rc != HG_OK) return rc;
    HG_HIP_CHECK(hipMemcpyAsync(b.as<char>() + bv, keys, bk, hipMemcpyHostToDevice, s));
    HG_HIP_CHECK(hipMemcpyAsync(b.as<char>(), voxels, bv, hipMemcpyHostToDevice, s));
    d_keys = reinterpret_cast<const unsigned long long*>(b.as<char>() + bv);
    d_vox = reinterpret_cast<const uint32_t*>(b.as<char>());
  }
  const unsigned wg = 256, per = wg / kWave;
  hipLaunchKernelGGL(k_import_blocks, dim3((nb + per - 1) / per), dim3(wg), 0, s, g->view, d_keys,
                     d_vox, nb);
  HG_HIP_CHECK(hipGetLastError());
  uint32_t c[8];
  int rc = read_counters(g, c);
  if (rc != HG_OK) return rc;
  if (c[1] & kFlagCapacity) return HG_ERR_CAPACITY;
  return HG_OK;
}


// ---- wire format: proto::HybridGridTSDF (mapping/proto/3d/hybrid_grid_tsdf.proto:19-31) ------
// Written exactly as HybridGridTSDF::ToProto does (hybrid_grid_tsdf.h:119-134): cells in iterator
// order, raw codes (update marker included), and — reference quirk — field 8
// "relative_truncation_distance" holds ValueConverter().getMaxTSD(), not the relative distance.
}  // extern "C"
namespace {
void put_varint(std::vector<uint8_t>& b, uint64_t v) {
  while (v >= 0x80) { b.push_back(static_cast<uint8_t>(v) | 0x80); v >>= 7; }
  b.push_back(static_cast<uint8_t>(v));
}
void put_float_field(std::vector<uint8_t>& b, int field, float f) {
  if (f == 0.f && !std::signbit(f)) return;  // proto3: default values are not serialised
  put_varint(b, (static_cast<uint64_t>(field) << 3) | 5);
  uint32_t u;
  std::memcpy(&u, &f, 4);
  for (int i = 0; i < 4; ++i) b.push_back(static_cast<uint8_t>(u >> (8 * i)));
}
template <typename It, typename F>
void put_packed(std::vector<uint8_t>& b, int field, It begin, It end, size_t stride, F enc) {
  if (begin == end) return;
  std::vector<uint8_t> body;
  for (It it = begin; it != end; it += stride) put_varint(body, enc(*it));
  put_varint(b, (static_cast<uint64_t>(field) << 3) | 2);
  put_varint(b, body.size());
  b.insert(b.end(), body.begin(), body.end());
}
bool get_varint(const uint8_t*& p, const uint8_t* end, uint64_t* v) {
  uint64_t r = 0;
  for (int shift = 0; shift < 64 && p < end; shift += 7) {
    const uint8_t c = *p++;
    r |= static_cast<uint64_t>(c & 0x7F) << shift;
    if (!(c & 0x80)) { *v = r; return true; }
  }
  return false;
}
}  // namespace

extern "C" {

int hg_grid_to_proto(hg_grid* g, uint8_t* buf, size_t cap, size_t* len) {
  HG_REQUIRE_CTX(g);
  if (!g || !len) return HG_ERR_INVALID;
  size_t n = 0;
  int rc = hg_grid_count(g, &n);
  if (rc != HG_OK) return rc;
  std::vector<int32_t> ijk(3 * std::max<size_t>(n, 1));
  std::vector<uint16_t> t(std::max<size_t>(n, 1)), w(std::max<size_t>(n, 1));
  if (n) {
    rc = hg_grid_export(g, ijk.data(), t.data(), w.data(), n, &n);
    if (rc != HG_OK) return rc;
  }
  std::vector<uint8_t> b;
  b.reserve(16 + n * 12);
  put_float_field(b, 1, g->view.resolution);
  auto zz = [](int32_t v) { return static_cast<uint64_t>((static_cast<uint32_t>(v) << 1) ^ static_cast<uint32_t>(v >> 31)); };
  auto pos = [](uint16_t v) { return static_cast<uint64_t>(v); };
  put_packed(b, 3, ijk.data(), ijk.data() + 3 * n, 3, zz);
  put_packed(b, 4, ijk.data() + 1, ijk.data() + 1 + 3 * n, 3, zz);
  put_packed(b, 5, ijk.data() + 2, ijk.data() + 2 + 3 * n, 3, zz);
  put_packed(b, 6, t.data(), t.data() + n, 1, pos);
  put_packed(b, 7, w.data(), w.data() + n, 1, pos);
  put_float_field(b, 8, g->view.max_tsd);  // sic: ToProto stores getMaxTSD() here
  put_float_field(b, 9, g->view.max_weight);
  *len = b.size();
  if (buf) {
    if (cap < b.size()) return HG_ERR_CAPACITY;
    std::memcpy(buf, b.data(), b.size());
  }
  return HG_OK;
}

// HybridGridTSDF(const proto::HybridGridTSDF&) (hybrid_grid_tsdf.h:69-83): a grid with
// (proto.resolution, proto.relative_truncation_distance, proto.max_weight) whose cells are set by
// SetCell(index, ValueToTSD(values_tsd[i]), ValueToWeight(values_weight[i])).
int hg_grid_from_proto(hg_ctx* ctx, const uint8_t* buf, size_t len, uint32_t max_blocks, hg_grid** out) {
  if (!ctx || (!buf && len) || !out) return HG_ERR_INVALID;
  float resolution = 0.f, rel_trunc = 0.f, max_weight = 0.f;
  std::vector<int32_t> xs, ys, zs;
  std::vector<uint32_t> ts, ws;
  const uint8_t* p = buf;
  const uint8_t* end = buf + len;
  while (p < end) {
    uint64_t tag;
    if (!get_varint(p, end, &tag)) return HG_ERR_INVALID;
    const int field = static_cast<int>(tag >> 3), wire = static_cast<int>(tag & 7);
    auto take = [&](uint64_t v) {
      const int32_t s32 = static_cast<int32_t>((static_cast<uint32_t>(v) >> 1) ^ (~(static_cast<uint32_t>(v) & 1) + 1));
      switch (field) {
        case 3: xs.push_back(s32); break;
        case 4: ys.push_back(s32); break;
        case 5: zs.push_back(s32); break;
        case 6: ts.push_back(static_cast<uint32_t>(v)); break;
        case 7: ws.push_back(static_cast<uint32_t>(v)); break;
        default: break;
      }
    };
    if (wire == 5) {
      if (end - p < 4) return HG_ERR_INVALID;
      uint32_t u = p[0] | (p[1] << 8) | (p[2] << 16) | (static_cast<uint32_t>(p[3]) << 24);
      p += 4;
      float f;
      std::memcpy(&f, &u, 4);
      if (field == 1) resolution = f;
      else if (field == 8) rel_trunc = f;
      else if (field == 9) max_weight = f;
    } else if (wire == 2) {
      uint64_t l;
      if (!get_varint(p, end, &l) || static_cast<uint64_t>(end - p) < l) return HG_ERR_INVALID;
      const uint8_t* q = p;
      const uint8_t* qe = p + l;
      while (q < qe) {
        uint64_t v;
        if (!get_varint(q, qe, &v)) return HG_ERR_INVALID;
        take(v);
      }
      p = qe;
    } else if (wire == 0) {
      uint64_t v;
      if (!get_varint(p, end, &v)) return HG_ERR_INVALID;
      take(v);
    } else if (wire == 1) {
      if (end - p < 8) return HG_ERR_INVALID;
      p += 8;
    } else {
      return HG_ERR_INVALID;
    }
  }
  const size_t n = ts.size();
  if (xs.size() != n || ys.size() != n || zs.size() != n || ws.size() != n) {  // the reference CHECK_EQs
    set_last_error("proto::HybridGridTSDF: index / value arrays differ in length");
    return HG_ERR_INVALID;
  }
  hg_grid* g = nullptr;
  int rc = hg_grid_create(ctx, resolution, rel_trunc, max_weight, max_blocks, &g);
  if (rc != HG_OK) return rc;
  if (n) {
    const GridView& v = g->view;
    std::vector<int32_t> ijk(3 * n);
    std::vector<float> tf(n), wf(n);
    for (size_t i = 0; i < n; ++i) {
      ijk[3 * i] = xs[i]; ijk[3 * i + 1] = ys[i]; ijk[3 * i + 2] = zs[i];
      const uint32_t tc = ts[i] & 0x7FFFu, wc = ws[i] & 0x7FFFu;  // LUT lookups of the new converter
      tf[i] = tc == 0 ? v.min_tsd : static_cast<float>(tc) * v.tsd_scale + v.tsd_offset;
      wf[i] = wc == 0 ? 0.f : static_cast<float>(wc) * v.weight_scale + v.weight_offset;
    }
    rc = hg_grid_set_cells(g, ijk.data(), n, tf.data(), wf.data());
    if (rc != HG_OK) {
      hg_grid_destroy(g);
      return rc;
    }
  }
  *out = g;
  return HG_OK;
}

}  // extern "C"
