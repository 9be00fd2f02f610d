// hg_internal.h — host-side structures behind the opaque C handles.
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "hg_device.h"

namespace hg {

void set_last_error(const std::string& msg);

#define HG_HIP_CHECK(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      ::hg::set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e));          \
      return HG_ERR_HIP;                                                                \
    }                                                                                   \
  } while (0)

// Growable device buffer owned by a context (never shrinks; no allocation in steady state).
struct DeviceBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
  int reserve(size_t need) {
    if (need <= bytes) return HG_OK;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
    size_t cap = need + need / 4 + 256;
    hipError_t e = hipMalloc(&ptr, cap);
    if (e != hipSuccess) {
      set_last_error(std::string("hipMalloc workspace: ") + hipGetErrorString(e));
      ptr = nullptr;
      return HG_ERR_HIP;
    }
    bytes = cap;
    return HG_OK;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
  }
  template <typename T>
  T* as() const { return static_cast<T*>(ptr); }
};

}  // namespace hg

namespace hg {
struct ProfRecord {
  int kernel;
  unsigned long long units;
  hipEvent_t start, stop;
  unsigned launches = 1;  // back-to-back launches bracketed by this pair
};
// Export order key of a block (ref hybrid_grid_base.h:304-372: meta cells of 64^3 voxels z-major,
// then 8^3 leaves z-major). The DynamicGrid's centring shift is a multiple of 64 cells, so the
// order is independent of how far the reference tree has grown.
inline uint64_t export_order_key(unsigned long long key) {
  const uint64_t bx = key & 2047u, by = (key >> 11) & 2047u, bz = (key >> 22) & 2047u;
  const uint64_t mx = bx >> 3, my = by >> 3, mz = bz >> 3;
  const uint64_t lx = bx & 7u, ly = by & 7u, lz = bz & 7u;
  return (((((mz << 8 | my) << 8 | mx) << 3 | lz) << 3 | ly) << 3) | lx;
}
}  // namespace hg

namespace hg {
// Per-context switches (hg_ctx_set_option); the environment variable HG_<KEY> is the default at hg_ctx_create.
enum Opt {
  OPT_PERSISTENT_SOLVE, OPT_TICKET_HANDOVER, OPT_PREPARE_KERNEL, OPT_EAGER_SOLVE, OPT_LAZY_TAIL, OPT_LM_GENERAL, OPT_LM_BAND,
  OPT_LM_BTD_GENERIC, OPT_LM_BTD_CHAIN, OPT_LM_BTD_CR, OPT_WINDOW_CAPACITY, OPT_WINDOW_TILES, OPT_BATCH_TILES,
  OPT_WINDOW_BATCH, OPT_PARTITION_MIN, OPT_PARTITION_AT, OPT_PARTITION_FOLD, OPT_HOST_TIMES, OPT_STREAM_GROUP, OPT_STREAM_SLICE,
  OPT_STREAM_MERGE, OPT_APPLY_TURNS, OPT_DEFER_LONG_CHAINS, OPT_INSERT_SORT, OPT_INSERT_PIPELINE, OPT_COUNT
};
struct OptDesc { const char* key; long long def; };
constexpr OptDesc kOptDesc[OPT_COUNT] = {
  {"persistent_solve", 0}, {"ticket_handover", 0}, {"prepare_kernel", 0}, {"eager_solve", 0}, {"lazy_tail", 3},
  {"lm_general", 0}, {"lm_band", 0}, {"lm_btd_generic", 0}, {"lm_btd_chain", 0}, {"lm_btd_cr", 0},
  {"window_capacity", 0}, {"window_tiles", 0}, {"batch_tiles", 0}, {"window_batch", 1}, {"partition_min", 32},
  {"partition_at", 2}, {"partition_fold", 1}, {"host_times", 0}, {"stream_group", 32}, {"stream_slice", 1024}, {"stream_merge", 1}, {"apply_turns", 0},
  {"defer_long_chains", 1}, {"insert_sort", 0}, {"insert_pipeline", 1},
};
int live_contexts();  // contexts of this process (hg_grid.hip)
}  // namespace hg

struct hg_problem;
struct hg_ctx {
  long long opts[hg::OPT_COUNT] = {0};
  long long opt(hg::Opt o) const { return opts[o]; }
  uint32_t stream_epoch = 0;    // groups of merged stream applies so far (claim tags, hg_insert.hip)
  bool persist_failed = false;  // a persistent solve timed out on this context: later solves take a launch per evaluation
  int persist_blocks_per_cu = -1;  // hipOccupancyMaxActiveBlocksPerMultiprocessor of the persistent kernel (-1: not asked yet)
  int num_cus = 0;
  int device = 0;
  int prof_on = 0;  // 0 off, 1 every kernel family, 2 the residual family only
  std::vector<hg::ProfRecord> prof_records;   // pending (not yet resolved)
  std::vector<hipEvent_t> prof_free_events;   // recycled events
  unsigned long long prof_launches[16] = {0};
  unsigned long long prof_units[16] = {0};
  double prof_ms[16] = {0};
  hipStream_t stream = nullptr;
  bool own_stream = false;
  bool pooled_stream = false;  // a CU-mask stream: returned to the process-wide pool, never destroyed
  // insertion workspace
  hg::DeviceBuffer ws_points, ws_scan_table, ws_gate, ws_counts, ws_offsets, ws_keys_a, ws_keys_b,
      ws_vals_a, ws_vals_b, ws_temp, ws_misc, ws_filter, ws_jobs;
  // per-point unwarping (hg_unwarp.hip): the unwarped cloud, its small tables (+ first-valid word and
  // origin slot), the upload of host-resident timed points
  hg::DeviceBuffer ws_unwarp, ws_unwarp_tab, ws_unwarp_in;
  const float* unwarp_xyz = nullptr;     // results of the last unwarp call (device)
  const float* unwarp_origin = nullptr;
  size_t unwarp_count = 0;
  const unsigned* unwarp_time_ok = nullptr;  // status word of the last unwarp call (0 = a time outside the control points)
  // second set of record / work-list buffers and the apply stream of the pipelined scan stream
  // (insert_chunk_binned with `pipe`): the front end of scan k + 1 runs next to the apply pass of scan k
  hg::DeviceBuffer ws_keys_c, ws_vals_c, ws_offsets_b;
  hg::DeviceBuffer ws_heavy, ws_heavy_list;  // deferred long chains of the binned apply pass (hg_insert.hip)
  // grouped scan stream (insert_stream_grouped): job table of the whole call (pinned staging + device
  // copy; ev_sjobs marks the staging free again) and the per-scan bin arrays of the scans in flight
  hg::DeviceBuffer ws_sjobs, ws_shadow;
  // layout of ws_shadow whose all-zero state (bin counts, call counters) the last grouped call left:
  // {ptr, bytes, call words, words per (slot, level), pool slots, levels, group}. Another layout in the
  // same buffer would alias bin counts with stale offsets / touched lists, so any change re-zeroes it.
  unsigned long long shadow_layout[7] = {0, 0, 0, 0, 0, 0, 0};
  void* pinned_sjobs = nullptr;
  size_t sjobs_capacity = 0;
  hipEvent_t ev_sjobs = nullptr;
  bool sjobs_pending = false;
  // hg_register_scan_sequence with scans in host memory: scan k + 1 travels to one of two device slots on
  // a copy stream while step k runs (ev_up: slot filled; ev_used: the step that read the slot is enqueued)
  hg::DeviceBuffer ws_seq[2];
  void* pin_seq[2] = {nullptr, nullptr};  // pinned staging of the two slots (the caller's memory is pageable)
  size_t pin_seq_bytes[2] = {0, 0};
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_used[2] = {nullptr, nullptr};
  hipStream_t apply_stream = nullptr;
  bool apply_stream_pooled = false;
  hipEvent_t ev_front[2] = {nullptr, nullptr}, ev_apply[2] = {nullptr, nullptr};
  hipEvent_t ev_batch[2] = {nullptr, nullptr};  // fork / join of a split batched solve (hg_match.hip)
  const uint32_t* filter_idx = nullptr;  // results of the last voxel-filter call (device)
  const float* filter_xyz = nullptr;
  size_t filter_count = 0;
  void* pinned = nullptr;  // small pinned host staging (4 KiB)
  void* pinned_jobs = nullptr;  // pinned staging of a batched solve's job table
  // a batched solve of general problems holds its last launches back until some problem is seen to need them
  // (hg_match.hip, batch_settle): how many iterations are left, what to launch, whose mailboxes to watch
  int lazy_batch_left = 0;
  unsigned lazy_batch_dims[3] = {0, 0, 0};            // workgroups of the plain / unwarp pass, problems
  unsigned long long lazy_batch_units[2] = {0, 0};    // profiling units of the two passes
  std::vector<const volatile unsigned long long*> lazy_batch_flags;
  std::vector<unsigned long long> lazy_batch_seqs;
  hipEvent_t ev_lazy_batch = nullptr;
  size_t jobs_capacity = 0;       // bytes of ONE slot of pinned_jobs: a ring of kJobSlots slots (jobs_staging, hg_match.hip)
  unsigned jobs_slot = 0;
  hipEvent_t ev_jobs[4] = {nullptr, nullptr, nullptr, nullptr};  // slot's copy to the device has been read
  void* pinned_ijobs = nullptr;  // pinned staging of a batched insertion's job table
  size_t ijobs_capacity = 0;
  // Sticky error flags that insert calls without a stats read-back leave for the host
  // (hg_register_scan, hg_pyramid_insert(stats = NULL)): ONE mapped pinned word per grid of the context
  // (slot = hg_grid::flag_slot), so that a full grid neither fails calls on other grids nor has its error
  // dropped when another grid is cleared. Checked by the next call that works on the grid
  // (async_status_grids) and by hg_ctx_synchronize (async_status: any grid of the context).
  static constexpr uint32_t kFlagSlots = 4096;
  volatile uint32_t* flag_words = nullptr;  // host address of the kFlagSlots words
  uint32_t* async_flags = nullptr;          // their device address
  uint32_t flag_next = 0;                   // slots [0, flag_next) have been handed out at some time
  std::vector<uint16_t> flag_free;          // released slots
  // Live grids and problems of the context. hg_ctx_destroy ORPHANS what is still alive (child->ctx =
  // nullptr) instead of leaving dangling pointers behind: a garbage-collected host (Python finalisers
  // run in any order) may destroy a grid after its context; that then only frees the grid's memory.
  std::vector<hg_grid*> live_grids;
  std::vector<hg_problem*> live_problems;
};

namespace hg {
// Brackets a launch (or library call) with events when profiling is on.
struct ProfScope {
  hg_ctx* c;
  hipEvent_t stop = nullptr;
  // `launches` back-to-back launches of the same kernel may share one event pair (every event
  // costs ~4 us of stream serialisation); `enabled` = false makes the scope a no-op.
  hipStream_t stream;
  ProfScope(hg_ctx* ctx, int kernel, unsigned long long units, unsigned launches = 1, bool enabled = true,
            hipStream_t on = nullptr)
      : c(ctx), stream(on ? on : ctx->stream) {
    if (!c->prof_on || !enabled) return;
    if (c->prof_on == 2 && kernel != HG_K_RESIDUALS) return;
    ProfRecord r;
    r.kernel = kernel;
    r.units = units;
    r.launches = launches;
    auto get = [&]() {
      hipEvent_t e = nullptr;
      if (!c->prof_free_events.empty()) {
        e = c->prof_free_events.back();
        c->prof_free_events.pop_back();
      } else {
        (void)hipEventCreate(&e);
      }
      return e;
    };
    r.start = get();
    r.stop = get();
    (void)hipEventRecord(r.start, stream);
    stop = r.stop;
    c->prof_records.push_back(r);
  }
  ~ProfScope() {
    if (stop) (void)hipEventRecord(stop, stream);
  }
};
}  // namespace hg

// A grid or problem whose context has been destroyed keeps its memory until it is destroyed itself (hg_ctx_destroy
// orphans its children), but nothing can run on it any more: every entry point refuses it.
#define HG_REQUIRE_CTX(handle)                                                              \
  do {                                                                                      \
    if (!(handle) || !(handle)->ctx) {                                                      \
      if (handle) ::hg::set_last_error("the handle's context has been destroyed");         \
      return HG_ERR_INVALID;                                                                \
    }                                                                                       \
  } while (0)

namespace hg {
int flags_to_status(uint32_t flags);
// Status of the asynchronous insert calls issued so far on this context (HG_OK or the error of a
// sticky flag that has arrived in the mailbox words).
inline int async_status(const hg_ctx* c) {
  if (!c->flag_words) return HG_OK;
  uint32_t f = 0;
  for (uint32_t i = 0; i < c->flag_next; ++i) f |= c->flag_words[i];
  return f ? flags_to_status(f) : HG_OK;
}
// The same for the grids a call works on (errors of other grids of the context do not concern it).
int async_status_grids(hg_grid* const* grids, int count);
int pyramid_insert_jobs(hg_ctx* c, int count, hg_grid* const* grids, const hg_insert_opts* opts, int levels,
                        const float* origins, const float* const* xyz, const size_t* n, size_t width,
                        const double* const* d_poses);
int pyramid_insert_impl(hg_grid* const* grids, const hg_insert_opts* opts, int levels,
                        const float* origins, const float* xyz, const uint64_t* scan_offsets,
                        size_t n_scans, size_t width, const float* poses_tq,
                        const double* d_pose_tq, int mode, int memspace, hg_insert_stats* stats,
                        const float* d_origin = nullptr);
// hg_unwarp.hip
int unwarp_enqueue(hg_ctx* c, hg_grid* const* grids, int levels, const float* points, size_t n, int memspace,
                   const hg_timed_cloud* clouds, int n_clouds, const double* control_poses,
                   const double* d_poses, const int* pose_index, int pose_stride,
                   const int64_t* control_times, int n_control, int optimized, int to_local,
                   const float* post_tq);
int unwarp_insert(hg_grid* const* grids, const hg_insert_opts* opts, int levels, const float* points, size_t n,
                  size_t width, int memspace, const hg_timed_cloud* clouds, int n_clouds,
                  const double* control_poses, const double* d_poses, const int* pose_index, int pose_stride,
                  const int64_t* control_times, int n_control, const float* pose_tq, int mode,
                  hg_insert_stats* stats);
}

namespace hg {
// Streams with a CU mask of all CUs (a hardware queue of their own, hg_ctx_create) from a process-wide pool:
// such a stream is never destroyed -- hipStreamDestroy on one shortly before exit() deadlocks inside the
// runtime (hg_ctx_destroy) -- but handed to the next context of the device. nullptr if none can be made.
hipStream_t acquire_masked_stream(int device);
int ensure_apply_stream(hg_ctx* c);  // the context's second stream (apply_stream) and its events, on first use
void release_masked_stream(int device, hipStream_t stream);
int grid_block_order(hg_grid* g, std::vector<uint32_t>* order);
void orphan_problem(hg_problem* p);  // hg_match.hip: the problem's context is going away
}

struct hg_grid {
  hg_ctx* ctx = nullptr;
  hg::GridView view{};
  hg::DeviceBuffer pack;  // packed (keys, voxels) copy handed out by hg_grid_block_arrays
  uint32_t table_capacity = 0;
  float relative_truncation_distance = 0.f;
  uint32_t flag_slot = 0xFFFFFFFFu;  // this grid's sticky-error word in the context's mapped flag page
  hg_insert_stats last_stats{};
};
