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

struct hg_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // insertion workspace
  hg::DeviceBuffer ws_points, ws_scan_table, ws_gate, ws_counts, ws_offsets, ws_keys_a, ws_keys_b,
      ws_vals_a, ws_vals_b, ws_temp, ws_misc;
  void* pinned = nullptr;  // small pinned host staging (4 KiB)
};

struct hg_grid {
  hg_ctx* ctx = nullptr;
  hg::GridView view{};
  uint32_t table_capacity = 0;
  float relative_truncation_distance = 0.f;
  hg_insert_stats last_stats{};
};
