// hg_xray.hip — X-ray texture of a TSDF submap on the device (the view Submap3D::ToResponseProto
// serves: mapping/3d/submap_3d.cc:245-276 AddToTextureProto(HybridGridTSDF), :142-177
// ExtractVoxelData, :80-105 AccumulatePixelData, :179-214 ComputePixelValues).
//
// The reference walks the voxels in iterator order, keeps those with 1 - |tsd| / max_tsd >= 0.501,
// rounds their global position to a pixel and accumulates count / min z / max z / max probability
// and an fp32 probability SUM per pixel — the sum depends on the order. Device form:
//   k_xray_extract (count)  per-block wave walks its voxels: how many qualify, pixel bounding box
//   k_xray_extract (write)  key = pixel << 31 | iterator rank of the voxel, payload = z and value
//   rocPRIM radix sort      records of a pixel become contiguous, in iterator order
//   k_xray_pixels           the first record of each pixel walks its run sequentially (the
//                           reference's accumulation order) and writes the two texture bytes
// ProbabilityToLogOddsInteger (mapping/submaps.h:36-52) goes through the host libm's logf in the
// reference; the device uses the 254 probability thresholds at which that integer changes, found
// once on the host with the same expression, so the bytes are the host's.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <mutex>
#include <numeric>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "hg_internal.h"

namespace hg {

constexpr float kMinProbability = 0.1f;
constexpr float kMaxProbability = 1.f - kMinProbability;
constexpr int kRankBits = 31;  // sorted block index (22 bits) << 9 | voxel in block

struct XrayParams {
  float pose[7];  // global_submap_pose.cast<float>(): t xyz, q wxyz
  float resolution_inverse;
};

struct XrayState {
  int min_x, min_y, max_x, max_y;
  unsigned count;  // qualifying voxels (count pass) / write cursor (write pass)
  unsigned pad[3];
};

struct XrayThresholds {
  float t[254];  // t[k]: smallest probability whose log-odds integer is >= k + 2
};

// BoundedFloatToValue(probability, kMinProbability, kMaxProbability) (probability_values.h:32-44)
__device__ inline int probability_to_value(float probability) {
  return round_to_int((clampf(probability, kMinProbability, kMaxProbability) - kMinProbability) *
                      (32766.f / (kMaxProbability - kMinProbability))) + 1;
}

// kValueToProbability[value] (probability_values.cc:30-62), value in [1, 32767]
__device__ inline float value_to_probability(int value) {
  const float kScale = (kMaxProbability - kMinProbability) / (32768 - 2.f);
  return static_cast<float>(value) * kScale + (kMinProbability - kScale);
}

// One wave per block, blocks in iterator order. WRITE = false: count + bounding box only.
template <bool WRITE>
__global__ void k_xray_extract(GridView g, XrayParams P, const uint32_t* order, uint32_t nblocks,
                               XrayState* st, unsigned long long* keys, unsigned long long* payload) {
  const uint32_t b = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  const uint32_t lane = threadIdx.x % kWave;
  if (b >= nblocks) return;
  const uint32_t slot = order[b];
  const uint32_t* vox = g.voxels + static_cast<size_t>(slot) * kVoxelsPerBlock;
  int ox, oy, oz;
  key_to_block_origin(g.block_keys[slot] - 1ull, &ox, &oy, &oz);
  const int by0 = st->min_y, bx1 = st->max_x, by1 = st->max_y;  // WRITE: the final box
  int lmin_x = INT_MAX, lmin_y = INT_MAX, lmax_x = INT_MIN, lmax_y = INT_MIN;
  unsigned lcount = 0;
  for (int it = 0; it < 8; ++it) {
    const uint32_t idx = it * kWave + lane;
    const uint32_t v = vox[idx];
    bool keep = false;
    int px = 0, py = 0, pz = 0, value = 0;
    if (v != 0u) {  // the iterator skips default voxels (hybrid_grid_base.h:343-354)
      const float tsd = value_to_tsd(g, v & 0xFFFFu);
      const float probability = 1.f - fabsf(tsd) / g.max_tsd;
      value = probability_to_value(probability);
      if (!(probability < 0.501f)) {  // kXrayObstructedCellProbabilityLimit
        float x = static_cast<float>(ox + static_cast<int>(idx & 7u)) * g.resolution;
        float y = static_cast<float>(oy + static_cast<int>((idx >> 3) & 7u)) * g.resolution;
        float z = static_cast<float>(oz + static_cast<int>(idx >> 6)) * g.resolution;
        transform_point(P.pose, x, y, z);
        px = round_to_int(x * P.resolution_inverse);
        py = round_to_int(y * P.resolution_inverse);
        pz = round_to_int(z * P.resolution_inverse);
        keep = true;
      }
    }
    if (WRITE) {
      const unsigned long long m = __ballot(keep);
      unsigned base = 0;
      if (m) {
        if (lane == 0) base = atomicAdd(&st->count, static_cast<unsigned>(__popcll(m)));
        base = __shfl(base, 0);
      }
      if (keep) {
        const unsigned pos = base + __popcll(m & ((1ull << lane) - 1ull));
        const int width = by1 - by0 + 1;
        const unsigned long long pixel = static_cast<unsigned long long>(bx1 - px) * width + (by1 - py);
        keys[pos] = (pixel << kRankBits) | (static_cast<unsigned long long>(b) << 9) | idx;
        payload[pos] = (static_cast<unsigned long long>(static_cast<uint32_t>(pz)) << 16) |
                       static_cast<unsigned long long>(value);
      }
    } else if (keep) {
      ++lcount;
      lmin_x = min(lmin_x, px); lmax_x = max(lmax_x, px);
      lmin_y = min(lmin_y, py); lmax_y = max(lmax_y, py);
    }
  }
  if (!WRITE) {
    for (int off = 32; off > 0; off >>= 1) {
      lcount += __shfl_xor(lcount, off);
      lmin_x = min(lmin_x, __shfl_xor(lmin_x, off)); lmax_x = max(lmax_x, __shfl_xor(lmax_x, off));
      lmin_y = min(lmin_y, __shfl_xor(lmin_y, off)); lmax_y = max(lmax_y, __shfl_xor(lmax_y, off));
    }
    if (lane == 0 && lcount) {
      atomicAdd(&st->count, lcount);
      atomicMin(&st->min_x, lmin_x); atomicMax(&st->max_x, lmax_x);
      atomicMin(&st->min_y, lmin_y); atomicMax(&st->max_y, lmax_y);
    }
  }
}

// The first record of each pixel accumulates its run in iterator order (AccumulatePixelData) and
// encodes the pixel (ComputePixelValues). Pixels without records keep the zero fill.
__global__ void k_xray_pixels(const unsigned long long* keys, const unsigned long long* payload, unsigned n,
                              XrayThresholds th, uint8_t* cells) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long pixel = keys[i] >> kRankBits;
  if (i > 0 && (keys[i - 1] >> kRankBits) == pixel) return;
  int min_z = INT_MAX, max_z = INT_MIN, count = 0;
  float probability_sum = 0.f, max_probability = 0.5f;
  for (unsigned j = i; j < n && (keys[j] >> kRankBits) == pixel; ++j) {
    const unsigned long long pl = payload[j];
    const int z = static_cast<int>(static_cast<uint32_t>(pl >> 16));
    const float probability = value_to_probability(static_cast<int>(pl & 0xFFFFu));
    ++count;
    min_z = min(min_z, z);
    max_z = max(max_z, z);
    probability_sum += probability;
    max_probability = fmaxf(max_probability, probability);
  }
  const float z_difference = static_cast<float>(max_z - min_z);
  if (z_difference < 3.f) return;  // kMinZDifference: value 0, alpha 0
  const float free_space = fmaxf(z_difference - static_cast<float>(count), 0.f);
  const float free_space_weight = 0.15f * free_space;  // kFreeSpaceWeight
  const float total_weight = static_cast<float>(count) + free_space_weight;
  const float free_space_probability = 1.f - max_probability;
  const float average_probability =
      clampf((probability_sum + free_space_probability * free_space_weight) / total_weight, kMinProbability,
             kMaxProbability);
  // ProbabilityToLogOddsInteger: 1 + the number of thresholds <= average_probability
  int lo = 0, hi = 254;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (th.t[mid] <= average_probability) lo = mid + 1; else hi = mid;
  }
  const int delta = 128 - (lo + 1);
  const uint8_t alpha = delta > 0 ? 0 : static_cast<uint8_t>(-delta);
  const uint8_t value = delta > 0 ? static_cast<uint8_t>(delta) : 0;
  cells[2 * pixel] = value;
  cells[2 * pixel + 1] = (value || alpha) ? alpha : 1;
}

// ---- host: thresholds of ProbabilityToLogOddsInteger under the host libm ----------------------
static float host_logit(float probability) { return std::log(probability / (1.f - probability)); }

static int host_log_odds_integer(float probability) {
  static const float kMaxLogOdds = host_logit(kMaxProbability);
  static const float kMinLogOdds = host_logit(kMinProbability);
  return static_cast<int>(std::lround((host_logit(probability) - kMinLogOdds) * 254.f /
                                      (kMaxLogOdds - kMinLogOdds))) + 1;
}

static bool build_thresholds(XrayThresholds* out) {
  auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
  auto from = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
  const uint32_t lo_bits = bits(kMinProbability), hi_bits = bits(kMaxProbability);
  for (int k = 2; k <= 255; ++k) {
    // smallest p in [kMin, kMax] with integer >= k (positive floats order like their bit patterns)
    uint32_t lo = lo_bits, hi = hi_bits;
    if (host_log_odds_integer(from(hi)) < k) {
      out->t[k - 2] = INFINITY;
      continue;
    }
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (host_log_odds_integer(from(mid)) >= k) hi = mid; else lo = mid + 1;
    }
    // the search assumed a monotone logf: confirm it around the threshold
    for (int d = 1; d <= 16; ++d) {
      if (lo - d >= lo_bits && host_log_odds_integer(from(lo - d)) >= k) return false;
      if (lo + d <= hi_bits && host_log_odds_integer(from(lo + d)) < k) return false;
    }
    out->t[k - 2] = from(lo);
  }
  return true;
}

}  // namespace hg

using namespace hg;

extern "C" int hg_grid_xray(hg_grid* g, const double* global_submap_pose, uint8_t* cells, size_t cap,
                            int32_t* width, int32_t* height, int32_t* max_index_xy, size_t* bytes) {
  HG_REQUIRE_CTX(g);
  if (!g || !global_submap_pose || !width || !height || !max_index_xy || !bytes) return HG_ERR_INVALID;
  static XrayThresholds thresholds;
  static bool thresholds_ok = false;
  static std::once_flag once;
  std::call_once(once, [] { thresholds_ok = build_thresholds(&thresholds); });
  if (!thresholds_ok) {
    set_last_error("x-ray texture: the host logf is not monotone around a log-odds threshold");
    return HG_ERR_UNSUPPORTED;
  }
  hg_ctx* c = g->ctx;
  HG_HIP_CHECK(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  *width = *height = 0;
  max_index_xy[0] = max_index_xy[1] = 0;
  *bytes = 0;
  // blocks in iterator order (as hg_grid_export)
  std::vector<uint32_t> order;
  int rc = grid_block_order(g, &order);
  if (rc != HG_OK) return rc;
  const uint32_t nb = static_cast<uint32_t>(order.size());
  if (nb == 0) return HG_OK;
  if (nb >= (1u << 22)) return HG_ERR_CAPACITY;
  DeviceBuffer& mb = c->ws_misc;
  if ((rc = mb.reserve(nb * sizeof(uint32_t) + 256)) != HG_OK) return rc;
  XrayState* d_st = mb.as<XrayState>();
  uint32_t* d_order = reinterpret_cast<uint32_t*>(mb.as<char>() + 256);
  HG_HIP_CHECK(hipMemcpyAsync(d_order, order.data(), nb * sizeof(uint32_t), hipMemcpyHostToDevice, s));
  XrayState h_st{INT_MAX, INT_MAX, INT_MIN, INT_MIN, 0u, {0, 0, 0}};
  HG_HIP_CHECK(hipMemcpyAsync(d_st, &h_st, sizeof(h_st), hipMemcpyHostToDevice, s));
  XrayParams P;
  for (int i = 0; i < 7; ++i) P.pose[i] = static_cast<float>(global_submap_pose[i]);  // Rigid3d::cast<float>()
  P.resolution_inverse = 1.f / g->view.resolution;
  const unsigned wg = 256, per = wg / kWave, nwg = (nb + per - 1) / per;
  hipLaunchKernelGGL(k_xray_extract<false>, dim3(nwg), dim3(wg), 0, s, g->view, P, d_order, nb, d_st,
                     static_cast<unsigned long long*>(nullptr), static_cast<unsigned long long*>(nullptr));
  HG_HIP_CHECK(hipGetLastError());
  HG_HIP_CHECK(hipMemcpyAsync(&h_st, d_st, sizeof(h_st), hipMemcpyDeviceToHost, s));
  HG_HIP_CHECK(hipStreamSynchronize(s));
  if (h_st.count == 0) return HG_OK;  // the reference's bounding box is undefined without voxels
  const long long w = static_cast<long long>(h_st.max_y) - h_st.min_y + 1;
  const long long h = static_cast<long long>(h_st.max_x) - h_st.min_x + 1;
  const unsigned long long pixels = static_cast<unsigned long long>(w) * h;
  if (pixels >= (1ull << (64 - kRankBits)) || 2 * pixels > 0x7FFFFFFFull) return HG_ERR_CAPACITY;
  *width = static_cast<int32_t>(w);
  *height = static_cast<int32_t>(h);
  max_index_xy[0] = h_st.max_x;
  max_index_xy[1] = h_st.max_y;
  *bytes = static_cast<size_t>(2 * pixels);
  if (!cells) return HG_OK;
  if (cap < *bytes) return HG_ERR_CAPACITY;
  // records, sorted copies, texture
  const size_t n = h_st.count;
  const size_t rec = ((n * sizeof(unsigned long long) + 255) / 256) * 256;
  const size_t tex = ((*bytes + 255) / 256) * 256;
  DeviceBuffer& rb = c->ws_keys_b;
  if ((rc = rb.reserve(4 * rec + tex)) != HG_OK) return rc;
  char* base = rb.as<char>();
  unsigned long long* k_in = reinterpret_cast<unsigned long long*>(base);
  unsigned long long* k_out = reinterpret_cast<unsigned long long*>(base + rec);
  unsigned long long* p_in = reinterpret_cast<unsigned long long*>(base + 2 * rec);
  unsigned long long* p_out = reinterpret_cast<unsigned long long*>(base + 3 * rec);
  uint8_t* d_cells = reinterpret_cast<uint8_t*>(base + 4 * rec);
  HG_HIP_CHECK(hipMemsetAsync(&d_st->count, 0, sizeof(unsigned), s));
  HG_HIP_CHECK(hipMemsetAsync(d_cells, 0, *bytes, s));
  hipLaunchKernelGGL(k_xray_extract<true>, dim3(nwg), dim3(wg), 0, s, g->view, P, d_order, nb, d_st, k_in, p_in);
  HG_HIP_CHECK(hipGetLastError());
  unsigned end_bit = kRankBits;
  while (end_bit < 64 && (pixels >> (end_bit - kRankBits)) != 0) ++end_bit;
  size_t tb = 0;
  HG_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tb, k_in, k_out, p_in, p_out, n, 0, end_bit, s));
  if ((rc = c->ws_temp.reserve(tb)) != HG_OK) return rc;
  HG_HIP_CHECK(rocprim::radix_sort_pairs(c->ws_temp.ptr, tb, k_in, k_out, p_in, p_out, n, 0, end_bit, s));
  hipLaunchKernelGGL(k_xray_pixels, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, k_out, p_out,
                     static_cast<unsigned>(n), thresholds, d_cells);
  HG_HIP_CHECK(hipGetLastError());
  HG_HIP_CHECK(hipMemcpyAsync(cells, d_cells, *bytes, hipMemcpyDeviceToHost, s));
  HG_HIP_CHECK(hipStreamSynchronize(s));
  return HG_OK;
}
