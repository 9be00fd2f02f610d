// hg_chain.h -- the exact per-voxel update chain of the binned insertion (UpdateCell with unit update weight,
// mapping/3d/tsdf_range_data_inserter_3d.cc:725-737 + SetCell, hybrid_grid_tsdf.h:87-92, on raw codes), and its
// segmented evaluation by a whole workgroup (seg_chains). Device code only; included by hg_insert.hip and by the
// microbenchmark scripts/seg_chain_bench.hip.
#pragma once

#include "hg_device.h"

namespace hg {

// lround(t) for 0 <= t < 2^23 without the generic half-away-from-zero sequence: trunc is exact,
// the fraction t - trunc(t) is exact, ties (>= 0.5) go up. Returns the rounded value as float.
__device__ inline float round_nonneg(float t) {
  const float r = truncf(t);
  return (t - r >= 0.5f) ? r + 1.0f : r;
}

// lround(t) + 1 for t = v * resolution >= 0, given y = v * (2 * resolution) = 2t (scaling by two
// commutes with the rounding of the product): floor(y) = 2n + [frac(t) >= 0.5] for t = n + frac, so
// floor((floor(y) + 1) / 2) = lround(t); every step is exact in fp32 for t < 2^22. Four
// instructions instead of the six of trunc / subtract / compare / select / add.
__device__ inline float round_nonneg_plus1(float y) {
  return floorf(__builtin_fmaf(floorf(y), 0.5f, 1.5f));
}

// num / den, correctly rounded, for the operands of the unit-weight update chain: den = w + 1 in
// [1, maximum_weight + 1], |num| <= (|tsd| * w + |update|) — far inside the range where
// v_div_scale_f32 leaves both operands unscaled and v_div_fixup_f32 has nothing to fix. This is the
// Newton-Raphson sequence the compiler emits for an IEEE fdiv without those two wrappers (same
// instructions, same operands, hence the same bits); only the sign of a zero quotient can differ,
// which the quantisation that follows does not see. The reciprocal depends on the weight chain
// only, so the dependent chain through the TSD value is 5 FMAs instead of 10 instructions.
// The codec constants of a grid, by value: what the chain needs of a GridView (the segmented evaluation is a
// function of its own -- not inlined -- and must not drag the whole view along).
struct ChainCodec {
  float min_tsd, max_tsd, max_weight, tsd_resolution, weight_resolution, tsd_scale, tsd_offset, weight_scale, weight_offset;
};
template <typename G>
__device__ inline ChainCodec chain_codec(const G& g) {
  return ChainCodec{g.min_tsd, g.max_tsd, g.max_weight, g.tsd_resolution, g.weight_resolution,
                    g.tsd_scale, g.tsd_offset, g.weight_scale, g.weight_offset};
}
template <typename G>
__device__ inline bool div_in_range_ok(const G& g) {
  return g.max_weight <= 1.0e6f && g.max_tsd <= 1.0e3f && g.min_tsd >= -1.0e3f;
}
__device__ inline float div_in_range(float num, float den) {
  const float r0 = __builtin_amdgcn_rcpf(den);
  const float e0 = __builtin_fmaf(-den, r0, 1.0f);
  const float r = __builtin_fmaf(e0, r0, r0);
  const float q0 = num * r;
  const float rem0 = __builtin_fmaf(-den, q0, num);
  const float q1 = __builtin_fmaf(rem0, r, q0);
  const float rem1 = __builtin_fmaf(-den, q1, num);
  return __builtin_fmaf(rem1, r, q1);
}

// Incremental form of the same chain: begin(code) ... step(u) ... end() == update_cell in a loop
// with update weight 1.
struct UnitChain {
  float d, w, rt, rw;
  uint32_t code0;
  bool any, fast;
  __device__ inline void begin(const GridView& g, uint32_t code) {
    fast = div_in_range_ok(g);
    const uint32_t tc = code & 0x7FFFu, wc = (code >> 16) & 0x7FFFu;
    d = tc == 0 ? g.min_tsd : static_cast<float>(tc) * g.tsd_scale + g.tsd_offset;
    w = wc == 0 ? 0.f : static_cast<float>(wc) * g.weight_scale + g.weight_offset;
    rt = rw = 0.f;
    code0 = code;
    any = false;
  }
  template <bool FAST>
  __device__ inline void step_t(const GridView& g, float maximum_weight, float u) {
    float uw = w + 1.0f;
    const float ud = FAST ? div_in_range(d * w + u, uw) : (d * w + u) / uw;
    uw = (maximum_weight < uw) ? maximum_weight : uw;
    rt = round_nonneg_plus1((__builtin_amdgcn_fmed3f(ud, g.min_tsd, g.max_tsd) - g.min_tsd) * (2.0f * g.tsd_resolution));
    rw = round_nonneg_plus1((__builtin_amdgcn_fmed3f(uw, 0.f, g.max_weight) - 0.f) * (2.0f * g.weight_resolution));
    d = rt * g.tsd_scale + g.tsd_offset;  // rt, rw hold code = lround(..) + 1
    w = rw * g.weight_scale + g.weight_offset;
    any = true;
  }
  __device__ inline void step(const GridView& g, float maximum_weight, float u) {
    if (fast) step_t<true>(g, maximum_weight, u); else step_t<false>(g, maximum_weight, u);
  }
  // `count` consecutive steps on LDS values (blocked prefetch, see chain_run below)
  __device__ inline void run(const GridView& g, float maximum_weight, const uint32_t* vals, unsigned count);
  __device__ inline uint32_t end() const {
    if (!any) return code0;
    const uint32_t nt = static_cast<uint32_t>(static_cast<int>(rt));
    const uint32_t nw = static_cast<uint32_t>(static_cast<int>(rw));
    return (nt + kUpdateMarker) | (nw << 16);
  }
};

// `count` consecutive UpdateCell calls with update weight 1 on one voxel (values vals[0..count)),
// bit-identical to calling update_cell in a loop: codes stay in float form (code - 1 as a float)
// between updates. One update is a chain of 15 dependent fp32 operations through the TSD value
// (about 7 cycles each for a wavefront on its own); everything else has to stay off that chain:
//   * the values come from LDS four at a time, one block AHEAD of their use (a read issued and
//     awaited inside an update exposes an LDS round trip, which used to double the time per update);
//   * the weight follows its own recurrence, which does not depend on the TSD value; once it has
//     reached its fixed point (the clamp at maximum_weight: the heavy voxels next to the sensor sit
//     there from their second scan on) the weight arithmetic is skipped for as long as every lane of
//     the wavefront that still has updates is there too.
// (Round 4, measured and dropped: a double-precision shortcut for the tail of a pass, where one or a few voxels
// next to the sensor still have thousands of updates. With the weight at its fixed point the new code is
// floor(A c + Ku u + B) in real arithmetic -- one f64 fma and one floor on the dependent chain -- and equals the
// reference's code unless that value lies within E = 2^-24 res (4 T + 3 R) of an integer (the six fp32 roundings of
// the exact chain; about one update in forty), near the clamps, or u is out of range, in which case the block of
// four is redone exactly. Bit-exact in every test, but the guards make the block as many instructions as the 64
// dependent fp32 operations it replaces, the fall-backs come on top, and the constants cost the apply kernels
// scratch at their 64 registers: exact stream B = 32 18.4k -> 15.0k scans/s, B = 64 9.9k -> 7.8k. The chain stays
// as it is: 16 dependent operations per update.)
struct ChainState {
  float d, w;    // decoded TSD value and weight
  float rt, rw;  // their codes (lround(..) + 1) as floats, valid after the first update
  bool fixed;    // the weight no longer changes under updates
};
template <bool FAST, typename G>
__device__ inline void chain_run(const G& g, float maximum_weight, ChainState& st, const uint32_t* vals,
                                 unsigned count) {
  float d = st.d, w = st.w, rt = st.rt, rw = st.rw;
  bool fixed = st.fixed;
  const float res2_t = 2.0f * g.tsd_resolution, res2_w = 2.0f * g.weight_resolution;
  auto step = [&](float u) {
    float uw = w + 1.0f;
    const float ud = FAST ? div_in_range(d * w + u, uw) : (d * w + u) / uw;  // u * 1.0f == u
    uw = (maximum_weight < uw) ? maximum_weight : uw;
    // TSDToValue / WeightToValue (values are finite: med3 == the reference's two-sided clamp);
    // rt, rw hold the codes lround(..) + 1 as floats
    rt = round_nonneg_plus1((__builtin_amdgcn_fmed3f(ud, g.min_tsd, g.max_tsd) - g.min_tsd) * res2_t);
    rw = round_nonneg_plus1((__builtin_amdgcn_fmed3f(uw, 0.f, g.max_weight) - 0.f) * res2_w);
    // ValueToTSD / ValueToWeight of the codes (never 0)
    d = rt * g.tsd_scale + g.tsd_offset;
    const float wn = rw * g.weight_scale + g.weight_offset;
    fixed = wn == w;
    w = wn;
  };
  constexpr unsigned K = 4;  // a block of 4 updates (~450 cycles) covers the LDS latency; 8 costs a workgroup per CU in registers
  unsigned base = 0;
  if (count >= K) {
    uint32_t cur[K], nx[K];
#pragma unroll
    for (unsigned k = 0; k < K; ++k) cur[k] = vals[k];
    while (base + K <= count) {
      const unsigned nb = base + K;
#pragma unroll
      for (unsigned k = 0; k < K; ++k) nx[k] = vals[min(nb + k, count - 1u)];  // in flight during this block
      if (__all(fixed)) {
        // fixed weight: the update is d <- quantise((d * w + u) / (w + 1)) with constants w, 1 / (w + 1)
        const float uw = w + 1.0f;
        const float r0 = __builtin_amdgcn_rcpf(uw);
        const float r = __builtin_fmaf(__builtin_fmaf(-uw, r0, 1.0f), r0, r0);
#pragma unroll
        for (unsigned k = 0; k < K; ++k) {
          const float num = d * w + __uint_as_float(cur[k]);
          float ud;
          if (FAST) {  // div_in_range with the reciprocal hoisted
            const float q0 = num * r;
            const float q1 = __builtin_fmaf(__builtin_fmaf(-uw, q0, num), r, q0);
            ud = __builtin_fmaf(__builtin_fmaf(-uw, q1, num), r, q1);
          } else {
            ud = num / uw;
          }
          rt = round_nonneg_plus1((__builtin_amdgcn_fmed3f(ud, g.min_tsd, g.max_tsd) - g.min_tsd) * res2_t);
          d = rt * g.tsd_scale + g.tsd_offset;
        }
      } else {
#pragma unroll
        for (unsigned k = 0; k < K; ++k) step(__uint_as_float(cur[k]));
      }
#pragma unroll
      for (unsigned k = 0; k < K; ++k) cur[k] = nx[k];
      base = nb;
    }
  }
  if (base < count) {
    uint32_t next = vals[base];
    for (unsigned j = base; j < count; ++j) {
      const float u = __uint_as_float(next);
      if (j + 1 < count) next = vals[j + 1];
      step(u);
    }
  }
  st.d = d; st.w = w; st.rt = rt; st.rw = rw; st.fixed = fixed;
}
template <bool FAST, typename G>
__device__ inline uint32_t update_chain_unit_t(const G& g, float maximum_weight, uint32_t code,
                                               const uint32_t* vals, unsigned count) {
  const uint32_t tc = code & 0x7FFFu, wc = (code >> 16) & 0x7FFFu;
  ChainState st;
  st.d = tc == 0 ? g.min_tsd : static_cast<float>(tc) * g.tsd_scale + g.tsd_offset;
  st.w = wc == 0 ? 0.f : static_cast<float>(wc) * g.weight_scale + g.weight_offset;
  st.rt = st.rw = 0.f;
  st.fixed = false;
  chain_run<FAST>(g, maximum_weight, st, vals, count);
  const uint32_t nt = static_cast<uint32_t>(static_cast<int>(st.rt));
  const uint32_t nw = static_cast<uint32_t>(static_cast<int>(st.rw));
  return (nt + kUpdateMarker) | (nw << 16);
}
__device__ inline void UnitChain::run(const GridView& g, float maximum_weight, const uint32_t* vals, unsigned count) {
  if (count == 0) return;
  ChainState st{d, w, rt, rw, false};
  if (fast) chain_run<true>(g, maximum_weight, st, vals, count);
  else chain_run<false>(g, maximum_weight, st, vals, count);
  d = st.d; w = st.w; rt = st.rt; rw = st.rw;
  any = true;
}
__device__ inline uint32_t update_chain_unit(const GridView& g, float maximum_weight, uint32_t code,
                                             const uint32_t* vals, unsigned count) {
  if (count == 0) return code;
  return div_in_range_ok(g) ? update_chain_unit_t<true>(g, maximum_weight, code, vals, count)
                            : update_chain_unit_t<false>(g, maximum_weight, code, vals, count);
}

// ==========================================================================================================
// Segmented evaluation of long chains (round 5).
//
// A voxel's n updates are a sequential chain c_{k+1} = F(c_k, w_k, u_k) on its 15-bit TSD code: 16 dependent fp32
// operations per update, one active lane, 0.061 us per update -- next to a wall one voxel takes thousands of updates
// per scan and its chain IS the apply launch. Shortening a step was exhausted in rounds 2-4; this cuts the LENGTH:
//   * the weight does not depend on the TSD value, and under unit updates its code walks a known orbit
//     (c + round(weight_resolution) per update until the clamp), so the weight at any position of the chain is
//     known up front -- and VERIFIED: every segment runs the exact weight recurrence from its assumed start and
//     compares its end with the next segment's assumed start;
//   * the TSD code at the start of a segment is not known, but it is close to the value the chain WITHOUT
//     re-quantisation reaches there (an affine recurrence d <- a_k d + b_k, a_k = w_k / (w_k + 1), composed per
//     segment by a lane-parallel fp64 reduction): the re-quantisation noise is a random walk damped by a_k, standard
//     deviation 0.29 sqrt((w + 1) / 2) LSB <= 6.5 LSB at weight 1000;
//   * so a chain is cut into segments, one wavefront each, and the 64 lanes of the wavefront run the EXISTING exact
//     chain (chain_run) over the segment from the 64 candidate start codes p - 32 ... p + 31 around the prediction p
//     (all lanes read the same LDS values: broadcasts); the end codes form a table per segment;
//   * one lane then walks the tables from the voxel's true code: c <- T_s[c - (p_s - 32)]. A code outside a
//     table's window (a miss: |c - p_s| > 32, i.e. beyond 4.9 standard deviations) runs that segment
//     sequentially from c; a failed weight check runs the whole chain sequentially. Results never depend on the
//     prediction or on the closed form of the weight: both only decide how much work is wasted.
// A wavefront costs the same with 1 or 64 active lanes, so the candidates are free; what the scheme spends is the
// other seven wavefronts of the workgroup, which the one-lane chain left idle. Time for n updates: n / 8 steps plus
// about 1.5 us of set-up (list, prediction, three barriers, walk) against n steps.
// Up to kSegUnits (voxel, segment) units share a round: the heavy voxels of a pass get segments in proportion to
// their lengths (seg_assign), so several heavy voxels cost sum(n) / 8, not max(n).
// ==========================================================================================================
#ifdef HG_SEG_STATS
__device__ long long g_seg_stamps[8];
__device__ unsigned long long g_seg_longest;  // (diagnostics: the phases of the longest call so far are kept)
#define SEG_STAMP(i) do { if (tid == 0) seg_ts[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
__device__ unsigned g_seg_stats[8];  // [0] table lookups, [1] misses, [2] failed weight checks, [3] max |c - p|, [4] sum |c - p|
#else
#define SEG_STAMP(i)
#endif
constexpr unsigned kSegUnits = 8;    // = wavefronts of the apply workgroup: one (voxel, segment) unit each
constexpr int kSegHalf = 32;         // candidates p - 32 ... p + 31
// LDS scratch of seg_chains, in 32-bit words:
constexpr unsigned kSegTab = 0;                                  // kSegUnits x 64 end codes
constexpr unsigned kSegAB = kSegTab + kSegUnits * kWave;         // kSegUnits x (A, B) floats
constexpr unsigned kSegLo = kSegAB + kSegUnits * 2;              // kSegUnits window bases p - 32 (int)
constexpr unsigned kSegBad = kSegLo + kSegUnits;                 // per voxel: the weight check failed
constexpr unsigned kSegFrom = kSegBad + kSegUnits;               // per voxel: first segment the running round evaluates,
constexpr unsigned kSegCur = kSegFrom + kSegUnits;               //   the voxel's exact TSD code at the start of that segment,
constexpr unsigned kSegDone = kSegCur + kSegUnits;               //   the voxel has its result
constexpr unsigned kSegAgain = kSegDone + kSegUnits;             // some voxel needs the second round
constexpr unsigned kSegWords = kSegAgain + 1u;
// the list of the round's heavy voxels (filled by the threads that own them), in words of `list`:
constexpr unsigned kSegB0 = 0;                                   // first value of the voxel in `vals`,
constexpr unsigned kSegCnt = kSegB0 + kSegUnits;                 // number of updates,
constexpr unsigned kSegVox = kSegCnt + kSegUnits;                // voxel inside the block,
constexpr unsigned kSegCode = kSegVox + kSegUnits;               // its code before the updates (afterwards: the new code)
constexpr unsigned kSegListWords = kSegCode + kSegUnits;
constexpr unsigned kSegMin = 64;                                 // shorter chains are never worth a unit

// The weight code k unit updates after code `c0`: c0 + step k, pinned at cmax (see fast_survival in hg_insert.hip
// for the same orbit in closed form; here every use is verified against the exact recurrence).
struct WeightOrbit {
  int c0, step, cmax;
  __device__ inline void init(const ChainCodec& g, float maximum_weight, uint32_t wcode) {
    c0 = static_cast<int>(wcode & 0x7FFFu);
    if (c0 == 0) c0 = 1;  // unknown and code 1 both decode to weight 0
    step = static_cast<int>(roundf(g.weight_resolution));
    // weight_to_value (hg_device.h) of the smaller of the inserter's and the codec's maximum
    const float wmax = maximum_weight < g.max_weight ? maximum_weight : g.max_weight;
    cmax = static_cast<int>(roundf((clampf(wmax, 0.f, g.max_weight) - 0.f) * g.weight_resolution)) + 1;
  }
  __device__ inline int code(unsigned k) const {
    const long long c = static_cast<long long>(c0) + static_cast<long long>(step) * static_cast<long long>(k);
    return c > cmax ? cmax : static_cast<int>(c);
  }
};

// ceil(n / s) for n < 2^22, 1 <= s <= 8, without the integer-division sequence: the float quotient of two small
// integers is either an exact integer or at least 1/8 away from one, so truncating the correctly rounded
// quotient gives floor((n + s - 1) / s).
__device__ inline unsigned seg_ceil_div(unsigned n, unsigned s) {
  return static_cast<unsigned>(static_cast<float>(n + s - 1u) / static_cast<float>(s));
}

// Segment counts for the H heavy voxels of a round over `units` wavefronts, evaluated in lanes 0..7 of every
// wavefront (lane h holds voxel h): every voxel starts with one segment, the remaining wavefronts go one at a
// time to the voxel with the longest segments. Returns this lane's count S_h (0 for lanes >= H).
__device__ inline unsigned seg_counts(unsigned n_h, unsigned H, unsigned units, unsigned lane) {
  unsigned S = lane < H ? 1u : 0u;
  if (H == 1u) return lane == 0u ? units : 0u;
  for (unsigned used = H; used < units; ++used) {
    const unsigned m = lane < H ? seg_ceil_div(n_h, S) : 0u;
    unsigned key = (m << 3) | (7u - (lane & 7u));  // longest segments first, lowest voxel on ties
    key = max(key, static_cast<unsigned>(__shfl_xor(static_cast<int>(key), 1)));
    key = max(key, static_cast<unsigned>(__shfl_xor(static_cast<int>(key), 2)));
    key = max(key, static_cast<unsigned>(__shfl_xor(static_cast<int>(key), 4)));
    if ((key >> 3) <= 16u) break;  // nothing left worth cutting
    if (lane == 7u - (key & 7u)) ++S;
  }
  return S;
}

// Applies the H <= kSegUnits chains listed in list[kSegB0 / kSegCnt / kSegVox / kSegCode] (values
// vals[b0 .. b0 + cnt) in LDS, in update order) to the voxels block[vox]. Called by all kSegUnits x 64 threads of a
// workgroup with uniform arguments (tid: the thread's index); list and vals must be visible (barrier) on entry; the
// caller places a barrier behind it before it reuses scratch, list or vals. When the plan does not pay (many short
// chains), thread h < H applies chain h on its own, as k_bin_apply's passes do. Requires div_in_range_ok(g).
// (first_wave > 0 keeps the wavefronts below it free of units; unused since the chains are applied in the tail of
// k_bin_apply, where nothing runs next to them.)
__device__ __forceinline__ void seg_chains(const ChainCodec& g, float maximum_weight, const uint32_t* vals,
                                           uint32_t* list, uint32_t* scratch, uint32_t* block, unsigned H,
                                           unsigned first_wave, unsigned tid) {
  constexpr bool FAST = true;
  auto light = [&](bool seg) {
    if (!seg && tid < H && list[kSegCnt + tid] != 0u) {
      const uint32_t c = update_chain_unit_t<true>(g, maximum_weight, list[kSegCode + tid], vals + list[kSegB0 + tid],
                                                   list[kSegCnt + tid]);
      block[list[kSegVox + tid]] = c;
      list[kSegCode + tid] = c;
    }
  };
  const unsigned wave = tid / kWave, lane = tid % kWave;
#ifdef HG_SEG_STATS
  long long seg_ts[6] = {0, 0, 0, 0, 0, 0};
#endif
  SEG_STAMP(0);
  // ---- units: lane h < H of every wavefront holds voxel h ----
  const unsigned n_l = lane < H ? list[kSegCnt + lane] : 0u;
  const unsigned S_l = seg_counts(n_l, H, kSegUnits - first_wave, lane);
  {
    // does it pay? longest segment (at the pace of eight wavefronts sharing four SIMDs) + set-up against the
    // longest chain as it stands, in update steps
    unsigned key = (lane < H ? seg_ceil_div(n_l, S_l) : 0u) | (n_l << 16);  // n < 2^16 (one LDS pass), m <= n
    unsigned mm = key & 0xFFFFu, nn = key >> 16;
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
      mm = max(mm, static_cast<unsigned>(__shfl_xor(static_cast<int>(mm), off)));
      nn = max(nn, static_cast<unsigned>(__shfl_xor(static_cast<int>(nn), off)));
    }
    mm = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(mm)));
    nn = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(nn)));
    if (!(mm + (mm >> 2) + 36u < nn)) {
      light(false);
      return;
    }
  }
  unsigned incl = S_l;  // inclusive prefix over lanes 0..7
  {
    unsigned t = static_cast<unsigned>(__shfl_up(static_cast<int>(incl), 1));
    if ((lane & 7u) >= 1u) incl += t;
    t = static_cast<unsigned>(__shfl_up(static_cast<int>(incl), 2));
    if ((lane & 7u) >= 2u) incl += t;
    t = static_cast<unsigned>(__shfl_up(static_cast<int>(incl), 4));
    if ((lane & 7u) >= 4u) incl += t;
  }
  const unsigned unit = wave - first_wave;  // (wraps for the light wavefronts: never below any prefix)
  const unsigned long long later = __ballot(lane < H && incl > unit);
  const bool active = wave >= first_wave && later != 0ull;
  const int hsel = active ? __ffsll(static_cast<long long>(later)) - 1 : 0;
  const unsigned u_h = static_cast<unsigned>(hsel);
  const unsigned u_n = static_cast<unsigned>(__shfl(static_cast<int>(n_l), hsel));
  const unsigned u_S = static_cast<unsigned>(__shfl(static_cast<int>(S_l), hsel));
  const unsigned u_base = static_cast<unsigned>(__shfl(static_cast<int>(incl - S_l), hsel));  // first unit of the voxel
  const unsigned u_s = unit - u_base;                                                         // this unit's segment
  const unsigned u_m = active ? seg_ceil_div(u_n, u_S) : 0u;
  const unsigned sb = min(u_s * u_m, u_n), se = min(sb + u_m, u_n);
  if (tid < kSegUnits) {
    scratch[kSegBad + tid] = 0u;
    scratch[kSegFrom + tid] = 0u;
    scratch[kSegDone + tid] = 0u;
    scratch[kSegCur + tid] = tid < H ? (list[kSegCode + tid] & 0x7FFFu) : 0u;
    if (tid == 0) scratch[kSegAgain] = 0u;
  }
  uint32_t code0 = 0;
  WeightOrbit orbit;
  orbit.c0 = 1; orbit.step = 0; orbit.cmax = 1;
  const uint32_t* uv = vals;
  if (active) {
    code0 = list[kSegCode + u_h];
    orbit.init(g, maximum_weight, code0 >> 16);
    uv = vals + list[kSegB0 + u_h];
  }
  SEG_STAMP(1);
  // ---- the segment's affine map d -> A d + B of the chain without re-quantisation ----
  // (fp32: the map is only a prediction, and its rounding, ~1e-6 relative, is a fiftieth of a code)
  if (active && u_S > 1u && u_s + 1u < u_S) {  // (nobody starts behind the last segment)
    const unsigned len = se - sb;
    const unsigned q = (len + kWave - 1u) / kWave;
    const unsigned j0 = min(sb + lane * q, se), j1 = min(j0 + q, se);
    float A = 1.0f, B = 0.0f;
    for (unsigned j = j0; j < j1; ++j) {
      const float w = static_cast<float>(orbit.code(j)) * g.weight_scale + g.weight_offset;
      const float r = refined_rcp(w + 1.0f);
      const float a = w * r, b = __uint_as_float(uv[j]) * r;
      A = a * A;
      B = a * B + b;
    }
    // ordered tree reduction over the lanes: (earlier, later) -> later o earlier
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const float A2 = __shfl_down(A, off), B2 = __shfl_down(B, off);
      B = A2 * B + B2;  // (only lanes at multiples of 2 off hold meaningful values afterwards)
      A = A2 * A;
    }
    if (lane == 0) {
      scratch[kSegAB + 2u * unit] = __float_as_uint(A);
      scratch[kSegAB + 2u * unit + 1u] = __float_as_uint(B);
    }
  }
  __syncthreads();
  SEG_STAMP(2);
  // ---- rounds: candidates + exact chains per unit, then the walk per voxel ----
  // Round 0 centres every segment's window on the affine prediction. That prediction is good while the chain
  // contracts (young voxels) or its inputs scatter; it fails where the re-quantisation is NOT noise: at a saturated
  // weight W an update moves the code only when it differs from it by more than (W + 1) / 2 codes, so a voxel
  // whose updates agree (thousands of rays through one wall voxel) stays put while the affine chain drifts
  // towards them -- hundreds of codes over a segment. There the candidates' own trajectories are the better
  // prediction (they run parallel to the true one): a walk that leaves a window continues APPROXIMATELY through
  // the remaining tables (edge value + distance x the table's own slope) and re-centres the later windows on what
  // it finds; round 1 re-evaluates from the first miss on (its start code is exact by then). A miss in round 1
  // runs that segment sequentially, as does every miss of a call that was not worth a second round.
  int lo0 = 0;  // this unit's round-0 window base
  if (active && u_s > 0u) {
    const uint32_t tc = code0 & 0x7FFFu;
    float d = tc == 0u ? g.min_tsd : static_cast<float>(tc) * g.tsd_scale + g.tsd_offset;
    for (unsigned t = 0; t < u_s; ++t)
      d = __uint_as_float(scratch[kSegAB + 2u * (u_base + t)]) * d + __uint_as_float(scratch[kSegAB + 2u * (u_base + t) + 1u]);
    float x = (d - g.min_tsd) * g.tsd_resolution;
    x = x < 0.f ? 0.f : (x > 32766.f ? 32766.f : x);
    lo0 = static_cast<int>(x + 0.5f) + 1 - kSegHalf;
    if (lane == 0) scratch[kSegLo + unit] = static_cast<uint32_t>(lo0);
  }
  for (int round = 0; round < 2; ++round) {
    if (active) {
      const unsigned from = scratch[kSegFrom + u_h];
      if (scratch[kSegDone + u_h] == 0u && u_s >= from && (round == 0 || se > sb)) {
        uint32_t cand;
        if (u_s == from) {
          cand = scratch[kSegCur + u_h];  // exact: the voxel's code (round 0) or the code the walk arrived with
        } else {
          const int lo = round == 0 ? lo0 : static_cast<int>(scratch[kSegLo + unit]);
          int c = lo + static_cast<int>(lane);
          c = c < 1 ? 1 : (c > 32767 ? 32767 : c);
          cand = static_cast<uint32_t>(c);
        }
        const int wc_begin = u_s == 0u ? static_cast<int>((code0 >> 16) & 0x7FFFu) : orbit.code(sb);
        ChainState st;
        st.d = cand == 0u ? g.min_tsd : static_cast<float>(cand) * g.tsd_scale + g.tsd_offset;
        st.w = wc_begin == 0 ? 0.f : static_cast<float>(wc_begin) * g.weight_scale + g.weight_offset;
        st.rt = static_cast<float>(cand);
        st.rw = static_cast<float>(wc_begin);
        st.fixed = false;
        if (se > sb) chain_run<FAST>(g, maximum_weight, st, uv + sb, se - sb);
        scratch[kSegTab + unit * kWave + lane] = static_cast<uint32_t>(static_cast<int>(st.rt));
        // the weight orbit, verified: exact end of this segment == assumed start of the next (== final weight code)
        if (lane == 0 && se > sb && static_cast<int>(st.rw) != orbit.code(se)) scratch[kSegBad + u_h] = 1u;
      }
    }
    if (round == 0 && (first_wave == 0u || wave < first_wave)) light(true);
    if (round == 0) { SEG_STAMP(3); }
    __syncthreads();
    // ---- walk: wavefront first_wave + h takes voxel h (its first lane) ----
    {
      const unsigned h = wave - first_wave;
      const bool walker = wave >= first_wave && h < H;
      const int hs = walker ? static_cast<int>(h) : 0;
      const unsigned n = static_cast<unsigned>(__shfl(static_cast<int>(n_l), hs));
      const unsigned nseg = static_cast<unsigned>(__shfl(static_cast<int>(S_l), hs));
      const unsigned ubase = static_cast<unsigned>(__shfl(static_cast<int>(incl - S_l), hs));
      if (walker && lane == 0 && n != 0u && scratch[kSegDone + h] == 0u) {
        const unsigned m = seg_ceil_div(n, nseg);
        const uint32_t* hv = vals + list[kSegB0 + h];
        const uint32_t c0 = list[kSegCode + h];
        uint32_t out = 0;
        bool exact = true;
#ifdef HG_SEG_STATS
        if (scratch[kSegBad + h]) atomicAdd(&g_seg_stats[2], 1u);
#endif
        if (scratch[kSegBad + h]) {
          out = update_chain_unit_t<FAST>(g, maximum_weight, c0, hv, n);  // (never seen: the orbit is exact for the defaults)
        } else {
          WeightOrbit ob;
          ob.init(g, maximum_weight, c0 >> 16);
          const unsigned from = scratch[kSegFrom + h];
          uint32_t c = scratch[kSegTab + (ubase + from) * kWave];
          int ca = 0;  // the approximate code once the walk has left a window (round 0)
          for (unsigned s = from + 1u; s < nseg; ++s) {
            const unsigned sb_s = s * m;
            if (sb_s >= n) break;  // (empty trailing segments)
            const uint32_t* tab = scratch + kSegTab + (ubase + s) * kWave;
            const int lo_s = static_cast<int>(scratch[kSegLo + ubase + s]);
            if (exact) {
              const int idx = static_cast<int>(c) - lo_s;
#if defined(HG_SEG_STATS) && HG_SEG_STATS > 1
              {
                const int dev = idx - kSegHalf < 0 ? kSegHalf - idx : idx - kSegHalf;
                atomicAdd(&g_seg_stats[0], 1u);
                atomicMax(&g_seg_stats[3], static_cast<unsigned>(dev));
                atomicAdd(&g_seg_stats[4], static_cast<unsigned>(dev));
                if (!(idx >= 0 && idx < kWave)) atomicAdd(&g_seg_stats[1 + 4 * round], 1u);
              }
#endif
              if (idx >= 0 && idx < kWave) {
                c = tab[idx];
                continue;
              }
              if (round == 1 || n < 4u * kSegMin) {  // this segment sequentially from the true code
                const unsigned se_s = min(sb_s + m, n);
                const int wc = ob.code(sb_s);
                ChainState st;
                st.d = static_cast<float>(c) * g.tsd_scale + g.tsd_offset;
                st.w = static_cast<float>(wc) * g.weight_scale + g.weight_offset;
                st.rt = static_cast<float>(c);
                st.rw = static_cast<float>(wc);
                st.fixed = false;
                chain_run<FAST>(g, maximum_weight, st, hv + sb_s, se_s - sb_s);
                c = static_cast<uint32_t>(static_cast<int>(st.rt));
                continue;
              }
              // first miss of round 0: the next round starts here, exactly
              scratch[kSegFrom + h] = s;
              scratch[kSegCur + h] = c;
              scratch[kSegAgain] = 1u;
              exact = false;
              ca = static_cast<int>(c);
            }
            // approximate walk: nearest table entry + distance x the table's slope
            const int idx = ca - lo_s;
            const int ic = idx < 0 ? 0 : (idx > kWave - 1 ? kWave - 1 : idx);
            const float slope = static_cast<float>(static_cast<int>(tab[kWave - 1]) - static_cast<int>(tab[0])) * (1.0f / (kWave - 1));
            ca = static_cast<int>(tab[ic]) + static_cast<int>(rintf(static_cast<float>(idx - ic) * slope));
            ca = ca < 1 ? 1 : (ca > 32767 ? 32767 : ca);
            if (s + 1u < nseg) scratch[kSegLo + ubase + s + 1u] = static_cast<uint32_t>(ca - kSegHalf);
          }
          out = (c + kUpdateMarker) | (static_cast<uint32_t>(ob.code(n)) << 16);
        }
        if (exact) {
          block[list[kSegVox + h]] = out;
          list[kSegCode + h] = out;  // (a voxel applied in several chunks continues from here)
          scratch[kSegDone + h] = 1u;
        }
      }
    }
    if (round == 0) {
      SEG_STAMP(4);
      __syncthreads();
      if (scratch[kSegAgain] == 0u) break;
    }
  }
  SEG_STAMP(5);
#ifdef HG_SEG_STATS
  if (tid == 0) {
    const unsigned long long dur = static_cast<unsigned long long>(seg_ts[5] - seg_ts[0]);
    if (dur > atomicMax(&g_seg_longest, dur)) {
      for (int k = 0; k < 6; ++k) g_seg_stamps[k] = seg_ts[k];
      g_seg_stamps[6] = list[kSegCnt]; g_seg_stamps[7] = H + 100 * scratch[kSegAgain];
    }
  }
#endif
}

}  // namespace hg
