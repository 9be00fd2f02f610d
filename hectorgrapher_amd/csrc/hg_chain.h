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
__device__ inline bool div_in_range_ok(const GridView& g) {
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
template <bool FAST>
__device__ inline void chain_run(const GridView& g, float maximum_weight, ChainState& st, const uint32_t* vals,
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
template <bool FAST>
__device__ inline uint32_t update_chain_unit_t(const GridView& g, float maximum_weight, uint32_t code,
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
constexpr unsigned kSegUnits = 8;    // = wavefronts of the apply workgroup: one (voxel, segment) unit each
constexpr int kSegHalf = 32;         // candidates p - 32 ... p + 31
// LDS scratch of seg_chains, in 32-bit words (8-byte aligned base):
constexpr unsigned kSegTab = 0;                                  // kSegUnits x 64 end codes
constexpr unsigned kSegAB = kSegTab + kSegUnits * kWave;         // kSegUnits x (A, B) doubles
constexpr unsigned kSegLo = kSegAB + kSegUnits * 4;              // kSegUnits window bases p - 32 (int)
constexpr unsigned kSegBad = kSegLo + kSegUnits;                 // per voxel: the weight check failed
constexpr unsigned kSegB0 = kSegBad + kSegUnits;                 // the heavy voxels of the round: first value,
constexpr unsigned kSegCnt = kSegB0 + kSegUnits;                 //   number of updates,
constexpr unsigned kSegVox = kSegCnt + kSegUnits;                //   voxel inside the block
constexpr unsigned kSegWords = kSegVox + kSegUnits;

// The weight code k unit updates after code `c0`: c0 + step k, pinned at cmax (see fast_survival in hg_insert.hip
// for the same orbit in closed form; here every use is verified against the exact recurrence).
struct WeightOrbit {
  int c0, step, cmax;
  __device__ inline void init(const GridView& g, float maximum_weight, uint32_t wcode) {
    c0 = static_cast<int>(wcode & 0x7FFFu);
    if (c0 == 0) c0 = 1;  // unknown and code 1 both decode to weight 0
    step = static_cast<int>(roundf(g.weight_resolution));
    cmax = static_cast<int>(weight_to_value(g, maximum_weight < g.max_weight ? maximum_weight : g.max_weight));
  }
  __device__ inline int code(unsigned k) const {
    const long long c = static_cast<long long>(c0) + static_cast<long long>(step) * static_cast<long long>(k);
    return c > cmax ? cmax : static_cast<int>(c);
  }
};

struct SegUnit {
  unsigned h, s, nseg, ubase, sb, se;  // voxel, segment, segments of the voxel, first unit of the voxel, [sb, se)
  bool active;
};
// Segments for the H <= kSegUnits heavy voxels of a round: the smallest common segment length m with
// sum ceil(n_h / m) <= kSegUnits; unit index = wavefront. Every thread evaluates this from the list in LDS.
__device__ inline SegUnit seg_assign(const uint32_t* cnt, unsigned H, unsigned wave) {
  unsigned total = 0;
  for (unsigned h = 0; h < H; ++h) total += cnt[h];
  unsigned m = (total + kSegUnits - 1u) / kSegUnits;
  if (m < 8u) m = 8u;
  while (true) {
    unsigned S = 0;
    for (unsigned h = 0; h < H; ++h) S += (cnt[h] + m - 1u) / m;
    if (S <= kSegUnits) break;
    m += (m >> 3) + 1u;
  }
  SegUnit r;
  r.h = r.s = r.nseg = r.ubase = r.sb = r.se = 0u;
  r.active = false;
  unsigned u = 0;
  for (unsigned h = 0; h < H; ++h) {
    const unsigned nseg = (cnt[h] + m - 1u) / m;
    if (wave >= u && wave < u + nseg) {
      r.h = h; r.s = wave - u; r.nseg = nseg; r.ubase = u;
      r.sb = r.s * m;
      r.se = min(r.sb + m, cnt[h]);
      r.active = true;
    }
    u += nseg;
  }
  return r;
}

// Applies the H <= kSegUnits chains listed in scratch[kSegB0 / kSegCnt / kSegVox] (values vals[b0 .. b0 + cnt) in
// LDS, in update order) to the voxels block[vox]. Called by all threads of a workgroup of kSegUnits wavefronts with
// uniform arguments; the list must be visible (barrier) on entry; ends with a barrier (scratch and vals reusable).
template <bool FAST>
__device__ inline void seg_chains_t(const GridView& g, float maximum_weight, const uint32_t* vals, uint32_t* scratch,
                                    unsigned H, uint32_t* block, unsigned tid) {
  const unsigned wave = tid / kWave, lane = tid % kWave;
  const SegUnit U = seg_assign(scratch + kSegCnt, H, wave);
  if (tid < kSegUnits) scratch[kSegBad + tid] = 0u;
  uint32_t code0 = 0;
  WeightOrbit orbit;
  orbit.c0 = 1; orbit.step = 0; orbit.cmax = 1;
  const uint32_t* uv = vals;
  if (U.active) {
    code0 = block[scratch[kSegVox + U.h]];
    orbit.init(g, maximum_weight, code0 >> 16);
    uv = vals + scratch[kSegB0 + U.h];
  }
  // ---- the segment's affine map d -> A d + B of the chain without re-quantisation (fp64) ----
  if (U.active && U.nseg > 1u) {
    const unsigned len = U.se - U.sb;
    const unsigned q = (len + kWave - 1u) / kWave;
    const unsigned j0 = min(U.sb + lane * q, U.se), j1 = min(j0 + q, U.se);
    double A = 1.0, B = 0.0;
    for (unsigned j = j0; j < j1; ++j) {
      const float wf = static_cast<float>(orbit.code(j)) * g.weight_scale + g.weight_offset;
      const double w = static_cast<double>(wf);
      double r = static_cast<double>(__builtin_amdgcn_rcpf(wf + 1.0f));
      r = r * (2.0 - (w + 1.0) * r);  // one Newton step: ~1e-14
      const double a = w * r, b = static_cast<double>(__uint_as_float(uv[j])) * r;
      A = a * A;
      B = a * B + b;
    }
    // ordered tree reduction over the lanes: (earlier, later) -> later o earlier
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const double A2 = __shfl_down(A, off), B2 = __shfl_down(B, off);
      if ((lane & (2 * off - 1)) == 0) {
        B = A2 * B + B2;
        A = A2 * A;
      }
    }
    if (lane == 0) {
      double* ab = reinterpret_cast<double*>(scratch + kSegAB) + 2u * wave;
      ab[0] = A;
      ab[1] = B;
    }
  }
  __syncthreads();
  // ---- prediction of the segment's start code, candidates, exact chain ----
  if (U.active) {
    int lo = 0;
    uint32_t cand = code0 & 0x7FFFu;  // segment 0 starts from the voxel's code itself, in every lane
    if (U.s > 0u) {
      const uint32_t tc = code0 & 0x7FFFu;
      double d = tc == 0u ? static_cast<double>(g.min_tsd)
                          : static_cast<double>(static_cast<float>(tc) * g.tsd_scale + g.tsd_offset);
      const double* ab = reinterpret_cast<const double*>(scratch + kSegAB) + 2u * U.ubase;
      for (unsigned t = 0; t < U.s; ++t) d = ab[2 * t] * d + ab[2 * t + 1];
      double x = (d - static_cast<double>(g.min_tsd)) * static_cast<double>(g.tsd_resolution);
      x = x < 0.0 ? 0.0 : (x > 32766.0 ? 32766.0 : x);
      lo = static_cast<int>(x + 0.5) + 1 - kSegHalf;
      int c = lo + static_cast<int>(lane);
      c = c < 1 ? 1 : (c > 32767 ? 32767 : c);
      cand = static_cast<uint32_t>(c);
      if (lane == 0) scratch[kSegLo + wave] = static_cast<uint32_t>(lo);
    }
    const int wc_begin = U.s == 0u ? static_cast<int>((code0 >> 16) & 0x7FFFu) : orbit.code(U.sb);
    ChainState st;
    st.d = cand == 0u ? g.min_tsd : static_cast<float>(cand) * g.tsd_scale + g.tsd_offset;
    st.w = wc_begin == 0 ? 0.f : static_cast<float>(wc_begin) * g.weight_scale + g.weight_offset;
    st.rt = static_cast<float>(cand);
    st.rw = static_cast<float>(wc_begin);
    st.fixed = false;
    if (U.se > U.sb) chain_run<FAST>(g, maximum_weight, st, uv + U.sb, U.se - U.sb);
    scratch[kSegTab + wave * kWave + lane] = static_cast<uint32_t>(static_cast<int>(st.rt));
    // the weight orbit, verified: exact end of this segment == assumed start of the next (== final weight code)
    if (lane == 0 && static_cast<int>(st.rw) != orbit.code(U.se)) scratch[kSegBad + U.h] = 1u;
  }
  __syncthreads();
  // ---- walk: wavefront h < H takes voxel h (its first lane) ----
  if (wave < H && lane == 0) {
    const unsigned h = wave, n = scratch[kSegCnt + h];
    const uint32_t* hv = vals + scratch[kSegB0 + h];
    uint32_t* cell = block + scratch[kSegVox + h];
    const uint32_t c0 = *cell;
    // (the voxel's units: recomputed for voxel h -- seg_assign of wave' = first unit of h)
    unsigned ubase = 0, nseg = 0, m = 0;
    {
      unsigned total = 0;
      for (unsigned k = 0; k < H; ++k) total += scratch[kSegCnt + k];
      m = (total + kSegUnits - 1u) / kSegUnits;
      if (m < 8u) m = 8u;
      while (true) {
        unsigned S = 0;
        for (unsigned k = 0; k < H; ++k) S += (scratch[kSegCnt + k] + m - 1u) / m;
        if (S <= kSegUnits) break;
        m += (m >> 3) + 1u;
      }
      for (unsigned k = 0; k < h; ++k) ubase += (scratch[kSegCnt + k] + m - 1u) / m;
      nseg = (n + m - 1u) / m;
    }
    uint32_t out;
    if (scratch[kSegBad + h]) {
      out = update_chain_unit_t<FAST>(g, maximum_weight, c0, hv, n);  // (never seen: the orbit is exact for the defaults)
    } else {
      WeightOrbit ob;
      ob.init(g, maximum_weight, c0 >> 16);
      uint32_t c = scratch[kSegTab + ubase * kWave];
      for (unsigned s = 1; s < nseg; ++s) {
        const int idx = static_cast<int>(c) - static_cast<int>(scratch[kSegLo + ubase + s]);
        if (idx >= 0 && idx < kWave) {
          c = scratch[kSegTab + (ubase + s) * kWave + static_cast<unsigned>(idx)];
        } else {  // outside the window: this segment sequentially from the true code
          const unsigned sb = s * m, se = min(sb + m, n);
          const int wc = ob.code(sb);
          ChainState st;
          st.d = static_cast<float>(c) * g.tsd_scale + g.tsd_offset;
          st.w = static_cast<float>(wc) * g.weight_scale + g.weight_offset;
          st.rt = static_cast<float>(c);
          st.rw = static_cast<float>(wc);
          st.fixed = false;
          chain_run<FAST>(g, maximum_weight, st, hv + sb, se - sb);
          c = static_cast<uint32_t>(static_cast<int>(st.rt));
        }
      }
      out = (c + kUpdateMarker) | (static_cast<uint32_t>(ob.code(n)) << 16);
    }
    *cell = out;
  }
  __syncthreads();
}

__device__ inline void seg_chains(const GridView& g, float maximum_weight, const uint32_t* vals, uint32_t* scratch,
                                  unsigned H, uint32_t* block, unsigned tid) {
  if (div_in_range_ok(g)) seg_chains_t<true>(g, maximum_weight, vals, scratch, H, block, tid);
  else seg_chains_t<false>(g, maximum_weight, vals, scratch, H, block, tid);
}

}  // namespace hg
