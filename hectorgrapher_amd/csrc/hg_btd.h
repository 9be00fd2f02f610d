// Twisted block-tridiagonal Cholesky of the LM step's normal equations (k_lm, hg_match.hip), in a header of its own
// so that scripts/tw_bench.hip can time and check it outside the kernel. Included INSIDE namespace hg (twice in the
// library: the plain and the -DHG_BIG build of hg_match.hip), after kWave, wave_sync() and the typedefs lds_f64 /
// lds_i32 (LDS-typed in the plain build, generic in the big one).
// (no include guard: the two builds include it in different namespaces of different translation units)

// The same system by a TWISTED block factorisation (round 5): elimination runs from BOTH ends of the chain towards
// the middle group and the solution is unwound from the middle outwards. Five block steps deep for nine groups where
// cyclic reduction has four levels, but a step is one 9 x 9 Cholesky, ONE coupling block and the neighbour's Schur
// update -- no fill blocks, no second coupling -- and EVERY GROUP HAS ITS OWN WAVEFRONT (lower chain: wavefronts
// 0 .. 3, upper chain 4 .. 7), which keeps the group's factor, its reciprocal diagonal and y in registers from its
// elimination to its unwinding: nothing of the factor is stored or reloaded (a single lane storing 63 doubles took
// 860 cycles, reloading them 390; in-kernel stamps). Inside a step every lane factorises the block redundantly
// (right-looking: the updates of a column are independent fused multiply-adds, the serial chain is rsqrt -> scale ->
// update per column) and carries two more rows through the same column loop: the right-hand side (y = L^-1 b) and,
// lane i, row i of the coupling block (X = E L^-T) -- no separate substitution passes. The blocks are read from and
// updated in the band matrix where it is (no dense copies); the solution goes straight to x. Wavefront 0 also takes
// the middle group: it parks its own factor in LDS while the others work and fetches it back while they unwind.
// 81 columns: 40k cycles (cyclic reduction) -> see DESIGN.md 3.3 for the stamps. 2 <= groups <= 9, 512 threads.
// Same factorisation up to the elimination order (a symmetric permutation), hence the same solution to rounding.
// (A function of its own, not inlined: inside k_lm's one register allocation its state pushed a hundred loop
// invariants of the surrounding step out to scratch memory and the reloads landed in the chain; the pointers are
// LDS-typed in the plain build so that the accesses stay ds operations.)
// Workspace: U / Ub (the update the middle group takes from the upper chain) | wavefront 0's parked factor |
// t per wavefront.
#if defined(HG_LM_STAMPS) && HG_LM_STAMPS >= 2  // (a stamp costs its wavefront several hundred cycles: s_memtime's round trip)
__device__ long long g_tw_stamps[64];
#define TW_STAMP(i) do { if (threadIdx.x == 0) g_tw_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
// (the value operand pins the stamp behind the arithmetic that produces it; ~170 cycles each)
#define TW_STAMP_DEP(i, val) do { long long t_; double v_ = (val); asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_), "+v"(v_) :: "memory"); \
    (val) = v_; if (threadIdx.x == 0) g_tw_stamps[i] = t_; } while (0)
#else
#define TW_STAMP(i) do {} while (0)
#define TW_STAMP_DEP(i, val) do {} while (0)
#endif
#ifndef HG_TW_SKIP
#define HG_TW_SKIP 0  // scripts/tw_bench.hip: bit 0 skip the Schur update, 1 the column loop, 2 the X / y stores, 3 the backsolve, 4 t
#endif
#ifndef HG_TW_CUT
#define HG_TW_CUT 0  // scripts/tw_bench.hip: leave the solve at point 1 .. 4 (timing by elimination; wrong results)
#endif
constexpr int kTwWs = 46 + 10 + 46 + 180;
#ifndef HG_TW_ATTR
#define HG_TW_ATTR __forceinline__
#endif
#ifndef HG_TW_UPSHIFT
#define HG_TW_UPSHIFT 2
#endif
template <int MB>
__device__ HG_TW_ATTR bool cholesky_solve_twisted(int W, lds_f64* A, lds_f64* b, lds_f64* x, lds_f64* ws,
                                                                 int groups, lds_i32* ok_flag) {
  constexpr int TRI = MB * (MB + 1) / 2;
  constexpr int oU = 0, oUb = 46, oP = 56, oT = 102;
  const int tid = threadIdx.x;
  const int wave = tid / kWave, lane = tid % kWave;
  const int Wm = W - 1;
  const int mid = groups / 2, nlo = mid, nup = groups - 1 - mid;
  const int maxlen = nlo > nup ? nlo : nup;
  const bool lower = wave < 4;
  // (wavefront w runs on SIMD w % 4: the two wavefronts of a slot -- lower step t, upper step t -- sit on different
  // SIMDs, t and (t + 2) % 4; on one SIMD each ran at half speed)
  const int my_step = lower ? wave : (wave - 4 + HG_TW_UPSHIFT) & 3;
  const bool have = lower ? my_step < nlo : my_step < nup;
  const int g = have ? (lower ? my_step : groups - 1 - my_step) : 0;  // this wavefront's group
  const int n = lower ? g + 1 : g - 1;                                 // takes its Schur update; solved before it
  const bool into_u = !lower && n == mid;
  // band entry (i, j), j <= i, at i * Wm + Wm + j: block entry (r, c) of group g at bg + r * Wm + c, its coupling
  // with the next group (rows r of group g + 1) at bg + (MB + r) * Wm + c
  const int bg = g * MB * W + Wm, bn = n * MB * W + Wm;
  // X(i, k) = row i of (coupling block of g and n, rows n's / columns g's) L^-T, kept where the coupling block is
  const int eb = (lower ? bg : bn) + MB * Wm, si = lower ? Wm : 1, sk = lower ? 1 : Wm;
  TW_STAMP(0);
  if (tid == 0) *ok_flag = 1;
  int er = static_cast<int>((sqrtf(8.0f * static_cast<float>(lane) + 1.0f) - 1.0f) * 0.5f);
  while (er * (er + 1) / 2 > lane) --er;
  while ((er + 1) * (er + 2) / 2 <= lane) ++er;
  const int ec = lane - er * (er + 1) / 2;
  // L: the factor's strict lower triangle; its diagonal holds the RECIPROCALS of the factor's diagonal
  double L[MB][MB];
  lds_f64* ty = ws + oT + wave * 20;  // this wavefront's y (kept until it unwinds) | t
  // The block at band offset `base` factorised in every lane, right-looking. Every lane carries one more row through
  // the same column loop: lane i < MB row i of the coupling block (-> X = E L^-T, stored back when `coupled`), lane MB
  // the right-hand side (-> y = L^-1 b, left in ty).
  auto factor = [&](int base, int rhs, bool coupled, lds_f64* ydst) {
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) L[i][j] = A[base + i * Wm + j];
    double X[MB];
    {
      const bool row = lane < MB;
      const lds_f64* px = row ? A + eb + lane * si : b + rhs;
      const int sx = row ? sk : 1;
#pragma unroll
      for (int k = 0; k < MB; ++k) X[k] = px[k * sx];
    }
    bool ok = true;
    if (!(HG_TW_SKIP & 2))
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      const double d = L[j][j];
      ok = ok && d > 0.0 && d < 1e300;
      // 1 / sqrt(d): the hardware estimate (2^-26) and one Newton step -- (1 - 1.5 e^2) / sqrt(d), two units in the
      // last place -- in four dependent operations; the library call is ten with its second-order term and its
      // special cases (d is positive and finite here, or the solve is reported as failed)
      const double y0 = __builtin_amdgcn_rsq(d);
      const double r = fma(y0, fma(-(d * y0), 0.5 * y0, 0.5), y0);
      L[j][j] = r;
#pragma unroll
      for (int i = j + 1; i < MB; ++i) L[i][j] *= r;
      X[j] *= r;
#pragma unroll
      for (int i = j + 1; i < MB; ++i) {
#pragma unroll
        for (int k = j + 1; k <= i; ++k) L[i][k] = fma(-L[i][j], L[k][j], L[i][k]);
        X[i] = fma(-X[j], L[i][j], X[i]);
      }
    }
    if (!ok && lane == 0) *ok_flag = 0;
    if (!(HG_TW_SKIP & 4) && lane <= MB && (coupled || lane == MB)) {
      lds_f64* px = lane < MB ? A + eb + lane * si : ydst;
      const int sx = lane < MB ? sk : 1;
#pragma unroll
      for (int k = 0; k < MB; ++k) px[k * sx] = X[k];
    }
  };
  // x = L^-T t in every lane, in place
  auto backsolve = [&](double (&tt)[MB]) {
    if (!(HG_TW_SKIP & 8))
#pragma unroll
    for (int k = MB - 1; k >= 0; --k) {
      tt[k] *= L[k][k];
#pragma unroll
      for (int j = 0; j < k; ++j) tt[j] = fma(-L[k][j], tt[k], tt[j]);
    }
  };
  auto park = [&](bool store) {  // wavefront 0's factor: to LDS / back (its y stays in ty)
    lds_f64* pk = ws + oP;
    if (store) {
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) pk[i * (i + 1) / 2 + j] = L[i][j];
      }
    } else {
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) L[i][j] = pk[i * (i + 1) / 2 + j];
    }
  };
  __syncthreads();
  TW_STAMP(1);
  if (HG_TW_CUT == 1) return true;
  // ---- elimination: slot t = step t of both chains
  const int park_slot = maxlen > 1 ? 1 : 0;
  for (int t = 0; t < maxlen; ++t) {
    if (have && t == my_step) {
      TW_STAMP(30);
      factor(bg, g * MB, true, ty);
      wave_sync();
      TW_STAMP(33);
      // Schur update of the neighbour: D_n -= X X^T, b_n -= X y, in one pass: y is row MB of X (lanes TRI .. pair it
      // with the rows of X). The upper chain's last step leaves its share for the middle group in U: the lower
      // chain's last step may be updating that block in the same slot.
      if (!(HG_TW_SKIP & 1) && lane < TRI + MB) {
        const bool isb = lane >= TRI;
        const int r = isb ? MB : er, c = isb ? lane - TRI : ec;
        const lds_f64* pr = isb ? ty : A + eb + r * si;
        const lds_f64* pc = A + eb + c * si;
        const int sr = isb ? 1 : sk;
        lds_f64* tgt = into_u ? ws + (isb ? oUb + c : oU + lane) : (isb ? b + n * MB + c : A + bn + er * Wm + ec);
        const double old = into_u ? 0.0 : *tgt;  // (in flight with the rows)
        double s0 = 0.0, s1 = 0.0;               // (two chains: half the dependent latency)
#pragma unroll
        for (int k = 0; k < MB; ++k) {
          if (k & 1) s1 = fma(pr[k * sr], pc[k * sk], s1);
          else s0 = fma(pr[k * sr], pc[k * sk], s0);
        }
        *tgt = old - (s0 + s1);
      }
      TW_STAMP(34);
    }
    if (wave == 0 && t == park_slot) park(true);
    __syncthreads();
    TW_STAMP(2 + t);
  }
  if (HG_TW_CUT == 2) return true;
  // ---- the middle group (wavefront 0), solved at once: its factor is in registers
  if (wave == 0) {
    const int bm = mid * MB * W + Wm;
    if (nup > 0) {
      if (lane < TRI) A[bm + er * Wm + ec] += ws[oU + lane];
      else if (lane >= 48 && lane < 48 + MB) b[mid * MB + lane - 48] += ws[oUb + lane - 48];
      wave_sync();
    }
    lds_f64* tm = ws + oT + 8 * 20;  // (wavefront 0's own slot holds the y of its group)
    factor(bm, mid * MB, false, tm);
    wave_sync();
    double tt[MB];
#pragma unroll
    for (int k = 0; k < MB; ++k) tt[k] = tm[k];
    backsolve(tt);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < MB; ++k) x[mid * MB + k] = tt[k];
    }
  }
  __syncthreads();
  TW_STAMP(20);
  if (HG_TW_CUT == 3) return true;
  // ---- unwinding, slot t: groups mid - 1 - t and mid + 1 + t: x_g = L_g^-T (y_g - X^T x_n)
  for (int t = 0; t < maxlen; ++t) {
    const bool mine = have && (lower ? mid - 1 - t : mid + 1 + t) == g;
    if (wave == 0 && t == 0) park(false);  // (in its own slot only when the lower chain is one group long)
    if (mine) {
      double tt[MB];
      lds_f64* tw = ty + 10;
      if (!(HG_TW_SKIP & 16) && lane < MB) {
        double t0 = ty[lane], t1 = 0.0;  // (two chains)
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          if (i & 1) t1 = fma(-A[eb + i * si + lane * sk], x[n * MB + i], t1);
          else t0 = fma(-A[eb + i * si + lane * sk], x[n * MB + i], t0);
        }
        tw[lane] = t0 + t1;
      }
      wave_sync();
#pragma unroll
      for (int k = 0; k < MB; ++k) tt[k] = tw[k];
      backsolve(tt);
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < MB; ++k) x[g * MB + k] = tt[k];
      }
    }
    __syncthreads();
    TW_STAMP(21 + t);
  }
  if (HG_TW_CUT == 4) return true;
  bool ok = *ok_flag != 0;
  for (int i = tid; i < groups * MB; i += static_cast<int>(blockDim.x))
    if (!isfinite(x[i])) *ok_flag = 0;  // benign race: same value
  __syncthreads();
  return ok && *ok_flag != 0;
}

