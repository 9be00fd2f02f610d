// Microbenchmark + self-check of the twisted block-tridiagonal Cholesky (csrc/hg_btd.h: cholesky_solve_twisted) outside
// k_lm: one workgroup of 512 threads, a random SPD block-tridiagonal band system (groups x MB columns, half bandwidth
// 2 MB - 1) in LDS, solved REPS times (the band matrix and the right-hand side restored from a pristine LDS copy before
// every solve; the restore alone is timed too and subtracted). Prints cycles per solve (wall clock of the kernel, not
// in-kernel stamps: a stamp costs its wavefront hundreds of cycles) and the error against a dense host Cholesky.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I../hectorgrapher_amd/csrc tw_bench.hip -o tw_bench
//   ./tw_bench [groups=9] [reps=2000]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

namespace hg {
constexpr int kWave = 64;
__device__ inline void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) int lds_i32;
#include "hg_btd.h"
}  // namespace hg

constexpr int kMaxN = 81, kMaxW = 18;

template <int MB>
__global__ __launch_bounds__(512) void k_bench(const double* A0, const double* b0, double* x_out, int groups, int reps,
                                               int solve, int* ok_out, long long* first) {
  __shared__ __align__(16) double A[kMaxN * kMaxW + 2];
  __shared__ __align__(16) double Ap[kMaxN * kMaxW + 2];
  __shared__ __align__(16) double b[kMaxN], bp[kMaxN], x[kMaxN];
  __shared__ __align__(16) double ws[hg::kTwWs + 2];
  __shared__ int ok;
  const int n = groups * MB, W = 2 * MB, nW = n * W;
  for (int i = threadIdx.x; i < nW; i += blockDim.x) Ap[i] = A0[i];
  for (int i = threadIdx.x; i < n; i += blockDim.x) bp[i] = b0[i];
  __syncthreads();
  int all_ok = 1;
  for (int r = 0; r < reps; ++r) {
    for (int i = threadIdx.x; i < nW; i += blockDim.x) A[i] = Ap[i];
    for (int i = threadIdx.x; i < n; i += blockDim.x) b[i] = bp[i];
    __syncthreads();
#ifdef TW_EVICT_ICACHE
    {  // 96 KB of straight-line code between the solves: the solve's code is not in the instruction cache when it starts,
       // as in k_lm, which runs once per launch behind a residual kernel (the time is in the solve = 0 run as well)
      float ev = static_cast<float>(r);
      asm volatile(".rept 12288\n v_add_f32 %0, %0, %0\n .endr" : "+v"(ev));
      if (ev == 123.456f) x[0] = ev;
    }
    __syncthreads();
#endif
    const long long t0 = r < 8 ? __builtin_amdgcn_s_memtime() : 0;
    if (solve) {
      const bool good = hg::cholesky_solve_twisted<MB>(W, (hg::lds_f64*)A, (hg::lds_f64*)b, (hg::lds_f64*)x, (hg::lds_f64*)ws, groups,
                                                      (hg::lds_i32*)&ok);
      all_ok &= good ? 1 : 0;
    }
    __syncthreads();
    if (r < 8 && threadIdx.x == 0) first[r] = __builtin_amdgcn_s_memtime() - t0;  // (cold instruction cache: the first solve)
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) x_out[i] = x[i];
  if (threadIdx.x == 0) *ok_out = all_ok;
}

int main(int argc, char** argv) {
  const int groups = argc > 1 ? std::atoi(argv[1]) : 9;
  const int reps = argc > 2 ? std::atoi(argv[2]) : 2000;
  const int MB = 9, n = groups * MB, W = 2 * MB, Wm = W - 1;
  // SPD block tridiagonal: J^T J of a random J whose rows couple one group or two neighbouring groups, + diagonal
  std::mt19937_64 rng(7);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::vector<double> dense(static_cast<size_t>(n) * n, 0.0), rhs(n);
  for (int g = 0; g < groups; ++g) {
    const int span = g + 1 < groups ? 2 * MB : MB;
    for (int row = 0; row < 40; ++row) {
      std::vector<double> j(span);
      for (double& v : j) v = nd(rng);
      for (int a = 0; a < span; ++a)
        for (int c = 0; c < span; ++c) dense[static_cast<size_t>(g * MB + a) * n + g * MB + c] += j[a] * j[c];
    }
  }
  for (int i = 0; i < n; ++i) {
    dense[static_cast<size_t>(i) * n + i] += 1.0;
    rhs[i] = nd(rng);
  }
  std::vector<double> band(static_cast<size_t>(n) * W + 2, 0.0);
  for (int i = 0; i < n; ++i)
    for (int j = std::max(0, i - Wm); j <= i; ++j) band[static_cast<size_t>(i) * Wm + Wm + j] = dense[static_cast<size_t>(i) * n + j];
  // host reference: dense Cholesky
  std::vector<double> Lh = dense, xr = rhs;
  for (int j = 0; j < n; ++j) {
    double d = Lh[static_cast<size_t>(j) * n + j];
    for (int k = 0; k < j; ++k) d -= Lh[static_cast<size_t>(j) * n + k] * Lh[static_cast<size_t>(j) * n + k];
    d = std::sqrt(d);
    Lh[static_cast<size_t>(j) * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double t = Lh[static_cast<size_t>(i) * n + j];
      for (int k = 0; k < j; ++k) t -= Lh[static_cast<size_t>(i) * n + k] * Lh[static_cast<size_t>(j) * n + k];
      Lh[static_cast<size_t>(i) * n + j] = t / d;
    }
  }
  for (int i = 0; i < n; ++i) {
    double t = xr[i];
    for (int k = 0; k < i; ++k) t -= Lh[static_cast<size_t>(i) * n + k] * xr[k];
    xr[i] = t / Lh[static_cast<size_t>(i) * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double t = xr[i];
    for (int k = i + 1; k < n; ++k) t -= Lh[static_cast<size_t>(k) * n + i] * xr[k];
    xr[i] = t / Lh[static_cast<size_t>(i) * n + i];
  }
  double *dA, *db, *dx;
  int* dok;
  long long* dfirst;
  (void)hipMalloc(&dfirst, 64);
  (void)hipMalloc(&dA, band.size() * 8);
  (void)hipMalloc(&db, n * 8);
  (void)hipMalloc(&dx, n * 8);
  (void)hipMalloc(&dok, 4);
  (void)hipMemcpy(dA, band.data(), band.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(db, rhs.data(), n * 8, hipMemcpyHostToDevice);
  double ms[2] = {0, 0};
  for (int solve = 0; solve < 2; ++solve) {
    for (int rep = 0; rep < 3; ++rep) {  // (the last of three runs counts)
      (void)hipDeviceSynchronize();
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(k_bench<9>, dim3(1), dim3(512), 0, 0, dA, db, dx, groups, reps, solve, dok, dfirst);
      (void)hipDeviceSynchronize();
      ms[solve] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
  }
  std::vector<double> xg(n);
  int ok = 0;
  (void)hipMemcpy(xg.data(), dx, n * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(&ok, dok, 4, hipMemcpyDeviceToHost);
  double err = 0.0, nrm = 0.0;
  for (int i = 0; i < n; ++i) {
    err = std::max(err, std::fabs(xg[i] - xr[i]));
    nrm = std::max(nrm, std::fabs(xr[i]));
  }
  long long first[8];
  (void)hipMemcpy(first, dfirst, 64, hipMemcpyDeviceToHost);
  printf("first solves of the launch (s_memtime ticks, incl. ~2 stamps): %lld %lld %lld %lld %lld\n", first[0], first[1], first[2], first[3], first[7]);
  int clk_khz = 0;
  (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  const double us = (ms[1] - ms[0]) * 1e3 / reps;
  printf("groups %d (n = %d): %.2f us per solve = %.0f cycles at %.2f GHz (restore loop alone %.2f us); ok %d, max |dx| %.3g of %.3g\n",
         groups, n, us, us * clk_khz * 1e-3, clk_khz * 1e-6, ms[0] * 1e3 / reps, ok, err, nrm);
  return (ok && err <= 1e-9 * std::max(1.0, nrm)) ? 0 : 1;
}
