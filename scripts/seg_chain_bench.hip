// Microbenchmark + self-check of the segmented chain evaluation (csrc/hg_chain.h: seg_chains) against the sequential
// exact chain (update_chain_unit), outside the insert kernels: one workgroup of 512 threads, H heavy voxels with n
// updates each, values in LDS. Prints, per case, whether all codes are identical, the misses seen and the cycles
// (s_memrealtime, 100 MHz) of both forms.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I../hectorgrapher_amd/csrc seg_chain_bench.hip -o seg_chain_bench
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#ifndef HG_SEG_STATS
#define HG_SEG_STATS 1  // 2: per-lookup statistics as well (their atomics slow the walk down)
#endif
#include "hg_chain.h"

using namespace hg;

__global__ __launch_bounds__(512) void k_case(GridView g, float maxw, const uint32_t* vals_g, unsigned total,
                                              const uint32_t* b0, const uint32_t* cnt, unsigned H, uint32_t* block_seg,
                                              uint32_t* block_ref, long long* cyc) {
  __shared__ uint32_t sv[8192];
  __shared__ uint32_t scratch[kSegWords];
  __shared__ uint32_t list[kSegListWords];
  const unsigned tid = threadIdx.x;
  for (unsigned i = tid; i < total; i += blockDim.x) sv[i] = vals_g[i];
  if (tid < H) {
    list[kSegB0 + tid] = b0[tid];
    list[kSegCnt + tid] = cnt[tid];
    list[kSegVox + tid] = tid;
    list[kSegCode + tid] = block_seg[tid];
  }
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memrealtime();
  seg_chains(chain_codec(g), maxw, sv, list, scratch, block_seg, H, 0u, tid);
  __syncthreads();
  long long t1 = __builtin_amdgcn_s_memrealtime();
  __syncthreads();
  // reference: thread h applies voxel h sequentially (as k_bin_apply does today)
  if (tid < H) block_ref[tid] = update_chain_unit(g, maxw, block_ref[tid], sv + b0[tid], cnt[tid]);
  long long t2 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}

static GridView make_view(float resolution, float rel_trunc, float max_weight) {
  GridView v{};
  v.resolution = resolution;
  const float tau = static_cast<float>(static_cast<double>(rel_trunc) * resolution);
  v.max_tsd = tau; v.min_tsd = -tau; v.max_weight = max_weight;
  v.tsd_resolution = 32766.f / (v.max_tsd - v.min_tsd);
  v.weight_resolution = 32766.f / v.max_weight;
  v.tsd_scale = (v.max_tsd - v.min_tsd) / 32766.f;
  v.tsd_offset = v.min_tsd - v.tsd_scale;
  v.weight_scale = (v.max_weight - 0.f) / 32766.f;
  v.weight_offset = 0.f - v.weight_scale;
  return v;
}

int main() {
  std::mt19937 rng(7);
  struct Case { unsigned H, n; int start; float maxw; const char* what; };
  // start: 0 = unknown voxel, 1 = young voxel (weight code 34 * 20), 2 = saturated weight
  const Case cases[] = {
      {1, 96, 0, 1000.f, "1 x 96 fresh"},       {1, 300, 0, 1000.f, "1 x 300 fresh"},
      {1, 300, 2, 1000.f, "1 x 300 saturated"}, {1, 1000, 1, 1000.f, "1 x 1000 young"},
      {1, 2048, 2, 1000.f, "1 x 2048 saturated"}, {1, 2048, 0, 1000.f, "1 x 2048 fresh"},
      {3, 300, 2, 1000.f, "3 x 300 saturated"}, {8, 200, 1, 1000.f, "8 x 200 young"},
      {2, 1024, 2, 1000.f, "2 x 1024 saturated"}, {1, 2048, 2, 50.f, "1 x 2048 saturated, max weight 50"},
      {5, 131, 0, 1000.f, "5 x 131 fresh (uneven)"}, {1, 2048, 1, 20000.f, "1 x 2048 young, max weight 20000"},
      {1, 2048, 3, 1000.f, "1 x 2048 saturated, updates agree"}, {2, 1000, 3, 1000.f, "2 x 1000 saturated, updates agree"},
  };
  uint32_t *d_vals, *d_b0, *d_cnt, *d_seg, *d_ref;
  long long* d_cyc;
  hipMalloc(&d_vals, 8192 * 4); hipMalloc(&d_b0, 32); hipMalloc(&d_cnt, 32); hipMalloc(&d_seg, 32); hipMalloc(&d_ref, 32);
  hipMalloc(&d_cyc, 16);
  int failures = 0;
  for (float res : {0.05f, 0.10f, 0.20f}) {
    for (const Case& c : cases) {
      GridView g = make_view(res, 2.5f, c.maxw);
      const float tau = g.max_tsd;
      long long seg_sum = 0, ref_sum = 0;
      int bad = 0;
      const int reps = 20;
      for (int rep = 0; rep < reps; ++rep) {
        std::vector<uint32_t> vals, b0(c.H), cnt(c.H), code(c.H);
        for (unsigned h = 0; h < c.H; ++h) {
          b0[h] = static_cast<uint32_t>(vals.size());
          cnt[h] = c.n + (c.H > 1 ? h * 7 : 0);
          // a surface drifting through the voxel plus noise, clamped to the truncation band like InsertHit's samples
          // start 3: a wall voxel -- saturated weight, updates that agree with each other (a few mm of spread) but
          // not with the voxel's value: the re-quantisation holds the code back, the affine prediction drifts
          std::normal_distribution<float> noise(0.f, c.start == 3 ? 0.004f * tau : 0.3f * tau);
          const float centre = std::uniform_real_distribution<float>(-0.8f * tau, 0.8f * tau)(rng);
          for (unsigned k = 0; k < cnt[h]; ++k) {
            float u = centre + noise(rng) + (c.start == 3 ? 0.f : 0.2f * tau * std::sin(0.01f * k));
            u = std::fmax(-tau, std::fmin(tau, u));
            uint32_t bits;
            std::memcpy(&bits, &u, 4);
            vals.push_back(bits);
          }
          const uint32_t tc = 1 + rng() % 32767;
          const int step = static_cast<int>(std::lround(g.weight_resolution));
          const uint32_t wc = c.start == 0 ? 0 : c.start == 1 ? 1 + step * 20 : 32767;
          (void)step;
          code[h] = c.start == 0 ? 0u : ((tc | 0x8000u) | (wc << 16));
        }
        hipMemcpy(d_vals, vals.data(), vals.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(d_b0, b0.data(), c.H * 4, hipMemcpyHostToDevice);
        hipMemcpy(d_cnt, cnt.data(), c.H * 4, hipMemcpyHostToDevice);
        hipMemcpy(d_seg, code.data(), c.H * 4, hipMemcpyHostToDevice);
        hipMemcpy(d_ref, code.data(), c.H * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_case, dim3(1), dim3(512), 0, 0, g, c.maxw > 1000.f ? 1000.f : c.maxw, d_vals,
                           static_cast<unsigned>(vals.size()), d_b0, d_cnt, c.H, d_seg, d_ref, d_cyc);
        std::vector<uint32_t> a(c.H), b(c.H);
        long long cyc[2];
        hipMemcpy(a.data(), d_seg, c.H * 4, hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), d_ref, c.H * 4, hipMemcpyDeviceToHost);
        hipMemcpy(cyc, d_cyc, 16, hipMemcpyDeviceToHost);
        for (unsigned h = 0; h < c.H; ++h) bad += a[h] != b[h];
        if (rep) { seg_sum += cyc[0]; ref_sum += cyc[1]; }
      }
      failures += bad;
      unsigned st[8] = {0};
      hipMemcpyFromSymbol(st, HIP_SYMBOL(hg::g_seg_stats), sizeof(st));
      long long sp[8];
      hipMemcpyFromSymbol(sp, HIP_SYMBOL(hg::g_seg_stamps), sizeof(sp));
      std::printf("    phases (10 ns): assign+load %lld  affine %lld  predict+chain(wave 0) %lld  barrier %lld  walk %lld\n",
                  sp[1] - sp[0], sp[2] - sp[1], sp[3] - sp[2], sp[4] - sp[3], sp[5] - sp[4]);
      unsigned zero[8] = {0};
      hipMemcpyToSymbol(HIP_SYMBOL(hg::g_seg_stats), zero, sizeof(zero));
      std::printf("res %.2f  %-36s  %s  segmented %7.2f us  sequential %7.2f us  (x%.1f)  lookups %u misses %u (second round: %u) weight-check failures %u  |c-p| max %u mean %.2f\n", res, c.what,
                  bad ? "MISMATCH" : "identical", seg_sum / (reps - 1) * 0.01, ref_sum / (reps - 1) * 0.01,
                  static_cast<double>(ref_sum) / static_cast<double>(seg_sum ? seg_sum : 1), st[0], st[1], st[5], st[2], st[3],
                  st[0] ? static_cast<double>(st[4]) / st[0] : 0.0);
    }
  }
  std::printf(failures ? "FAILED: %d mismatches\n" : "all identical\n", failures);
  return failures ? 1 : 0;
}
