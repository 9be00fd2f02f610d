// tcp_bench.hip — how the vector L1 (TCP) prices gather loads by the number of distinct 128-B lines a
// wave instruction touches. Every lane loads 8 bytes (global_load_dwordx2) from an L1-resident region; lanes are
// grouped g at a time onto one line (g = 1: 64 lines per instruction ... g = 64: one line).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/tcp_bench.hip -o scripts/tcp_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int G, int ADJ>
__global__ __launch_bounds__(256) void k_gather(const unsigned long long* __restrict__ buf, unsigned long long* out, int iters, unsigned lines_mask) {
  const unsigned lane = threadIdx.x & 63u;
  // ADJ = 1: the g lanes of a group are adjacent lanes; ADJ = 0: they are spread 64 / g apart
  const unsigned group = ADJ ? lane / G : lane % (64 / G);
  const unsigned within = ADJ ? lane % G : lane / (64 / G);
  unsigned long long acc = 0;
  unsigned line = (group * 37u + blockIdx.x * 11u + (threadIdx.x >> 6) * 5u) & lines_mask;
  const unsigned long long* base = buf + (blockIdx.x % 8) * 4096;  // 32 KB window per "CU slot"
#pragma unroll 8
  for (int i = 0; i < iters; ++i) {
    acc += base[line * 16u + (within & 15u)];
    line = (line + 13u) & lines_mask;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// exec-masked variant: only one lane in MASKN takes part (the face words of the matcher: one lane in eight)
template <int MASKN>
__global__ __launch_bounds__(256) void k_gather_masked(const unsigned long long* __restrict__ buf, unsigned long long* out, int iters, unsigned lines_mask) {
  const unsigned lane = threadIdx.x & 63u;
  unsigned long long acc = 0;
  unsigned line = (lane * 37u + blockIdx.x * 11u + (threadIdx.x >> 6) * 5u) & lines_mask;
  const unsigned long long* base = buf + (blockIdx.x % 8) * 4096;
  if (lane % MASKN == MASKN - 1) {
#pragma unroll 8
    for (int i = 0; i < iters; ++i) {
      acc += base[line * 16u + (lane & 15u)];
      line = (line + 13u) & lines_mask;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MASKN>
void run_masked(const unsigned long long* d, unsigned long long* o, unsigned mask) {
  const int iters = 4096, blocks = 256 * 8;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k_gather_masked<MASKN>), dim3(blocks), dim3(256), 0, 0, d, o, 64, mask);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k_gather_masked<MASKN>), dim3(blocks), dim3(256), 0, 0, d, o, iters, mask);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double wave_instr = double(blocks) * 4 * iters;
  printf("one lane in %2d active (%2d lines per instruction): %.3f ms, %.1f cycles per wave-instruction per CU\n", MASKN, 64 / MASKN, ms,
         ms * 1e-3 * 2.4e9 / (wave_instr / 256.0));
}

template <int G, int ADJ>
void run(const unsigned long long* d, unsigned long long* o, unsigned mask) {
  const int iters = 4096, blocks = 256 * 8;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k_gather<G, ADJ>), dim3(blocks), dim3(256), 0, 0, d, o, 64, mask);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k_gather<G, ADJ>), dim3(blocks), dim3(256), 0, 0, d, o, iters, mask);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double wave_instr = double(blocks) * 4 * iters;
  const double per_cu_cycles = ms * 1e-3 * 2.4e9;  // nominal clock
  printf("G=%2d adj=%d lines/instr=%2d: %.3f ms, %.1f cycles per wave-instruction per CU (nominal 2.4 GHz), %.2f lines/cycle/CU\n", G, ADJ,
         64 / G, ms, per_cu_cycles / (wave_instr / 256.0), (wave_instr / 256.0) * (64 / G) / per_cu_cycles);
}

int main() {
  unsigned long long *d, *o;
  hipMalloc(&d, 8 * 32768 * 8);
  hipMalloc(&o, 256 * 8 * 256 * 8);
  hipMemset(d, 1, 8 * 32768 * 8);
  const unsigned mask = 127;  // 128 lines = 16 KB per window: L1-resident
  run<1, 1>(d, o, mask); run<2, 1>(d, o, mask); run<4, 1>(d, o, mask); run<8, 1>(d, o, mask); run<16, 1>(d, o, mask); run<64, 1>(d, o, mask);
  run<2, 0>(d, o, mask); run<4, 0>(d, o, mask); run<16, 0>(d, o, mask);
  run_masked<1>(d, o, mask); run_masked<2>(d, o, mask); run_masked<4>(d, o, mask); run_masked<8>(d, o, mask); run_masked<16>(d, o, mask); run_masked<64>(d, o, mask);
  return 0;
}
