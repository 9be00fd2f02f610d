// Microbenchmark: cycles per update of the exact per-voxel update chain (one lane active).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

struct G { float min_tsd, max_tsd, max_weight, tsd_resolution, weight_resolution, tsd_scale, tsd_offset, weight_scale, weight_offset; };

__device__ inline float round_nonneg_plus1(float y) { return floorf(__builtin_fmaf(floorf(y), 0.5f, 1.5f)); }
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

// V0: the product's chain
__global__ void k_v0(G g, const float* vals, int n, float* out, long long* cyc) {
  __shared__ float sv[4096];
  for (int i = threadIdx.x; i < n; i += blockDim.x) sv[i] = vals[i];
  __syncthreads();
  if (threadIdx.x != 0) return;
  float d = g.min_tsd, w = 0.f, rt = 0, rw = 0;
  const float res2_t = 2.0f * g.tsd_resolution, res2_w = 2.0f * g.weight_resolution;
  long long t0 = __builtin_amdgcn_s_memrealtime();
  float next = sv[0];
  for (int j = 0; j < n; ++j) {
    const float u = next;
    if (j + 1 < n) next = sv[j + 1];
    float uw = w + 1.0f;
    const float ud = div_in_range(d * w + u, uw);
    uw = (g.max_weight < uw) ? g.max_weight : uw;
    rt = round_nonneg_plus1((__builtin_amdgcn_fmed3f(ud, g.min_tsd, g.max_tsd) - g.min_tsd) * res2_t);
    rw = round_nonneg_plus1((__builtin_amdgcn_fmed3f(uw, 0.f, g.max_weight) - 0.f) * res2_w);
    d = rt * g.tsd_scale + g.tsd_offset;
    w = rw * g.weight_scale + g.weight_offset;
  }
  long long t1 = __builtin_amdgcn_s_memrealtime();
  out[0] = rt; out[1] = rw;
  cyc[0] = t1 - t0;
}

// V1: weight sequence tabulated: per step (w, r = refined 1/(w+1), uw = w+1) from LDS
__global__ void k_v1(G g, const float* vals, int n, float* out, long long* cyc) {
  __shared__ float sv[4096];
  __shared__ float4 sw[4096];
  for (int i = threadIdx.x; i < n; i += blockDim.x) sv[i] = vals[i];
  __syncthreads();
  if (threadIdx.x == 0) {  // build the table (not timed)
    float w = 0.f;
    const float res2_w = 2.0f * g.weight_resolution;
    for (int j = 0; j < n; ++j) {
      float uw = w + 1.0f;
      const float r0 = __builtin_amdgcn_rcpf(uw);
      const float e0 = __builtin_fmaf(-uw, r0, 1.0f);
      const float r = __builtin_fmaf(e0, r0, r0);
      sw[j] = make_float4(w, r, uw, 0.f);
      float cw = (g.max_weight < uw) ? g.max_weight : uw;
      float rw = round_nonneg_plus1((__builtin_amdgcn_fmed3f(cw, 0.f, g.max_weight) - 0.f) * res2_w);
      w = rw * g.weight_scale + g.weight_offset;
    }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  float d = g.min_tsd, rt = 0;
  const float res2_t = 2.0f * g.tsd_resolution;
  long long t0 = __builtin_amdgcn_s_memrealtime();
  float next = sv[0];
  float4 nw = sw[0];
  for (int j = 0; j < n; ++j) {
    const float u = next;
    const float4 cw = nw;
    if (j + 1 < n) { next = sv[j + 1]; nw = sw[j + 1]; }
    const float num = d * cw.x + u;
    const float q0 = num * cw.y;
    const float rem0 = __builtin_fmaf(-cw.z, q0, num);
    const float q1 = __builtin_fmaf(rem0, cw.y, q0);
    const float rem1 = __builtin_fmaf(-cw.z, q1, num);
    const float ud = __builtin_fmaf(rem1, cw.y, q1);
    rt = round_nonneg_plus1((__builtin_amdgcn_fmed3f(ud, g.min_tsd, g.max_tsd) - g.min_tsd) * res2_t);
    d = rt * g.tsd_scale + g.tsd_offset;
  }
  long long t1 = __builtin_amdgcn_s_memrealtime();
  out[0] = rt;
  cyc[0] = t1 - t0;
}

// V4: the product's arithmetic with the values fetched 8 at a time, one block ahead
__global__ void k_v4(G g, const float* vals, int n, float* out, long long* cyc) {
  __shared__ float sv[4096];
  for (int i = threadIdx.x; i < n; i += blockDim.x) sv[i] = vals[i];
  __syncthreads();
  if (threadIdx.x != 0) return;
  float d = g.min_tsd, w = 0.f, rt = 0, rw = 0;
  const float res2_t = 2.0f * g.tsd_resolution, res2_w = 2.0f * g.weight_resolution;
  long long t0 = __builtin_amdgcn_s_memrealtime();
  constexpr int K = 8;
  float cur[K], nx[K];
  const unsigned count = n;
  auto step = [&](float u) {
    float uw = w + 1.0f;
    const float ud = div_in_range(d * w + u, uw);
    uw = (g.max_weight < uw) ? g.max_weight : uw;
    rt = round_nonneg_plus1((__builtin_amdgcn_fmed3f(ud, g.min_tsd, g.max_tsd) - g.min_tsd) * res2_t);
    rw = round_nonneg_plus1((__builtin_amdgcn_fmed3f(uw, 0.f, g.max_weight) - 0.f) * res2_w);
    d = rt * g.tsd_scale + g.tsd_offset;
    w = rw * g.weight_scale + g.weight_offset;
  };
  unsigned base = 0;
  if (count >= K) {
#pragma unroll
    for (int k = 0; k < K; ++k) cur[k] = sv[k];
    while (base + K <= count) {
      const unsigned nb = base + K;
#pragma unroll
      for (int k = 0; k < K; ++k) { const unsigned i = nb + k; nx[k] = sv[i < count ? i : count - 1]; }
#pragma unroll
      for (int k = 0; k < K; ++k) step(cur[k]);
#pragma unroll
      for (int k = 0; k < K; ++k) cur[k] = nx[k];
      base = nb;
    }
  }
  for (unsigned j = base; j < count; ++j) step(sv[j]);
  long long t1 = __builtin_amdgcn_s_memrealtime();
  out[0] = rt; out[1] = rw;
  cyc[0] = t1 - t0;
}

// V2: 14 dependent FMAs per step, nothing else
__global__ void k_v2(int n, float a, float* out, long long* cyc) {
  if (threadIdx.x != 0) return;
  float d = a;
  long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int j = 0; j < n; ++j) {
#pragma unroll
    for (int k = 0; k < 14; ++k) d = __builtin_fmaf(d, 1.0000001f, 1e-9f);
  }
  long long t1 = __builtin_amdgcn_s_memrealtime();
  out[0] = d;
  cyc[0] = t1 - t0;
}
// V3: like V2 but all 64 lanes active
__global__ void k_v3(int n, float a, float* out, long long* cyc) {
  float d = a + threadIdx.x;
  long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int j = 0; j < n; ++j) {
#pragma unroll
    for (int k = 0; k < 14; ++k) d = __builtin_fmaf(d, 1.0000001f, 1e-9f);
  }
  long long t1 = __builtin_amdgcn_s_memrealtime();
  out[threadIdx.x] = d;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  const int n = 4000;
  G g;
  const float res = 0.2f, rtd = 2.5f, maxw = 1000.f;
  g.max_tsd = rtd * res; g.min_tsd = -g.max_tsd; g.max_weight = maxw;
  g.tsd_resolution = 32766.f / (g.max_tsd - g.min_tsd); g.weight_resolution = 32766.f / maxw;
  g.tsd_scale = (g.max_tsd - g.min_tsd) / 32766.f; g.tsd_offset = g.min_tsd - g.tsd_scale;
  g.weight_scale = maxw / 32766.f; g.weight_offset = -g.weight_scale;
  std::vector<float> v(n);
  for (int i = 0; i < n; ++i) v[i] = 0.3f * sinf(0.37f * i);
  float *dv, *dout; long long* dc;
  hipMalloc(&dv, n * 4); hipMalloc(&dout, 256 * 4); hipMalloc(&dc, 64);
  hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice);
  float o0[2], o1[2]; long long c;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_v0, dim3(1), dim3(64), 0, 0, g, dv, n, dout, dc); hipDeviceSynchronize();
    hipMemcpy(o0, dout, 8, hipMemcpyDeviceToHost); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("V0 product chain:      %.1f ticks(100MHz)/update -> %.4f us/update  (rt %.0f rw %.0f)\n", double(c) / n, double(c) / n * 0.01, o0[0], o0[1]);
    hipLaunchKernelGGL(k_v1, dim3(1), dim3(64), 0, 0, g, dv, n, dout, dc); hipDeviceSynchronize();
    hipMemcpy(o1, dout, 8, hipMemcpyDeviceToHost); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("V1 tabulated weights:  %.4f us/update (rt %.0f, same %d)\n", double(c) / n * 0.01, o1[0], o1[0] == o0[0]);
    hipLaunchKernelGGL(k_v4, dim3(1), dim3(64), 0, 0, g, dv, n, dout, dc); hipDeviceSynchronize();
    hipMemcpy(o1, dout, 8, hipMemcpyDeviceToHost); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("V4 blocked prefetch:   %.4f us/update (rt %.0f rw %.0f, same %d)\n", double(c) / n * 0.01, o1[0], o1[1], o1[0] == o0[0] && o1[1] == o0[1]);
    hipLaunchKernelGGL(k_v2, dim3(1), dim3(64), 0, 0, n, 0.5f, dout, dc); hipDeviceSynchronize();
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("V2 14 dependent FMAs:  %.4f us/step = %.4f us per dependent op\n", double(c) / n * 0.01, double(c) / n * 0.01 / 14);
    hipLaunchKernelGGL(k_v3, dim3(1), dim3(64), 0, 0, n, 0.5f, dout, dc); hipDeviceSynchronize();
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("V3 same, 64 lanes:     %.4f us/step\n", double(c) / n * 0.01);
  }
  return 0;
}
