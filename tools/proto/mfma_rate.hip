// MFMA issue-rate micro-benchmark for the instruction patterns of the encoder kernels (fp32-input MFMA).
//   pattern 0: v_mfma_f32_16x16x4_f32, 12 independent accumulators (3 row x 4 col blocks), order (t, c, r) = fused_mlp.hip
//   pattern 1: v_mfma_f32_32x32x2_f32, 2 accumulators (wgrad_dma_body: NBM = 2, NBN = 1), order (tt, i)
//   pattern 2: v_mfma_f32_32x32x2_f32, 4 accumulators
//   pattern 3: v_mfma_f32_16x16x4_f32, 12 accumulators, one ds_read_b128 + one global float4 load per 12 MFMAs (operand traffic)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PATTERN>
__global__ __launch_bounds__(512) void rate_k(const float* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = src[lane + 64 * i]; b[i] = src[512 + lane + 64 * i]; }
  if (PATTERN == 0 || PATTERN == 3) {
    f32x4 acc[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (PATTERN == 3) {      // operand traffic of the real loop: per (t) step ~ (2*3 LDS b128 + 2*4 global b128) / 8
          const float4 v = *reinterpret_cast<const float4*>(&lds[((it * 8 + t) * 64 + lane) * 4 & 4092]);
          a[t] = v.x;
          if ((t & 1) == 0) { const float4 w = *reinterpret_cast<const float4*>(&src[(((it * 8 + t) * 64 + lane) * 4) & 4092]); b[t] = w.y; }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int r = 0; r < 3; ++r)
            acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(t + r) & 7], b[(t + c) & 7], acc[r][c], 0, 0, 0);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) s += acc[r][c][0] + acc[r][c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else {
    constexpr int NA = PATTERN == 1 ? 2 : 4;
    f32x16 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int i = 0; i < NA; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(t + i) & 7], b[t], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
}

template <int PATTERN>
static void run(int waves, const float* src, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_k<PATTERN>, dim3(256), dim3(64 * waves), 0, 0, src, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate_k<PATTERN>, dim3(256), dim3(64 * waves), 0, 0, src, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_wave = (double)iters * 8 * (PATTERN == 0 || PATTERN == 3 ? 12 : (PATTERN == 1 ? 2 : 4));
  const double flop_per = (PATTERN == 0 || PATTERN == 3) ? 16.0 * 16 * 4 * 2 : 32.0 * 32 * 2 * 2;
  const double tf = mfma_per_wave * flop_per * 256 * waves / (ms * 1e-3) / 1e12;
  printf("pattern %d  %d waves/WG (256 WGs): %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 157.3)\n", PATTERN, waves, ms, tf, 100 * tf / 157.3);
}

int main() {
  float *src, *out;
  hipMalloc(&src, 1 << 16); hipMemset(src, 0, 1 << 16);
  hipMalloc(&out, 256 * 512 * 4);
  for (int waves : {4, 8}) { run<0>(waves, src, out); run<1>(waves, src, out); run<2>(waves, src, out); run<3>(waves, src, out); }
  return 0;
}
