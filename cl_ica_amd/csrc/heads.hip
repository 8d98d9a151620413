// Output-normalisation heads of the MLP encoder: RescaleLayer (mode "eq",
// /root/reference/layers.py:63-66) and SoftclipLayer (layers.py:87-91), forward + backward.
// HBM-bound streaming kernels over the (M, n) embedding block, one thread per row.
#include "common.h"

namespace clica {
namespace heads {
constexpr int THREADS = 256;

__device__ __forceinline__ float block_sum(float v) {
  __shared__ float red[THREADS / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < THREADS / 64; ++w) t += red[w];
  return t;
}

__global__ __launch_bounds__(THREADS) void rescale_fwd_k(const float* __restrict__ X, int64_t ldx, const float* __restrict__ r,
                                                        float* __restrict__ Y, int64_t ldy, float* __restrict__ inv_norm,
                                                        int64_t M, int n) {
  const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  if (i >= M) return;
  float ss = 0.f;
  for (int k = 0; k < n; ++k) { const float v = X[i * ldx + k]; ss += v * v; }
  const float inv = 1.f / sqrtf(ss);
  if (inv_norm) inv_norm[i] = inv;
  const float sc = inv * r[0];
  for (int k = 0; k < n; ++k) Y[i * ldy + k] = X[i * ldx + k] * sc;
}

// y = r u, u = x/|x|:  dx = r (dy - u <dy,u>) / |x|,  dr = sum <dy,u>
__global__ __launch_bounds__(THREADS) void rescale_bwd_k(const float* __restrict__ X, int64_t ldx, const float* __restrict__ r,
                                                        const float* __restrict__ inv_norm, const float* __restrict__ dY, int64_t lddy,
                                                        float* __restrict__ dX, int64_t lddx, float* __restrict__ dr_partial,
                                                        int64_t M, int n) {
  const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  float dot = 0.f;
  if (i < M) {
    const float inv = inv_norm[i];
    for (int k = 0; k < n; ++k) dot += dY[i * lddy + k] * X[i * ldx + k];
    dot *= inv;  // <dy, u>
    if (dX) {
      const float rr = r[0];
      for (int k = 0; k < n; ++k) {
        const float u = X[i * ldx + k] * inv;
        dX[i * lddx + k] = rr * (dY[i * lddy + k] - u * dot) * inv;
      }
    }
  }
  if (dr_partial) {
    const float t = block_sum(dot);
    if (threadIdx.x == 0) dr_partial[blockIdx.x] = t;
  }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

__global__ __launch_bounds__(THREADS) void softclip_fwd_k(const float* __restrict__ X, int64_t ldx, const float* __restrict__ bound,
                                                         float* __restrict__ Y, int64_t ldy, int64_t M, int n) {
  const int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  if (idx >= M * n) return;
  const int64_t i = idx / n; const int k = (int)(idx - i * n);
  Y[i * ldy + k] = sigmoidf_(X[i * ldx + k]) * bound[k];
}

// one block = 256 rows; dbound_partial[block][k] = sum_rows dy * sigmoid(x)
__global__ __launch_bounds__(THREADS) void softclip_bwd_k(const float* __restrict__ X, int64_t ldx, const float* __restrict__ bound,
                                                         const float* __restrict__ dY, int64_t lddy, float* __restrict__ dX, int64_t lddx,
                                                         float* __restrict__ dbound_partial, int64_t M, int n) {
  const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  for (int k = 0; k < n; ++k) {
    float contrib = 0.f;
    if (i < M) {
      const float s = sigmoidf_(X[i * ldx + k]);
      const float g = dY[i * lddy + k];
      contrib = g * s;
      if (dX) dX[i * lddx + k] = g * bound[k] * s * (1.f - s);
    }
    if (dbound_partial) {
      const float t = block_sum(contrib);
      if (threadIdx.x == 0) dbound_partial[(int64_t)blockIdx.x * n + k] = t;
    }
  }
}

// stand-alone LeakyReLU (main_3dident.py:368: the activation BETWEEN the backbone output and the head's Linear, so it
// cannot ride in a GEMM epilogue).  Backward reads the saved OUTPUT (slope > 0: sign(y) = sign(x)).
__global__ __launch_bounds__(THREADS) void leaky_fwd_k(const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy,
                                                      int64_t M, int n, float slope) {
  const int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  if (idx >= M * n) return;
  const int64_t i = idx / n; const int k = (int)(idx - i * n);
  const float v = X[i * ldx + k];
  Y[i * ldy + k] = v > 0.f ? v : slope * v;
}
__global__ __launch_bounds__(THREADS) void leaky_bwd_k(const float* __restrict__ Yact, int64_t ldy, const float* __restrict__ dY, int64_t lddy,
                                                      float* __restrict__ dX, int64_t lddx, int64_t M, int n, float slope) {
  const int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  if (idx >= M * n) return;
  const int64_t i = idx / n; const int k = (int)(idx - i * n);
  const float g = dY[i * lddy + k];
  dX[i * lddx + k] = Yact[i * ldy + k] > 0.f ? g : slope * g;
}
}  // namespace heads
}  // namespace clica

using namespace clica;
using namespace clica::heads;

extern "C" int clica_rescale_fwd(const float* X, int64_t ldx, const float* r, float* Y, int64_t ldy,
                                 float* inv_norm, int64_t M, int32_t n, clica_stream_t stream) {
  CLICA_CHECK_ARG(X && r && Y && M > 0 && n > 0 && ldx >= n && ldy >= n, "clica_rescale_fwd: bad argument");
  hipLaunchKernelGGL(rescale_fwd_k, dim3((unsigned)ceil_div(M, THREADS)), dim3(THREADS), 0, as_stream(stream), X, ldx, r, Y, ldy, inv_norm, M, n);
  return launch_status("clica_rescale_fwd");
}
extern "C" int clica_rescale_bwd(const float* X, int64_t ldx, const float* r, const float* inv_norm,
                                 const float* dY, int64_t lddy, float* dX, int64_t lddx,
                                 float* dr_partial, int64_t M, int32_t n, clica_stream_t stream) {
  CLICA_CHECK_ARG(X && r && inv_norm && dY && M > 0 && n > 0 && ldx >= n && lddy >= n && (!dX || lddx >= n), "clica_rescale_bwd: bad argument");
  hipLaunchKernelGGL(rescale_bwd_k, dim3((unsigned)ceil_div(M, THREADS)), dim3(THREADS), 0, as_stream(stream), X, ldx, r, inv_norm, dY, lddy, dX, lddx, dr_partial, M, n);
  return launch_status("clica_rescale_bwd");
}
extern "C" int clica_softclip_fwd(const float* X, int64_t ldx, const float* bound, float* Y, int64_t ldy,
                                  int64_t M, int32_t n, clica_stream_t stream) {
  CLICA_CHECK_ARG(X && bound && Y && M > 0 && n > 0 && ldx >= n && ldy >= n, "clica_softclip_fwd: bad argument");
  hipLaunchKernelGGL(softclip_fwd_k, dim3((unsigned)ceil_div(M * n, THREADS)), dim3(THREADS), 0, as_stream(stream), X, ldx, bound, Y, ldy, M, n);
  return launch_status("clica_softclip_fwd");
}
extern "C" int clica_softclip_bwd(const float* X, int64_t ldx, const float* bound, const float* dY, int64_t lddy,
                                  float* dX, int64_t lddx, float* dbound_partial, int64_t M, int32_t n,
                                  clica_stream_t stream) {
  CLICA_CHECK_ARG(X && bound && dY && M > 0 && n > 0 && ldx >= n && lddy >= n && (!dX || lddx >= n), "clica_softclip_bwd: bad argument");
  hipLaunchKernelGGL(softclip_bwd_k, dim3((unsigned)ceil_div(M, THREADS)), dim3(THREADS), 0, as_stream(stream), X, ldx, bound, dY, lddy, dX, lddx, dbound_partial, M, n);
  return launch_status("clica_softclip_bwd");
}
extern "C" int clica_leaky_relu_fwd(const float* X, int64_t ldx, float* Y, int64_t ldy, int64_t M, int32_t n, float slope,
                                    clica_stream_t stream) {
  CLICA_CHECK_ARG(X && Y && M > 0 && n > 0 && ldx >= n && ldy >= n, "clica_leaky_relu_fwd: bad argument");
  hipLaunchKernelGGL(leaky_fwd_k, dim3((unsigned)ceil_div(M * n, THREADS)), dim3(THREADS), 0, as_stream(stream), X, ldx, Y, ldy, M, n, slope);
  return launch_status("clica_leaky_relu_fwd");
}
extern "C" int clica_leaky_relu_bwd(const float* Yact, int64_t ldy, const float* dY, int64_t lddy, float* dX, int64_t lddx,
                                    int64_t M, int32_t n, float slope, clica_stream_t stream) {
  CLICA_CHECK_ARG(Yact && dY && dX && M > 0 && n > 0 && ldy >= n && lddy >= n && lddx >= n, "clica_leaky_relu_bwd: bad argument");
  // slope = 0 is ReLU: its output is > 0 exactly where its derivative is 1, so the saved output still carries the gate
  CLICA_CHECK_ARG(slope >= 0.f, "clica_leaky_relu_bwd: slope=%g must be >= 0 (the backward recovers the gate from the output)", slope);
  hipLaunchKernelGGL(leaky_bwd_k, dim3((unsigned)ceil_div(M * n, THREADS)), dim3(THREADS), 0, as_stream(stream), Yact, ldy, dY, lddy, dX, lddx, M, n, slope);
  return launch_status("clica_leaky_relu_bwd");
}
