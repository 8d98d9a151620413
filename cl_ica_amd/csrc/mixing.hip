// Mixing network g: the frozen nn.Sequential that construct_invertible_mlp returns
// (/root/reference/invertible_network_utils.py:87-115) -- n x n bias-free Linear layers with
// LeakyReLU(slope) between them, forward only.  All layers run in ONE kernel: weights and the
// block's rows live in LDS, thread (row, j) produces output coordinate j of its row per layer.
#include "common.h"

namespace clica {
namespace mixing {
constexpr int THREADS = 256;

// hidden activation of the mixing net (invertible_network_utils.py:44-66): 0 LeakyReLU(a) [a = 0: ReLU], 1 ELU(alpha = a),
// 2 SmoothLeakyReLU: a x + (1 - a) log(1 + e^x), 3 Softplus(beta = a, threshold 20)
__device__ __forceinline__ float mix_act(float v, int kind, float a) {
  switch (kind) {
    case 1: return v > 0.f ? v : a * expm1f(v);
    case 2: return a * v + (1.f - a) * logf(1.f + expf(v));
    case 3: return a * v > 20.f ? v : log1pf(expf(a * v)) / a;
    default: return v > 0.f ? v : v * a;
  }
}

__global__ __launch_bounds__(THREADS) void mixing_fwd_k(const float* __restrict__ Z, int64_t ldz, const float* __restrict__ W,
                                                       int n_layers, int act_kind, float slope, float* __restrict__ X, int64_t ldx,
                                                       int64_t M, int n, int rows_per_block) {
  extern __shared__ float sm[];
  float* w = sm;                               // [n_layers][n][n]
  float* xa = sm + n_layers * n * n;           // [rows_per_block][n]
  float* xb = xa + rows_per_block * n;
  for (int idx = threadIdx.x; idx < n_layers * n * n; idx += THREADS) w[idx] = W[idx];
  const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
  for (int idx = threadIdx.x; idx < rows_per_block * n; idx += THREADS) {
    const int r = idx / n, k = idx - r * n;
    xa[idx] = (row0 + r < M) ? Z[(row0 + r) * ldz + k] : 0.f;
  }
  __syncthreads();
  for (int l = 0; l < n_layers; ++l) {
    const float* wl = w + l * n * n;
    for (int idx = threadIdx.x; idx < rows_per_block * n; idx += THREADS) {
      const int r = idx / n, j = idx - r * n;
      float acc = 0.f;
      for (int k = 0; k < n; ++k) acc = fmaf(xa[r * n + k], wl[j * n + k], acc);
      if (l < n_layers - 1) acc = mix_act(acc, act_kind, slope);
      xb[idx] = acc;
    }
    __syncthreads();
    float* t = xa; xa = xb; xb = t;
  }
  for (int idx = threadIdx.x; idx < rows_per_block * n; idx += THREADS) {
    const int r = idx / n, k = idx - r * n;
    if (row0 + r < M) X[(row0 + r) * ldx + k] = xa[idx];
  }
}
}  // namespace mixing
}  // namespace clica

using namespace clica;

extern "C" int clica_mixing_fwd_act(const float* Z, int64_t ldz, const float* W, int32_t n_layers, int32_t act_kind, float act_param,
                                    float* X, int64_t ldx, int64_t M, int32_t n, clica_stream_t stream);
extern "C" int clica_mixing_fwd(const float* Z, int64_t ldz, const float* W, int32_t n_layers, float slope,
                                float* X, int64_t ldx, int64_t M, int32_t n, clica_stream_t stream) {
  return clica_mixing_fwd_act(Z, ldz, W, n_layers, 0, slope, X, ldx, M, n, stream);
}
extern "C" int clica_mixing_fwd_act(const float* Z, int64_t ldz, const float* W, int32_t n_layers, int32_t act_kind, float slope,
                                    float* X, int64_t ldx, int64_t M, int32_t n, clica_stream_t stream) {
  CLICA_CHECK_ARG(Z && W && X && M > 0 && n > 0 && n_layers > 0 && ldz >= n && ldx >= n, "clica_mixing_fwd: bad argument");
  CLICA_CHECK_ARG(act_kind >= 0 && act_kind <= 3, "clica_mixing_fwd_act: activation kind %d (0 leaky/relu, 1 elu, 2 smooth_leaky_relu, 3 softplus)", act_kind);
  int rows = 256 / n; if (rows < 1) rows = 1; if (rows > 64) rows = 64;
  const size_t lds = ((size_t)n_layers * n * n + 2 * (size_t)rows * n) * sizeof(float);
  CLICA_CHECK_ARG(lds <= 64 * 1024, "clica_mixing_fwd: n_layers*n*n = %d floats do not fit LDS", n_layers * n * n);
  hipLaunchKernelGGL(mixing::mixing_fwd_k, dim3((unsigned)ceil_div(M, rows)), dim3(mixing::THREADS), lds, as_stream(stream),
                     Z, ldz, W, n_layers, act_kind, slope, X, ldx, M, n, rows);
  return launch_status("clica_mixing_fwd");
}
