// Skinny Linear kernels: the first and last encoder layers (n -> 10n and 10n -> n with n = 10:
// K = 10 or N = 10) are 0.04 % of the FLOPs of get_mlp (/root/reference/encoders.py:36-48,
// dims main_mlp.py:297-307) but cost a full 128-wide MFMA tile each when routed through the GEMM
// template (92 % padding, one launch-bound workgroup round).  They are HBM-bound streaming ops, so
// they get VALU kernels sized to their bytes:
//   small_contraction_k : out[m,c] = epi(sum_{q<Q} A[m,q] B(q,c)), Q <= 16
//                         fwd with K <= 16 (B(q,c) = W[c][q], bias + LeakyReLU epilogue)
//                         dgrad with N <= 16 (B(q,c) = W[q][c], act' epilogue)
//   small_output_k      : out[m,j] = sum_k X[m,k] W[j,k] + b[j], N <= 16 (fwd of the last layer)
// (wgrad of these layers stays on the split-K MFMA path: with the parallel slab reduction it measured
//  faster than a VALU outer-product kernel, 22 vs 32 us.)
// fp32 FMA in a fixed order; results differ from the MFMA path only by summation order.
#include "common.h"

namespace clica {
namespace skinny {

constexpr int THREADS = 256;
constexpr int MAXQ = 16;

// ---- out[m, c..c+3] = epi(sum_q A[m,q] * B[q][c..c+3]) ------------------------------------------
// B is staged in LDS as [Q][C4] (C4 = C rounded up to 4).  TRANSPOSED: B(q,c) = W[c*ldw + q].
template <bool TRANSPOSED, bool DACT>
__global__ __launch_bounds__(THREADS) void small_contraction_k(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ W, int64_t ldw,
    const float* __restrict__ bias, const float* __restrict__ xact, int64_t ldxa, float slope, int leaky,
    float* __restrict__ out, int64_t ldo, int64_t M, int C, int Q) {
  extern __shared__ __attribute__((aligned(16))) float bs[];   // [Q][C4]
  const int C4 = (C + 3) & ~3;
  for (int idx = threadIdx.x; idx < Q * C4; idx += THREADS) {
    const int q = idx / C4, c = idx - q * C4;
    float v = 0.f;
    if (c < C) v = TRANSPOSED ? W[(int64_t)c * ldw + q] : W[(int64_t)q * ldw + c];
    bs[idx] = v;
  }
  __syncthreads();
  const int cq = C4 / 4;                                   // float4 column groups per row
  const int64_t total = M * cq;
  for (int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * THREADS) {
    const int64_t m = idx / cq;
    const int c = (int)(idx - m * cq) * 4;
    float a[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) a[q] = (q < Q) ? A[m * lda + q] : 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
      if (q < Q) {
        const float4 b = *reinterpret_cast<const float4*>(&bs[q * C4 + c]);
        acc.x = fmaf(a[q], b.x, acc.x); acc.y = fmaf(a[q], b.y, acc.y);
        acc.z = fmaf(a[q], b.z, acc.z); acc.w = fmaf(a[q], b.w, acc.w);
      }
    }
    float r[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (c + u < C) {
        float v = r[u];
        if (DACT) {
          if (xact) v *= (xact[m * ldxa + c + u] > 0.f ? 1.f : slope);
        } else {
          if (bias) v += bias[c + u];
          if (leaky) v = v > 0.f ? v : v * slope;
        }
        out[m * ldo + c + u] = v;
      }
    }
  }
}

// ---- out[m, 0..N) = act(sum_k X[m,k] W[j,k] + b[j]),  N <= 16 -----------------------------------------
// 16 lanes cooperate on one row: lane q takes k = 4q..4q+3 (+64 per round), so a row is read as
// contiguous 256-byte pieces (coalesced) and there are 16x more threads in flight than rows; the N
// partial dot products are folded across the 16 lanes with xor-shuffles (fixed order).
__global__ __launch_bounds__(THREADS) void small_output_k(
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ W, int64_t ldw,
    const float* __restrict__ bias, float slope, int leaky, float* __restrict__ out, int64_t ldo,
    int64_t M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) float ws[];   // [N][K4]
  const int K4 = (K + 3) & ~3;
  for (int idx = threadIdx.x; idx < N * K4; idx += THREADS) {
    const int j = idx / K4, k = idx - j * K4;
    ws[idx] = (k < K) ? W[(int64_t)j * ldw + k] : 0.f;
  }
  __syncthreads();
  const int q = threadIdx.x & 15;
  const int64_t m = (int64_t)blockIdx.x * (THREADS / 16) + (threadIdx.x >> 4);
  const bool live = m < M;
  const float* x = X + (live ? m : 0) * ldx;
  const bool vec = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && (ldx % 4 == 0);
  float acc[MAXQ];
#pragma unroll
  for (int j = 0; j < MAXQ; ++j) acc[j] = 0.f;
  for (int k = 4 * q; k < K4; k += 64) {
    float4 xv;
    if (vec && k + 3 < K) xv = *reinterpret_cast<const float4*>(x + k);
    else xv = make_float4(k < K ? x[k] : 0.f, k + 1 < K ? x[k + 1] : 0.f, k + 2 < K ? x[k + 2] : 0.f, k + 3 < K ? x[k + 3] : 0.f);
#pragma unroll
    for (int j = 0; j < MAXQ; ++j) {
      if (j < N) {
        const float4 w = *reinterpret_cast<const float4*>(&ws[j * K4 + k]);
        acc[j] = fmaf(xv.x, w.x, fmaf(xv.y, w.y, fmaf(xv.z, w.z, fmaf(xv.w, w.w, acc[j]))));
      }
    }
  }
#pragma unroll
  for (int j = 0; j < MAXQ; ++j) {
    if (j < N) {
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) acc[j] += __shfl_xor(acc[j], off, 64);
    }
  }
  if (live && q < N) {
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < MAXQ; ++j) if (j == q) v = acc[j];
    v += bias ? bias[q] : 0.f;
    if (leaky) v = v > 0.f ? v : v * slope;
    out[m * ldo + q] = v;
  }
}

}  // namespace skinny

// ---- routing helpers used by linear.hip ----------------------------------------------------------------
bool skinny_fwd(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                int64_t M, int64_t N, int64_t K, int leaky, float slope, hipStream_t st) {
  using namespace skinny;
  if (K <= MAXQ && N <= 4096) {
    const int C4 = ((int)N + 3) & ~3;
    int64_t blocks = ceil_div(M * (C4 / 4), THREADS);
    if (blocks > kNumCU * 8) blocks = kNumCU * 8;
    hipLaunchKernelGGL((small_contraction_k<true, false>), dim3((unsigned)blocks), dim3(THREADS), (size_t)K * C4 * sizeof(float), st,
                       X, ldx, W, ldw, bias, (const float*)nullptr, (int64_t)0, slope, leaky, Y, ldy, M, (int)N, (int)K);
    return true;
  }
  if (N <= MAXQ && K <= 2048) {
    const int K4 = ((int)K + 3) & ~3;
    hipLaunchKernelGGL(small_output_k, dim3((unsigned)ceil_div(M, THREADS / 16)), dim3(THREADS), (size_t)N * K4 * sizeof(float), st,
                       X, ldx, W, ldw, bias, slope, leaky, Y, ldy, M, (int)N, (int)K);
    return true;
  }
  return false;
}

bool skinny_dgrad(const float* dY, int64_t lddy, const float* W, int64_t ldw, const float* Xact, int64_t ldxa, float slope,
                  float* dX, int64_t lddx, int64_t M, int64_t N, int64_t K, hipStream_t st) {
  using namespace skinny;
  if (N <= MAXQ && K <= 4096) {     // contraction over the (small) layer output width
    const int C4 = ((int)K + 3) & ~3;
    int64_t blocks = ceil_div(M * (C4 / 4), THREADS);
    if (blocks > kNumCU * 8) blocks = kNumCU * 8;
    hipLaunchKernelGGL((small_contraction_k<false, true>), dim3((unsigned)blocks), dim3(THREADS), (size_t)N * C4 * sizeof(float), st,
                       dY, lddy, W, ldw, (const float*)nullptr, Xact, ldxa, slope, 0, dX, lddx, M, (int)K, (int)N);
    return true;
  }
  return false;
}

}  // namespace clica
