// PROTOTYPE (not part of libclica_hip.so): one Linear layer Y = X W^T with fp32-grade accuracy on the bf16 matrix
// cores -- both fp32 operands are split EXACTLY into three bf16 pieces (8 + 8 + 8 mantissa bits) and the six
// products of order <= 2 (hi.hi, hi.mid, mid.hi, hi.lo, mid.mid, lo.hi) are accumulated in fp32 by
// v_mfma_f32_16x16x32_bf16.  Same decomposition as the fused encoder kernels: a workgroup owns 48 rows, the
// activation panel sits in LDS (three bf16 planes), weights stream from L2 in fragment order.
// Purpose: measure what this arithmetic would buy before committing the product kernels to it.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWS = 48, RB = 3, WAVES = 8, THREADS = 512, CBW = 4, KI = 32;
constexpr int LDPB = 520;                       // bf16 elements per panel row (1040 B: 16-byte aligned rows)
constexpr int PLANE = ROWS * LDPB;              // bf16 elements per plane

__device__ __forceinline__ void split3(float v, unsigned short& h, unsigned short& m, unsigned short& l) {
  const unsigned hb = __float_as_uint(v) & 0xFFFF0000u;
  const float r1 = v - __uint_as_float(hb);
  const unsigned mb = __float_as_uint(r1) & 0xFFFF0000u;
  const float r2 = r1 - __uint_as_float(mb);
  h = (unsigned short)(hb >> 16); m = (unsigned short)(mb >> 16); l = (unsigned short)(__float_as_uint(r2) >> 16);
}

// Wp: [piece 3][cb][ki][lane 64] x 16 B  (lane (n = cb*16 + (lane&15), kg = lane>>4) holds W_piece[n][ki*32 + kg*8 .. +7])
extern "C" __global__ __launch_bounds__(THREADS) void bf16x3_layer_k(const float* __restrict__ X, int64_t ldx, int64_t M, int K, int N,
                                                                    const u32x4* __restrict__ Wp, float* __restrict__ Y, int64_t ldy,
                                                                    int products, int repeat) {
  extern __shared__ __attribute__((aligned(16))) unsigned short planes[];   // [3][ROWS][LDPB]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i15 = lane & 15, kg = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int Kp = (K + KI - 1) / KI * KI, kiters = Kp / KI, ncb = (N + 15) / 16;
  for (int idx = threadIdx.x; idx < ROWS * Kp; idx += THREADS) {
    const int r = idx / Kp, k = idx - r * Kp;
    const float v = (row0 + r < M && k < K) ? X[(row0 + r) * ldx + k] : 0.f;
    unsigned short h, m, l; split3(v, h, m, l);
    planes[r * LDPB + k] = h; planes[PLANE + r * LDPB + k] = m; planes[2 * PLANE + r * LDPB + k] = l;
  }
  __syncthreads();
  f32x4 acc[RB][CBW];
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int c = 0; c < CBW; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int64_t piece_stride = (int64_t)ncb * kiters * 64;
  u32x4 wcur[3][CBW], wnxt[3][CBW];
  auto fetch_w = [&](u32x4 (&w)[3][CBW], int ki) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int c = 0; c < CBW; ++c) {
        const int cb = wave + c * WAVES;
        w[p][c] = Wp[p * piece_stride + ((int64_t)(cb < ncb ? cb : 0) * kiters + ki) * 64 + lane];
      }
  };
  fetch_w(wcur, 0);
  for (int kk = 0; kk < kiters * repeat; ++kk) {     // repeat > 1: the same layer again (marginal per-layer cost probe)
    const int ki = kk % kiters;
    fetch_w(wnxt, (kk + 1) % kiters);
    u32x4 x[3][RB];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int r = 0; r < RB; ++r)
        x[p][r] = *reinterpret_cast<const u32x4*>(&planes[p * PLANE + (r * 16 + i15) * LDPB + ki * KI + kg * 8]);
    // (weight piece, activation piece), small terms first
    constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      if (t < 6 - products) continue;         // products = 6: all; 3: orders 0..1 only; 1: hi.hi only (speed probes)
#pragma unroll
      for (int c = 0; c < CBW; ++c)
#pragma unroll
        for (int r = 0; r < RB; ++r)
          acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wcur[PW[t]][c]), __builtin_bit_cast(bf16x8, x[PX[t]][r]),
                                                              acc[r][c], 0, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int c = 0; c < CBW; ++c) wcur[p][c] = wnxt[p][c];
  }
  // D[n][m]: lane holds batch row m = rb*16 + (lane & 15) and output features n = cb*16 + (lane >> 4)*4 + e
#pragma unroll
  for (int c = 0; c < CBW; ++c) {
    const int cb = wave + c * WAVES;
    if (cb >= ncb) continue;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int64_t row = row0 + r * 16 + i15;
      const int n0 = cb * 16 + kg * 4;
      if (row < M && n0 + 3 < N) *reinterpret_cast<f32x4*>(&Y[row * ldy + n0]) = acc[r][c];
    }
  }
}

extern "C" int bf16x3_layer(const float* X, int64_t ldx, int64_t M, int K, int N, const void* Wp, float* Y, int64_t ldy, int products,
                            int repeat, void* stream) {
  constexpr size_t lds = 3 * (size_t)PLANE * sizeof(unsigned short);
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(bf16x3_layer_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
  (void)once;
  hipLaunchKernelGGL(bf16x3_layer_k, dim3((unsigned)((M + ROWS - 1) / ROWS)), dim3(THREADS), lds, (hipStream_t)stream, X, ldx, M, K, N,
                     (const u32x4*)Wp, Y, ldy, products, repeat);
  return (int)hipGetLastError();
}
