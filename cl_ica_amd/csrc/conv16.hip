// Convolution stack of the KITTI-masks encoder (BetaVAE_H, /root/reference/kitti_masks/model.py:41-56) in the f16x2 split
// arithmetic -- the implicit GEMMs of linear.hip's conv section (same layouts: channels-last, padded space-to-depth inputs,
// scattering epilogues, gate bits) with the fp32 matrix instructions replaced by three fp16 ones per product.
//
// Arithmetic (the encoder's f16x2, fused_mlp.hip / DESIGN 4.1): an fp32 operand v of a tensor with scale s (a power of two that
// puts the tensor's largest magnitude into [256, 512)) is hi = RN_f16(v s), lo = RN_f16(v s - hi); a product is hi.hi + hi.lo +
// lo.hi accumulated in fp32 by v_mfma_f32_32x32x16_f16 (lo.lo < 2^-22 |a||b| dropped), and the epilogue divides the two scales out.
// One 8-pass fp16 instruction covers K = 16 where the fp32 path needs eight 16-pass ones: 3/16 of the matrix time.
//
// What differs from the encoder's use of it: the operands stay fp32 in HBM (the same 4 B per element the planes would take) and are
// split ON THE WAY INTO LDS by the consuming kernel, so the scale in force is the one of THIS step: every producer leaves the
// maximum of what it stored in a 256-slot array (one atomicMax per wave, <= 64 per address), the consumer reduces the slots in its
// prologue.  No state carried between steps, nothing to calibrate, no overflow to flag (the scaled maximum is < 512 by construction).
// The small weight matrices are split once per step by the pack launch (which also re-orders them from nn.Conv2d's layout).
//
//   clica_conv16_pack        Conv2d weights -> [hi plane][lo plane] f16 GEMM operands [N][K] + their scales (one launch, <= 8 matrices)
//   clica_conv16_amax        |x| maximum of a tensor into a slot array (for tensors a non-conv16 kernel produced)
//   clica_conv16_k4s2_fwd    forward of a stage  (bias + ReLU + scatter into the next stage's input + gate bits + maximum)
//   clica_conv16_k4s2_dgrad  data gradient of a stage (ReLU gate from the bits + scatter + maximum)
//   clica_conv16_k4s2_wgrad  weight / bias gradient of a stage (fp32 slabs over contraction splits + a deterministic reduction)
#include "common.h"
#include <algorithm>
#include <stdlib.h>

namespace clica {
namespace conv16 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short half_t;                         // storage type of an f16 piece
typedef __attribute__((address_space(3))) char* lds_ptr;

constexpr int kSlots = 256;                            // maxima slots per tensor
constexpr int kTarget = 8;                             // scaled maximum in [2^8, 2^9)
constexpr int BK = 32;                                 // contraction elements per LDS tile (two 16-deep matrix steps)
constexpr int LDH = BK + 8;                            // f16 elements per LDS row: 80 B -- conflict-free 16-byte fragment reads

__device__ __forceinline__ int xcd_contiguous(int b, int nwg) {        // linear.hip: each XCD walks a contiguous range of work items
  const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// power-of-two scale for a tensor whose largest magnitude has the bits `bits` (0: all-zero tensor -> 1)
__device__ __forceinline__ float scale_of(unsigned bits) {
  if (bits == 0u) return 1.f;
  int e = (int)((bits >> 23) & 0xffu) - 127;
  int se = kTarget - e;
  se = se < -100 ? -100 : (se > 100 ? 100 : se);
  return __uint_as_float((unsigned)(se + 127) << 23);
}
// all threads of the workgroup: the tensor's scale from its slot array (nullptr: 1).  `word` = one LDS word.
__device__ __forceinline__ float tensor_scale(const unsigned* __restrict__ slots, unsigned* word) {
  if (!slots) return 1.f;
  __syncthreads();                                     // (a previous call's readers of `word` are done)
  if (threadIdx.x == 0) *word = 0u;
  unsigned m = 0u;
  for (int i = threadIdx.x; i < kSlots; i += blockDim.x) m = max(m, slots[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0 && m) atomicMax(word, m);
  __syncthreads();
  return scale_of(*word);
}
// one call per wave at the end of a producer: the wave's maximum of |stored value| into the tensor's slots
__device__ __forceinline__ void commit_amax(unsigned* __restrict__ slots, float m, unsigned wave_id) {
  if (!slots) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) {
    m = (m <= 3.0e38f) ? m : 3.4e38f;                 // NaN / inf -> a huge finite value
    if (m > 0.f) atomicMax(slots + (wave_id & (kSlots - 1)), __float_as_uint(m));
  }
}

// four scaled fp32 values -> their hi and lo pieces (two packed registers each)
__device__ __forceinline__ void split4(const float4 v, const float s, u32x2& hi, u32x2& lo) {
  const f32x2v a = {v.x * s, v.y * s}, b = {v.z * s, v.w * s};
  const f16x2v ha = __builtin_convertvector(a, f16x2v), hb = __builtin_convertvector(b, f16x2v);
  const f32x2v ra = a - __builtin_convertvector(ha, f32x2v), rb = b - __builtin_convertvector(hb, f32x2v);
  hi = (u32x2){__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
  lo = (u32x2){__builtin_bit_cast(unsigned, __builtin_convertvector(ra, f16x2v)), __builtin_bit_cast(unsigned, __builtin_convertvector(rb, f16x2v))};
}
__device__ __forceinline__ f32x16 mfma(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// ---- weight pack ----------------------------------------------------------------------------------------------------------------
constexpr int MAXPACK = 8;
struct PackArgs { int n; const float* src[MAXPACK]; const int* map[MAXPACK]; half_t* dst[MAXPACK]; int count[MAXPACK]; int src_count[MAXPACK]; float* scale; };
// PACK_CHUNKS workgroups of 1024 threads per matrix: each takes the maximum of the (small) source tensor for itself -- no second launch,
// no inter-workgroup traffic -- and writes its share of dst[plane][e] = piece(src[map[e]] * s).  (One workgroup per matrix, scalar loops:
// 69 us, a chain of dependent round trips.)
constexpr int PACK_CHUNKS = 16;
__global__ __launch_bounds__(1024) void pack_k(PackArgs P) {
  __shared__ unsigned word;
  const int q = blockIdx.x / PACK_CHUNKS, chunk = blockIdx.x % PACK_CHUNKS;
  const float* __restrict__ src = P.src[q];
  const int* __restrict__ map = P.map[q];
  if (threadIdx.x == 0) word = 0u;
  float m = 0.f;
  const int n4 = P.src_count[q] >> 2;
  for (int e0 = 0; e0 < n4; e0 += 4 * 1024) {          // four independent 16-byte loads in flight per thread
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int e = e0 + u * 1024 + threadIdx.x; v[u] = e < n4 ? reinterpret_cast<const float4*>(src)[e] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
    for (int u = 0; u < 4; ++u) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
  }
  for (int e = 4 * n4 + threadIdx.x; e < P.src_count[q]; e += 1024) m = fmaxf(m, fabsf(src[e]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) atomicMax(&word, __float_as_uint((m <= 3.0e38f) ? m : 3.4e38f));
  __syncthreads();
  const float s = scale_of(word);
  if (chunk == 0 && threadIdx.x == 0) P.scale[q] = s;
  half_t* __restrict__ hi = P.dst[q];
  half_t* __restrict__ lo = hi + P.count[q];
  const int per = (P.count[q] + PACK_CHUNKS - 1) / PACK_CHUNKS;
  const int e_end = min(P.count[q], (chunk + 1) * per);
  for (int e0 = chunk * per; e0 < e_end; e0 += 4 * 1024) {
    int mi[4]; float t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int e = e0 + u * 1024 + threadIdx.x; mi[u] = e < e_end ? (map ? map[e] : e) : -1; }
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = mi[u] >= 0 ? src[mi[u]] * s : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * 1024 + threadIdx.x;
      if (e >= e_end) continue;
      const _Float16 h = (_Float16)t[u];
      hi[e] = __builtin_bit_cast(half_t, h);
      lo[e] = __builtin_bit_cast(half_t, (_Float16)(t[u] - (float)h));
    }
  }
}

__global__ __launch_bounds__(256) void amax_k(const float* __restrict__ x, int64_t n4, unsigned* __restrict__ slots) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  commit_amax(slots, m, blockIdx.x * 4 + (threadIdx.x >> 6));
}
__global__ __launch_bounds__(256) void zero_slots_k(unsigned* __restrict__ slots, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) slots[i] = 0u;
}

// ---- forward / data gradient: C[M][N] = A[M][K] B[N][K]^T -----------------------------------------------------------------------
struct Geo {
  int mode;                  // 1: forward, scatter into the next stage's space-to-depth input; 3: forward, rows at their place on the storage
                             // grid; 2: data gradient, scatter back into the previous stage's output-gradient pixels (linear.hip: ConvX)
  int hs, ws;                // GEMM row r = (image * hs + y) * ws + x
  int ghs, gws;              // forward: the grid the A operand (and mode 3's output, the gate words) is stored on; rows run over its hs x ws outputs
  int dhs, dws, dho, dwo;    // destination pixel grid and its valid part
  int c;                     // channels per pixel of the scattered tensor (mode 1: N, mode 2: N / 4)
  float inv_pix, inv_ws;
};
struct GArgs {
  const float* A; int64_t lda; int seg, jump;
  const half_t* B;                                   // [2][N][K]
  const float* bscale;                               // device: scale of B
  const unsigned* amax_in; unsigned* amax_out;
  float* C; const float* bias; int relu;
  int64_t M; int N, K;
  unsigned* gate_out; const unsigned* gate_in;
  Geo g;
};
#ifndef STREAM16_ABLATE
#define STREAM16_ABLATE 0      // measurement builds only (make EXTRA=-DSTREAM16_ABLATE=k, WRONG results): 1 = no stores, 2 = no matrix products in the data gradient
#endif                         // (round 5, 2048-mask batch: 155 us as built, 127 without stores, 144 without products, 108 without both)

__device__ __forceinline__ void row_to_pixel(const Geo& g, int row, int& img, int& y, int& x) {      // rows < 2^24: exact in fp32
  const int pix = g.hs * g.ws;
  img = (int)((float)row * g.inv_pix);
  int rem = row - img * pix;
  if (rem < 0) { --img; rem += pix; } else if (rem >= pix) { ++img; rem -= pix; }
  y = (int)((float)rem * g.inv_ws); x = rem - y * g.ws;
  if (x < 0) { --y; x += g.ws; } else if (x >= g.ws) { ++y; x -= g.ws; }
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void gemm16_k(GArgs a) {
  constexpr int THREADS = 64 * WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN, NBM = TM / 32, NBN = TN / 32;
  static_assert(TM % 32 == 0 && TN % 32 == 0, "wave tile must be a multiple of 32");
  constexpr int A_UNITS = BM * (BK / 4) / THREADS;                 // float4 per thread and tile
  constexpr int B_TOTAL = 2 * BN * (BK / 8);                       // 16-byte units of both planes
  constexpr int B_UNITS = (B_TOTAL + THREADS - 1) / THREADS;
  static_assert(BM * (BK / 4) % THREADS == 0, "A tile must divide over the workgroup");
  constexpr int A_PLANE = BM * LDH, B_PLANE = BN * LDH;            // f16 elements
  constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;
  extern __shared__ __attribute__((aligned(16))) half_t smem[];    // two stages
  __shared__ unsigned s_word;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN, h = lane >> 5, l31 = lane & 31;
  const int gx = (a.N + BN - 1) / BN;
  const int id = xcd_contiguous(blockIdx.x, gridDim.x);
  const int by = id / gx, bx = id - by * gx;
  const int m0 = by * BM, n0 = bx * BN;
  const Geo g = a.g;

  const float sA = tensor_scale(a.amax_in, &s_word);

  // per-thread source pointers, set up once
  const float* pa[A_UNITS];
  const int kq = threadIdx.x % (BK / 4);
#pragma unroll
  for (int i = 0; i < A_UNITS; ++i) {
    const int row = min(m0 + (threadIdx.x + i * THREADS) / (BK / 4), (int)a.M - 1);
    if (g.mode == 2) {
      pa[i] = a.A + (int64_t)row * a.lda + 4 * kq;
    } else {
      int img, y, x;
      row_to_pixel(g, row, img, y, x);
      pa[i] = a.A + (((int64_t)img * g.ghs + y) * g.gws + x) * a.lda + 4 * kq;
    }
  }
  const int thr = a.seg - 4 * kq;
  const half_t* pb[B_UNITS];
  int pb_lds[B_UNITS];
#pragma unroll
  for (int i = 0; i < B_UNITS; ++i) {
    const int v = threadIdx.x + i * THREADS;
    const int plane = v / (BN * (BK / 8)), rem = v - plane * (BN * (BK / 8)), n = rem / (BK / 8), k8 = rem % (BK / 8);
    const bool ok = v < B_TOTAL;
    pb[i] = ok ? a.B + (int64_t)plane * a.N * a.K + (int64_t)min(n0 + n, a.N - 1) * a.K + 8 * k8 : nullptr;
    pb_lds[i] = 2 * A_PLANE + plane * B_PLANE + n * LDH + 8 * k8;
  }
  float4 ra[A_UNITS];
  u32x4 rb[B_UNITS];
  auto load = [&](int k0) {
    const int o = k0 + (k0 >= thr ? a.jump : 0);
#pragma unroll
    for (int i = 0; i < A_UNITS; ++i) ra[i] = *reinterpret_cast<const float4*>(pa[i] + o);
#pragma unroll
    for (int i = 0; i < B_UNITS; ++i)
      if (pb[i]) rb[i] = *reinterpret_cast<const u32x4*>(pb[i] + k0);
  };
  auto store = [&](half_t* st) {
#pragma unroll
    for (int i = 0; i < A_UNITS; ++i) {
      const int row = (threadIdx.x + i * THREADS) / (BK / 4);
      u32x2 hi, lo;
      split4(ra[i], sA, hi, lo);
      *reinterpret_cast<u32x2*>(st + row * LDH + 4 * kq) = hi;
      *reinterpret_cast<u32x2*>(st + A_PLANE + row * LDH + 4 * kq) = lo;
    }
#pragma unroll
    for (int i = 0; i < B_UNITS; ++i)
      if (pb[i]) *reinterpret_cast<u32x4*>(st + pb_lds[i]) = rb[i];
  };

  f32x16 acc[NBM][NBN];
#pragma unroll
  for (int i = 0; i < NBM; ++i)
#pragma unroll
    for (int j = 0; j < NBN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int ntiles = a.K / BK;
  load(0);
  store(smem);
  if (ntiles > 1) load(BK);
  __syncthreads();

  u32x4 af[2][NBM][2], bf[2][NBN][2];               // [buffer][block][hi / lo]
  auto frags = [&](int buf, const half_t* st, int s) {
#pragma unroll
    for (int i = 0; i < NBM; ++i) {
      const half_t* p = st + (wm * TM + i * 32 + l31) * LDH + 16 * s + 8 * h;
      af[buf][i][0] = *reinterpret_cast<const u32x4*>(p);
      af[buf][i][1] = *reinterpret_cast<const u32x4*>(p + A_PLANE);
    }
#pragma unroll
    for (int j = 0; j < NBN; ++j) {
      const half_t* p = st + 2 * A_PLANE + (wn * TN + j * 32 + l31) * LDH + 16 * s + 8 * h;
      bf[buf][j][0] = *reinterpret_cast<const u32x4*>(p);
      bf[buf][j][1] = *reinterpret_cast<const u32x4*>(p + B_PLANE);
    }
  };
  frags(0, smem, 0);
  int cur = 0;
  for (int t = 0; t < ntiles; ++t) {
    const half_t* st = smem + cur * STAGE;
    half_t* nx = smem + (cur ^ 1) * STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (s == 0) frags(1, st, 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NBM; ++i)
#pragma unroll
        for (int j = 0; j < NBN; ++j) {
          acc[i][j] = mfma(af[s][i][1], bf[s][j][0], acc[i][j]);
          acc[i][j] = mfma(af[s][i][0], bf[s][j][1], acc[i][j]);
          acc[i][j] = mfma(af[s][i][0], bf[s][j][0], acc[i][j]);
        }
      if (s == 0) {
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < ntiles) store(nx);
        if (t + 2 < ntiles) load((t + 2) * BK);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    if (t + 1 < ntiles) frags(0, nx, 0);
    cur ^= 1;
  }

  // ---- epilogue.  C/D layout of a 32 x 32 block: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const float cs = 1.f / (sA * *a.bscale);
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NBM; ++i) {
#pragma unroll
    for (int j = 0; j < NBN; ++j) {
      const int col = n0 + wn * TN + j * 32 + l31;
      if (col >= a.N) continue;
      const float bv = a.bias ? a.bias[col] : 0.f;
      const int q = col / g.c, ch = col - q * g.c;
      int64_t off[16];
      unsigned gw[16];
      int grow[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        off[r] = -1; gw[r] = ~0u; grow[r] = 0;
        if (row >= a.M) continue;
        int img, y, x;
        row_to_pixel(g, row, img, y, x);
        if (g.mode == 3) {
          off[r] = (int64_t)((img * g.ghs + y) * g.gws + x) * a.N + col;
        } else if (g.mode == 1) {
          const int Y = (y + 1) >> 1, X = (x + 1) >> 1, qq = ((y + 1) & 1) * 2 + ((x + 1) & 1);
          off[r] = (((int64_t)img * g.dhs + Y) * g.dws + X) * (4 * g.c) + qq * g.c + col;
          grow[r] = (img * g.ghs + y) * g.gws + x;
        } else {
          const int yy = 2 * y + (q >> 1) - 1, xx = 2 * x + (q & 1) - 1;
          if (yy < 0 || xx < 0 || yy >= g.dho || xx >= g.dwo) continue;
          const int64_t pix = ((int64_t)img * g.dhs + yy) * g.dws + xx;
          off[r] = pix * g.c + ch;
          if (a.gate_in) gw[r] = a.gate_in[pix * (g.c >> 5) + (ch >> 5)];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (off[r] < 0) continue;
        float v = acc[i][j][r] * cs;
        if (g.mode == 2) {
          v = ((gw[r] >> (ch & 31)) & 1u) ? v : 0.f;
        } else {
          v += bv;
          if (a.relu) v = v > 0.f ? v : 0.f;
        }
        a.C[off[r]] = v;
        amax = fmaxf(amax, fabsf(v));
        if (g.mode == 1 && a.gate_out) {
          const unsigned long long bits = __ballot(v > 0.f);
          if (l31 == 0) a.gate_out[(int64_t)grow[r] * (a.N >> 5) + (col >> 5)] = (unsigned)(bits >> (32 * h));
        }
      }
    }
  }
  commit_amax(a.amax_out, amax, blockIdx.x * (WM * WN) + wave);
}

template <int BM, int BN, int WM, int WN>
static int launch_gemm16(const GArgs& a, hipStream_t st, const char* who) {
  constexpr int THREADS = 64 * WM * WN;
  constexpr size_t lds = 2 * (2 * BM + 2 * BN) * LDH * sizeof(half_t);
  auto k = gemm16_k<BM, BN, WM, WN>;
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
  (void)once;
  const int64_t nwg = ceil_div(a.M, BM) * ceil_div(a.N, BN);
  hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(THREADS), lds, st, a);
  return launch_status(who);
}
static int launch_gemm16_auto(const GArgs& a, hipStream_t st, const char* who) {
  if (a.N <= 32) return launch_gemm16<128, 32, 4, 1>(a, st, who);
  if (a.N <= 64) return a.M >= 128 * 2 * kNumCU ? launch_gemm16<128, 64, 2, 2>(a, st, who) : launch_gemm16<64, 64, 2, 2>(a, st, who);
  return launch_gemm16<64, 128, 2, 2>(a, st, who);
}

// ---- the same products as a STREAMING kernel ------------------------------------------------------------------------------------
// The stages' GEMMs are tall and shallow (592 k rows x K = 128 ... 33 k rows x K = 1024): in the tiled kernel above a workgroup's
// prologue (scale lookup, pointers), barriers and scattering epilogue outweigh its few k-tiles (measured: 222 us for the widest data
// gradient, whose matrix work is 23 us).  Here the B operand -- this workgroup's NBN x 32 columns of the packed weights, <= 135 KB --
// is loaded into LDS once, and every WAVE walks over 32-row tiles on its own: its A fragments come straight from global memory
// (two 16-byte loads per lane and 16-deep step, three steps ahead; every A element is fetched and split by exactly one wave, because
// the wave covers all the workgroup's columns), B fragments from LDS, no barrier after the first one, so one wave's epilogue runs
// under the other waves' matrix work.  Wider outputs (N = 256, or K = 1024 x N = 64) are split along N over `nsplit` workgroups.
constexpr int ST_THREADS = 512;
template <int NBN>
__global__ __launch_bounds__(ST_THREADS, NBN == 1 ? 4 : 2) void stream16_k(GArgs a, int nsplit, int kperm) {
  extern __shared__ __attribute__((aligned(16))) half_t sB[];      // [2 planes][NBN * 32][K + 8]
  __shared__ unsigned s_word;
  constexpr int NC = NBN * 32, WAVES = ST_THREADS / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, l31 = lane & 31;
  const int LDB = a.K + 8;
  const int part = blockIdx.x % nsplit, wg = blockIdx.x / nsplit, nwg = gridDim.x / nsplit;
  const int n_off = part * NC;
  const Geo g = a.g;
  const float sA = tensor_scale(a.amax_in, &s_word);
  {   // this workgroup's columns of both planes -> LDS, eight 16-byte loads in flight per thread (one at a time this prologue was a chain
      // of 8-16 memory round trips)
    const int per_row = a.K / 8;                         // 16-byte units per row
    const int total = 2 * NC * per_row;
    for (int v0 = threadIdx.x; v0 < total; v0 += 8 * ST_THREADS) {
      u32x4 w[8];
      int dst[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int v = v0 + i * ST_THREADS;
        const int vv = min(v, total - 1);
        const int plane = vv / (NC * per_row), rem = vv - plane * (NC * per_row), n = rem / per_row, k8 = rem - n * per_row;
        w[i] = *reinterpret_cast<const u32x4*>(a.B + (int64_t)plane * a.N * a.K + (int64_t)(n_off + n) * a.K + 8 * k8);
        dst[i] = v < total ? (plane * NC + n) * LDB + 8 * k8 : -1;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (dst[i] >= 0) *reinterpret_cast<u32x4*>(sB + dst[i]) = w[i];
    }
  }
  __syncthreads();
  const float cs = 1.f / (sA * *a.bscale);
  const int ntiles = (int)((a.M + 31) / 32);
  const int NB = a.K / 32;                               // 32-deep contraction blocks, a multiple of 4
  const int kq = a.K >> 2;
  const half_t* bbase = sB + l31 * LDB + 16 * h;
  const int planeB = NC * LDB;
  float amax = 0.f;
  // A lane fetches 64 contiguous bytes per block -- floats [16 h, 16 h + 16) of the row's 32 -- so the two lanes of a row take one whole
  // 128-byte line with four back-to-back loads, and the block's two 16-deep matrix steps pair A floats 16 h + 8 u + j with the B
  // elements at the same offsets (the contraction order is free).  (Eight floats per lane and step touched every line twice, four
  // steps apart: with half-lines as the unit the L1 missed 16.8 M times per launch and the kernel waited 68 % of its cycles.)
  // kperm: the K axis is four pixel positions (the 2 x 2 window of the space-to-depth grid) of K / 4 channels each; visiting the four
  // positions of one 32-channel block back to back puts the loads of neighbouring rows -- whose windows overlap in two of the four
  // pixels -- a few instructions apart instead of K / 128 blocks.
  auto k_of = [&](int blk) { return kperm ? (blk & 3) * kq + 32 * (blk >> 2) : 32 * blk; };
  auto row_ptr = [&](int tile) -> const float* {
    const int row = min(tile * 32 + l31, (int)a.M - 1);
    if (g.mode == 2) return a.A + (int64_t)row * a.lda + 16 * h;
    int img, y, x;
    row_to_pixel(g, row, img, y, x);
    return a.A + (((int64_t)img * g.ghs + y) * g.gws + x) * a.lda + 16 * h;
  };
  float4 q[4][4];
  const float* pa = nullptr;
  auto lda = [&](int blk, int slot) {
    const int k = k_of(blk);
    const float* p = pa + k + (k >= a.seg ? a.jump : 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[slot][i] = *reinterpret_cast<const float4*>(p + 4 * i);
  };
  int tile = wg * WAVES + wave;
  if (tile < ntiles) { pa = row_ptr(tile); lda(0, 0); lda(1, 1); lda(2, 2); }
  constexpr int PP = (16 * NBN + 31) / 32;
  for (; tile < ntiles; tile += nwg * WAVES) {
    f32x16 acc[NBN];
#pragma unroll
    for (int j = 0; j < NBN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int next = tile + nwg * WAVES;
    int mypix[PP]; unsigned myword[PP];                    // data gradient: destination offsets and gate words of this tile's rows
    auto gate_request = [&]() {
#pragma unroll
      for (int pi = 0; pi < PP; ++pi) {
        const int p = l31 + 32 * pi, r = p & 15, j = p >> 4;
        const int orow = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int col0 = n_off + 32 * j, qd = col0 / g.c, cw = (col0 - qd * g.c) >> 5;
        mypix[pi] = -1; myword[pi] = 0u;
        if (p < 16 * NBN && orow < a.M) {
          int img, y, x;
          row_to_pixel(g, orow, img, y, x);
          const int yy = 2 * y + (qd >> 1) - 1, xx = 2 * x + (qd & 1) - 1;
          if (yy >= 0 && xx >= 0 && yy < g.dho && xx < g.dwo) {
            const int pix = (img * g.dhs + yy) * g.dws + xx;
            myword[pi] = a.gate_in[(int64_t)pix * (g.c >> 5) + cw];
            mypix[pi] = pix * g.c;                          // element offset of the pixel's channel row (< 2^31: checked at launch)
          }
        }
      }
    };
    for (int b0 = 0; b0 < NB; b0 += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int blk = b0 + u;
        if (blk + 3 < NB) lda(blk + 3, (u + 3) & 3);
        const int kb = k_of(blk);
        auto products = [&](int sub, const u32x4 ahi, const u32x4 alo) {
#if STREAM16_ABLATE & 2
          if (g.mode == 2) { acc[0][0] += __uint_as_float((ahi.x ^ alo.y) & 1u); return; }
#endif
#pragma unroll
          for (int j = 0; j < NBN; ++j) {
            const half_t* pb = bbase + j * 32 * LDB + kb + 8 * sub;
            const u32x4 bhi = *reinterpret_cast<const u32x4*>(pb);
            const u32x4 blo = *reinterpret_cast<const u32x4*>(pb + planeB);
            acc[j] = mfma(alo, bhi, acc[j]);
            acc[j] = mfma(ahi, blo, acc[j]);
            acc[j] = mfma(ahi, bhi, acc[j]);
          }
        };
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          u32x2 h0, l0, h1, l1;
          split4(q[u][2 * sub], sA, h0, l0);
          split4(q[u][2 * sub + 1], sA, h1, l1);
          products(sub, (u32x4){h0.x, h0.y, h1.x, h1.y}, (u32x4){l0.x, l0.y, l1.x, l1.y});
        }
      }
    }
    // the next tile's first loads go out before this tile's epilogue (their latency hides under the stores)
    if (next < ntiles) { pa = row_ptr(next); lda(0, 0); lda(1, 1); lda(2, 2); }

    // ---- epilogue of the wave's 32 x NC block.  C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  The index
    // arithmetic (row -> pixel -> destination, and the data gradient's gate words) is done ONCE per (row, column block) pair by one lane
    // of the half-wave that owns the row and handed round by shuffles: every gate word of the tile is requested before the first store,
    // one memory round trip per tile.  (Per lane and row, with a gate load in front of each group of stores: a chain of sixteen round
    // trips per tile -- 112 us per data-gradient launch.)
    if (g.mode == 2) {
      gate_request();
#pragma unroll
      for (int pi = 0; pi < PP; ++pi) asm volatile("" ::"v"(myword[pi]));      // (one wait, outside the row loop's divergent control flow)
      // The offsets / gate words of a column block's sixteen rows are fetched from their owner lanes EIGHT AT A TIME before the first of
      // their stores (one shuffle pair and its LDS round trip in front of each store before), offsets are 32-bit element indices (one add
      // per row instead of a 64-bit multiply-add).  Measured: the data-gradient launches do not move (131 / 63 / 44 us: with stores AND
      // matrix products compiled out -- STREAM16_ABLATE=3 -- the widest one still takes 0.7 of its time: it waits on its operand loads
      // with two waves per SIMD), the forward launches 60.8 -> 55 us with the bias fix below.
#pragma unroll
      for (int j = 0; j < NBN; ++j) {
        const int col0 = n_off + 32 * j, ch = col0 % g.c + l31;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 8) {
          int po[8]; unsigned wd[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int p = 16 * j + r0 + i, src = (p & 31) + 32 * h;
            po[i] = __shfl(mypix[p >> 5], src, 64);
            wd[i] = (unsigned)__shfl((int)myword[p >> 5], src, 64);
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (po[i] < 0) continue;
            const float v = ((wd[i] >> l31) & 1u) ? acc[j][r0 + i] * cs : 0.f;
#if STREAM16_ABLATE & 1
            if (v != 12345.678f) continue;
#endif
            a.C[(unsigned)(po[i] + ch)] = v;
            amax = fmaxf(amax, fabsf(v));
          }
        }
      }
    } else {
      int myoff = -1, mygrow = 0;                          // lanes 0..15 of each half: row r = lane & 15 of the half's sixteen
      {
        const int r = l31 & 15;
        const int orow = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (orow < a.M) {
          int img, y, x;
          row_to_pixel(g, orow, img, y, x);
          if (g.mode == 3) {
            myoff = ((img * g.ghs + y) * g.gws + x) * a.N;
          } else {
            const int Y = (y + 1) >> 1, X = (x + 1) >> 1, qq = ((y + 1) & 1) * 2 + ((x + 1) & 1);
            myoff = ((img * g.dhs + Y) * g.dws + X) * (4 * g.c) + qq * g.c;
            mygrow = ((img * g.ghs + y) * g.gws + x) * (a.N >> 5);
          }
        }
      }
      const bool gates = g.mode == 1 && a.gate_out;
#pragma unroll
      for (int j = 0; j < NBN; ++j) {
        const int col = n_off + j * 32 + l31;
        const float bv = a.bias ? a.bias[col] : 0.f;
        // consumed HERE, in uniform control flow: first used inside the row loop's `off < 0` branches, the compiler repeated the wait for
        // this load in front of every row -- a vmcnt(0) that also waited for the previous row's STORES: 16 serialised store round trips
        // per column block and tile
        asm volatile("" ::"v"(bv));
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 8) {
          int po[8], gr[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            po[i] = __shfl(myoff, r0 + i + 32 * h, 64);
            gr[i] = gates ? __shfl(mygrow, r0 + i + 32 * h, 64) : 0;
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (po[i] < 0) continue;
            float v = acc[j][r0 + i] * cs + bv;
            if (a.relu) v = v > 0.f ? v : 0.f;
            a.C[(unsigned)(po[i] + col)] = v;
            amax = fmaxf(amax, fabsf(v));
            if (gates) {
              const unsigned long long bits = __ballot(v > 0.f);
              if (l31 == 0) a.gate_out[(unsigned)(gr[i] + (col >> 5))] = (unsigned)(bits >> (32 * h));
            }
          }
        }
      }
    }
  }
  commit_amax(a.amax_out, amax, blockIdx.x * WAVES + wave);
}

template <int NBN>
static int launch_stream16_n(const GArgs& a, int nsplit, hipStream_t st, const char* who) {
  const size_t lds = (size_t)2 * NBN * 32 * (a.K + 8) * sizeof(half_t);
  auto k = stream16_k<NBN>;
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64), true);
  (void)once;
  const int per_cu = (NBN == 1 && lds <= 76 * 1024) ? 2 : 1;     // (the wider variants hold > 128 registers: one workgroup per CU)
  const int64_t tiles = ceil_div(a.M, 32);
  int64_t wgs = std::min<int64_t>((int64_t)per_cu * kNumCU / nsplit, ceil_div(tiles, ST_THREADS / 64));
  wgs = std::max<int64_t>(wgs, 1);
  hipLaunchKernelGGL(k, dim3((unsigned)(wgs * nsplit)), dim3(ST_THREADS), lds, st, a, nsplit, (a.K / 4) % 32 == 0 ? 1 : 0);
  return launch_status(who);
}
// false: the shape does not fit the streaming kernel (the caller takes the tiled one)
static bool stream16_fits(const GArgs& a, int& nbn, int& nsplit) {
  if (a.N % 32 || a.K % 128 || a.seg % 32) return false;
  for (nbn = std::min(4, a.N / 32); nbn >= 1; nbn >>= 1) {
    if ((a.N / 32) % nbn) continue;
    if ((size_t)2 * nbn * 32 * (a.K + 8) * sizeof(half_t) <= 150 * 1024) { nsplit = a.N / (32 * nbn); return nbn != 3; }
  }
  return false;
}
// ---- first stage (one input channel, K = 16) on the matrix cores ---------------------------------------------------------------
// 1 -> 32 channels, k = 4, stride 2: a [pixels x 16] x [16 x 32] product, one 16-deep matrix step.  On the vector ALUs (linear.hip:
// conv_fwd_image_valu_k) the stage is instruction-bound -- 230 vector instructions per 16 pixels, 64 of them FMAs; 90 us with its stores
// switched off, 126 with them -- so it runs here as the same three fp16 products as the other stages: a wave takes 32 consecutive output
// pixels, lane (pixel, h) fetches window rows 2h, 2h + 1 of its pixel straight from the image (the A fragment: k = 8h .. 8h + 7 =
// (ky, kx)), splits them in registers; the packed weights are 8 registers for the whole kernel; the epilogue is the other stages'
// (bias + ReLU + scatter into the next stage's input + gate word + maximum).  ~6 vector instructions per pixel instead of 14.
__global__ __launch_bounds__(256) void fwd_first16_k(const float* __restrict__ X, const half_t* __restrict__ W16, const float* __restrict__ wscale,
                                                      const float* __restrict__ bias, int64_t pixels, int H, int Wd, int relu, float* __restrict__ out,
                                                      unsigned* __restrict__ gate_out, const unsigned* __restrict__ amax_in, unsigned* __restrict__ amax_out) {
  __shared__ unsigned s_word;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, l31 = lane & 31;
  const int ho = H / 2, wo = Wd / 2, dhs = ho / 2 + 1, dws = wo / 2 + 1;
  const float sX = tensor_scale(amax_in, &s_word);
  const float cs = 1.f / (sX * *wscale);
  const u32x4 bhi = *reinterpret_cast<const u32x4*>(W16 + l31 * 16 + 8 * h);
  const u32x4 blo = *reinterpret_cast<const u32x4*>(W16 + 32 * 16 + l31 * 16 + 8 * h);
  const float bv = bias ? bias[l31] : 0.f;
  const float inv_pix = 1.f / (float)(ho * wo), inv_wo = 1.f / (float)wo;
  const int64_t ntiles = (pixels + 31) / 32;
  const int64_t tstride = (int64_t)gridDim.x * 4;
  float4 q0, q1;
  int py = 0, px = 0, myoff = -1, mypix = 0;
  auto fetch = [&](int64_t tile) {
    const int64_t p = tile * 32 + l31;
    const int pc = (int)min(p, pixels - 1);
    int img = (int)((float)pc * inv_pix);
    int rem = pc - img * (ho * wo);
    if (rem < 0) { --img; rem += ho * wo; } else if (rem >= ho * wo) { ++img; rem -= ho * wo; }
    int y = (int)((float)rem * inv_wo), x = rem - y * wo;
    if (x < 0) { --y; x += wo; } else if (x >= wo) { ++y; x -= wo; }
    py = y; px = x; mypix = pc;
    const int Y = (y + 1) >> 1, Xs = (x + 1) >> 1, qq = ((y + 1) & 1) * 2 + ((x + 1) & 1);
    myoff = p < pixels ? ((img * dhs + Y) * dws + Xs) * 128 + qq * 32 : -1;
    const float* ximg = X + (int64_t)img * H * Wd;
    const int wy = 2 * y - 1 + 2 * h;
    // raw loads only (clamped addresses): masked at use
    {
      const int yc = min(max(wy, 0), H - 1);
      const float* rowp = ximg + (int64_t)yc * Wd + 2 * x;
      const float l = rowp[x > 0 ? -1 : 0];
      const float2 m = *reinterpret_cast<const float2*>(rowp);
      const float r = rowp[2 * x + 2 < Wd ? 2 : 0];
      q0 = make_float4(l, m.x, m.y, r);
    }
    {
      const int yc = min(max(wy + 1, 0), H - 1);
      const float* rowp = ximg + (int64_t)yc * Wd + 2 * x;
      const float l = rowp[x > 0 ? -1 : 0];
      const float2 m = *reinterpret_cast<const float2*>(rowp);
      const float r = rowp[2 * x + 2 < Wd ? 2 : 0];
      q1 = make_float4(l, m.x, m.y, r);
    }
  };
  float amax = 0.f;
  int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += tstride) {
    const int wy = 2 * py - 1 + 2 * h;
    const bool lin = px > 0, rgt = 2 * px + 2 < Wd, r0 = wy >= 0 && wy < H, r1 = wy + 1 >= 0 && wy + 1 < H;
    const float4 a0 = make_float4((r0 && lin) ? q0.x : 0.f, r0 ? q0.y : 0.f, r0 ? q0.z : 0.f, (r0 && rgt) ? q0.w : 0.f);
    const float4 a1 = make_float4((r1 && lin) ? q1.x : 0.f, r1 ? q1.y : 0.f, r1 ? q1.z : 0.f, (r1 && rgt) ? q1.w : 0.f);
    u32x2 h0, l0, h1, l1;
    split4(a0, sX, h0, l0);
    split4(a1, sX, h1, l1);
    const u32x4 ahi = {h0.x, h0.y, h1.x, h1.y}, alo = {l0.x, l0.y, l1.x, l1.y};
    const int off_mine = myoff, pix_mine = mypix;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = mfma(alo, bhi, acc);
    acc = mfma(ahi, blo, acc);
    acc = mfma(ahi, bhi, acc);
    if (tile + tstride < ntiles) fetch(tile + tstride);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * h;          // row of the block = pixel i of the tile; its lane (i, any h) holds the offsets
      const int off = __shfl(off_mine, i, 64), pixg = __shfl(pix_mine, i, 64);
      float v = acc[r] * cs + bv;
      if (relu) v = fmaxf(v, 0.f);
      const unsigned long long bits = __ballot(v > 0.f);
      if (off < 0) continue;
      out[off + l31] = v;
      amax = fmaxf(amax, fabsf(v));
      if (gate_out && l31 == 0) gate_out[pixg] = (unsigned)(bits >> (32 * h));
    }
  }
  commit_amax(amax_out, amax, blockIdx.x * 4 + wave);
}

// ---- forward of a 16-pixel-wide stage with 32 -> 32 channels from IMAGE TILES in LDS ---------------------------------------------
// In the streaming kernel every output pixel fetches its own 2 x 2 window of S pixels: 1.07 GB requested through the L1 for the 303 MB
// of the widest stage's input, the waves wait 0.68 of their cycles on the vector-memory path (148 us).  Here a workgroup takes 8 output
// rows x 16 of one image: their 9 x 17 S pixels are ONE contiguous 78 KB block of the space-to-depth tensor, fetched once (registers,
// one tile ahead) and split to f16 pieces on the way into LDS ([plane][pixel][128 channels + 8]).  Wave (pg, kh) computes output rows
// 2 pg, 2 pg + 1 over the window row dy = kh (half of the contraction) with v_mfma_f32_16x16x32_f16: its A fragments -- lane (x, k
// group) = 8 channels of S pixel (row + kh, x + dx) -- come from LDS, its B fragments (the packed weights of its contraction half, 32
// columns x 256) stay in 128 REGISTERS for the whole kernel; the two halves meet through LDS.  (With the weights read from LDS by
// every wave the kernel moved 768 KB through LDS per tile, 6 k of its 12 k cycles: 95 us; this version 288 KB.)
constexpr int TL_TY = 8, TL_PIX = (TL_TY + 1) * 17, TL_LDA = 128 + 8, TL_THREADS = 512;
constexpr int TL_F4 = TL_PIX * 32;                                   // float4 units of a tile (4896)
constexpr int TL_UNITS = (TL_F4 + TL_THREADS - 1) / TL_THREADS;      // 10 per thread
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v mfma16(const u32x4 a, const u32x4 b, const f32x4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__global__ __launch_bounds__(TL_THREADS) void fwd_tile16_k(GArgs a, int ntiles, int tiles_per_image) {
  extern __shared__ __attribute__((aligned(16))) half_t tsm[];       // A: [2][TL_PIX][TL_LDA]; behind it: partial sums [4 pg][16][64] floats
  __shared__ unsigned s_word;
  half_t* const sA = tsm;
  float* const red = reinterpret_cast<float*>(tsm + 2 * TL_PIX * TL_LDA);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = lane & 15, kg = lane >> 4;
  const int pg = wave >> 1, kh = wave & 1;
  const Geo g = a.g;
  const float sA_scale = tensor_scale(a.amax_in, &s_word);
  // this wave's half of the packed weights: [step 0..7][column block 0..1][hi / lo], lane (column x, k group kg)
  u32x4 bw[8][2][2];
#pragma unroll
  for (int st = 0; st < 8; ++st)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        bw[st][blk][pl] = *reinterpret_cast<const u32x4*>(a.B + (int64_t)pl * 32 * 512 + (16 * blk + x) * 512 + kh * 256 + 32 * st + 8 * kg);
  const float cs = 1.f / (sA_scale * *a.bscale);
  const float bv0 = a.bias ? a.bias[x] : 0.f, bv1 = a.bias ? a.bias[16 + x] : 0.f;
  float4 rq[TL_UNITS];
  auto load = [&](int tile) {
    const int img = tile / tiles_per_image, ty = tile - img * tiles_per_image;
    const float* src = a.A + (((int64_t)img * g.ghs + ty * TL_TY) * g.gws) * 128;
#pragma unroll
    for (int i = 0; i < TL_UNITS; ++i) {
      const int f = threadIdx.x + i * TL_THREADS;
      rq[i] = *reinterpret_cast<const float4*>(src + 4 * min(f, TL_F4 - 1));
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < TL_UNITS; ++i) {
      const int f = threadIdx.x + i * TL_THREADS;
      if (f >= TL_F4) continue;
      const int pix = f >> 5, ch = (f & 31) * 4;
      u32x2 hi, lo;
      split4(rq[i], sA_scale, hi, lo);
      *reinterpret_cast<u32x2*>(sA + pix * TL_LDA + ch) = hi;
      *reinterpret_cast<u32x2*>(sA + TL_PIX * TL_LDA + pix * TL_LDA + ch) = lo;
    }
  };
  float amax = 0.f;
  int tile = blockIdx.x;
  if (tile < ntiles) load(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    __syncthreads();                                     // the previous tile's fragments and partial sums have been read
    store();
    if (tile + (int)gridDim.x < ntiles) load(tile + gridDim.x);
    __syncthreads();
    f32x4v acc[2][2];                                    // [output row of the pair][column block]
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) acc[rr][blk] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    const half_t* arow = sA + ((2 * pg + kh) * 17 + x) * TL_LDA + 8 * kg;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const int dx = st >> 2, cb = st & 3;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const half_t* pa = arow + (rr * 17 + dx) * TL_LDA + 32 * cb;
        const u32x4 ahi = *reinterpret_cast<const u32x4*>(pa), alo = *reinterpret_cast<const u32x4*>(pa + TL_PIX * TL_LDA);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          acc[rr][blk] = mfma16(alo, bw[st][blk][0], acc[rr][blk]);
          acc[rr][blk] = mfma16(ahi, bw[st][blk][1], acc[rr][blk]);
          acc[rr][blk] = mfma16(ahi, bw[st][blk][0], acc[rr][blk]);
        }
      }
    }
    // the two contraction halves meet: kh = 1 leaves its sums in LDS, kh = 0 adds them and runs the epilogue
    if (kh == 1) {
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(pg * 16 + (rr * 2 + blk) * 4 + r) * 64 + lane] = acc[rr][blk][r];
    }
    __syncthreads();
    if (kh == 0) {
      // C/D layout of a 16 x 16 block: column (channel) = lane & 15, row (output x) = 4 (lane >> 4) + r
      const int img = tile / tiles_per_image, ty = tile - img * tiles_per_image;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int y = ty * TL_TY + 2 * pg + rr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ox = 4 * kg + r;
          const int Y = (y + 1) >> 1, X = (ox + 1) >> 1, qq = ((y + 1) & 1) * 2 + ((ox + 1) & 1);
          float* dst = a.C + (((int64_t)img * g.dhs + Y) * g.dws + X) * (4 * 32) + qq * 32;
          float v0 = (acc[rr][0][r] + red[(pg * 16 + (rr * 2 + 0) * 4 + r) * 64 + lane]) * cs + bv0;
          float v1 = (acc[rr][1][r] + red[(pg * 16 + (rr * 2 + 1) * 4 + r) * 64 + lane]) * cs + bv1;
          if (a.relu) { v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f; }
          dst[x] = v0;
          dst[16 + x] = v1;
          amax = fmaxf(amax, fmaxf(fabsf(v0), fabsf(v1)));
          if (a.gate_out) {
            const unsigned long long b0 = __ballot(v0 > 0.f), b1 = __ballot(v1 > 0.f);
            const unsigned word = (unsigned)((b0 >> (16 * kg)) & 0xffffull) | ((unsigned)((b1 >> (16 * kg)) & 0xffffull) << 16);
            if (x == 0) a.gate_out[((int64_t)img * g.ghs + y) * g.gws + ox] = word;
          }
        }
      }
    }
  }
  commit_amax(a.amax_out, amax, blockIdx.x * (TL_THREADS / 64) + wave);
}
static bool tile16_fits(const GArgs& a) {
  const Geo& g = a.g;
  return g.mode == 1 && a.N == 32 && a.K == 512 && a.lda == 128 && g.ws == 16 && g.gws == 17 && g.hs % TL_TY == 0 && g.ghs == g.hs + 1 && g.c == 32;
}
static int launch_tile16(const GArgs& a, hipStream_t st, const char* who) {
  constexpr size_t lds = (size_t)(2 * TL_PIX * TL_LDA) * sizeof(half_t) + 4 * 16 * 64 * sizeof(float);
  auto k = fwd_tile16_k;
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
  (void)once;
  const int tiles_per_image = a.g.hs / TL_TY;
  const int ntiles = (int)(a.M / (a.g.hs * a.g.ws)) * tiles_per_image;
  hipLaunchKernelGGL(k, dim3((unsigned)std::min(ntiles, kNumCU)), dim3(TL_THREADS), lds, st, a, ntiles, tiles_per_image);
  return launch_status(who);
}

static int launch_conv16(const GArgs& a, hipStream_t st, const char* who) {
  if (tile16_fits(a)) return launch_tile16(a, st, who);
  int nbn = 0, nsplit = 0;
  if (stream16_fits(a, nbn, nsplit)) {
    if (nbn == 1) return launch_stream16_n<1>(a, nsplit, st, who);
    if (nbn == 2) return launch_stream16_n<2>(a, nsplit, st, who);
    return launch_stream16_n<4>(a, nsplit, st, who);
  }
  return launch_gemm16_auto(a, st, who);
}

// ---- weight gradient: dWg[Cout][K] = sum_r dO[r][Cout] A[r][K], contraction over the rows of the stage's grid --------------------
// The implicit operand is the same S pixel seen through four shifts: A[r][(dy, dx, j)] = S[r + dy ws + dx][j] (j < 4C), so with p = the
// S row,   dWg[co][(dy, dx), j] = sum_p dO[p - dy ws - dx][co] S[p][j]:   one pass over S serves all four (dy, dx) blocks of dWg, each
// against dO shifted by 0 / 1 / ws / ws + 1 rows.  (Taking the four blocks as four column tiles of one GEMM, as the fp32 path does, reads
// S four times -- 1.2 GB through L2 on the widest stage, 156 us.)  A workgroup (eight waves) owns all Cout rows x 128 columns j of the
// four blocks for one contraction split: per step of 32 S rows it fetches the 32 x 128 S tile once and the four shifted 32 x Cout windows
// of dO (overlapping, L1 / L2 hits), splits them to f16 pieces on the way into LDS, and wave (pair q, column block cb) accumulates
// shifts 2q, 2q + 1 against S columns 32 cb .. + 31.
// Both operands have the contraction along their ROWS, so an MFMA operand (eight consecutive contraction indices per lane) is a
// transposed read: the tiles sit in LDS as f16 pieces of 16 rows x 32 features in the order planes.h describes
//     byte = (k / 4) * 256 + (f / 16) * 128 + (k % 4) * 32 + (f % 16) * 2            (k = row % 16, f = feature % 32)
// and a fragment is two ds_read_b64_tr_b16.  Registers hold the next step's fp32 values, two LDS stages, one barrier per step.
// db = column sums of the unshifted dO window, kept in registers by the column tile 0 workgroups.
constexpr int WG_ROWS = 32;                     // contraction rows per step
constexpr int WG_COLS = 128;                    // S columns per workgroup
constexpr int WG_THREADS = 512;
constexpr int PIECE = 1024;                     // bytes of a 16 x 32 f16 piece
struct WArgs {
  const float* dO; const float* S;
  int Cout, C4;                                 // C4 = 4 C = floats per S row
  int ws;                                       // shifts 0, 1, ws, ws + 1
  int64_t rows;                                 // rows of the stage's grid (dO rows); the contraction runs over rows + ws + 1 S rows
  int64_t prows, rows_per_split;                // prows = rows + ws + 1; rows_per_split a multiple of WG_ROWS
  int K;                                        // 4 * C4
  const unsigned* amax_dO; const unsigned* amax_S;
  float* slab; float* dbslab;                   // [splits][Cout][K], [splits][Cout]
};
__device__ __forceinline__ u32x4 read_frag_tr(const char* p) {   // keys 8h .. 8h+7 of this lane's feature
  const u32x2 a = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds_ptr)p));
  const u32x2 b = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds_ptr)(p + 256)));
  return (u32x4){a.x, a.y, b.x, b.y};
}

template <int NBM>      // Cout = 32 NBM
__global__ __launch_bounds__(WG_THREADS, NBM == 1 ? 4 : 2) void wgrad16_k(WArgs a) {
  constexpr int COUT = 32 * NBM;
  constexpr int B_UNITS = WG_ROWS * WG_COLS / 4 / WG_THREADS;         // 2 float4 of S per thread and step
  constexpr int A_UNITS = 4 * WG_ROWS * COUT / 4 / WG_THREADS;        // 2 or 4 float4 of the four dO windows
  constexpr int UPW = WG_ROWS * COUT / 4;                             // float4 units per dO window
  // LDS stage: [row group 0..1][unit: 4 shifts x NBM of dO, then 4 of S][hi / lo][1 KB]; unit stride padded by 64 B (the units of a row
  // would otherwise land on the same banks when the split values are stored)
  constexpr int UNIT = 2 * PIECE + 64;
  constexpr int GROUP = (4 * NBM + 4) * UNIT;
  constexpr int STAGE = 2 * GROUP;
  extern __shared__ __attribute__((aligned(16))) char wsmem[];        // 2 * STAGE
  char* const smem = wsmem;
  __shared__ unsigned s_word;
  __shared__ float s_db[WG_THREADS * 4];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, l31 = lane & 31;
  const int q = wave >> 2, cb = wave & 3;
  const int ctiles = a.C4 / WG_COLS;
  const int id = xcd_contiguous(blockIdx.x, gridDim.x);
  const int split = id / ctiles, ct = id - split * ctiles;
  const int64_t p_beg = (int64_t)split * a.rows_per_split, p_end = min(a.prows, p_beg + a.rows_per_split);
  const int nsteps = (int)((p_end - p_beg + WG_ROWS - 1) / WG_ROWS);

  const float sD = tensor_scale(a.amax_dO, &s_word);
  const float sS = tensor_scale(a.amax_S, &s_word);

  const float* __restrict__ Sbase = a.S + ct * WG_COLS;
  float4 rs[B_UNITS], rd[A_UNITS];
  auto load = [&](int step) {
    const int64_t p0 = p_beg + (int64_t)step * WG_ROWS;
#pragma unroll
    for (int i = 0; i < B_UNITS; ++i) {
      const int u = threadIdx.x + i * WG_THREADS;
      const int64_t pr = p0 + u / 32;
      // (rows past the end of the split meet zero dO windows; S is readable up to row prows by the caller's zero tail)
      rs[i] = *reinterpret_cast<const float4*>(Sbase + min(pr, a.prows) * a.C4 + 4 * (u % 32));
    }
#pragma unroll
    for (int i = 0; i < A_UNITS; ++i) {
      const int u = threadIdx.x + i * WG_THREADS;
      const int w = u / UPW, rem = u - w * UPW;
      const int shift = (w >> 1) * a.ws + (w & 1);
      const int64_t pr = p0 + rem / (COUT / 4), r = pr - shift;          // r >= -(ws + 1): the caller's zero rows in front of dO
      rd[i] = (pr < p_end && r < a.rows) ? *reinterpret_cast<const float4*>(a.dO + r * COUT + 4 * (rem % (COUT / 4))) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  float4 dbsum = make_float4(0.f, 0.f, 0.f, 0.f);
  auto piece_off = [](int k, int f) { return (k >> 2) * 256 + ((f >> 4) & 1) * 128 + (k & 3) * 32 + (f & 15) * 2; };
  auto store = [&](char* st) {
#pragma unroll
    for (int i = 0; i < B_UNITS; ++i) {
      const int u = threadIdx.x + i * WG_THREADS;
      const int row = u / 32, f = 4 * (u % 32);
      u32x2 hi, lo;
      split4(rs[i], sS, hi, lo);
      char* p = st + (row >> 4) * GROUP + (4 * NBM + (f >> 5)) * UNIT + piece_off(row & 15, f & 31);
      *reinterpret_cast<u32x2*>(p) = hi;
      *reinterpret_cast<u32x2*>(p + PIECE) = lo;
    }
#pragma unroll
    for (int i = 0; i < A_UNITS; ++i) {
      const int u = threadIdx.x + i * WG_THREADS;
      const int w = u / UPW, rem = u - w * UPW;
      const int row = rem / (COUT / 4), f = 4 * (rem % (COUT / 4));
      u32x2 hi, lo;
      split4(rd[i], sD, hi, lo);
      char* p = st + (row >> 4) * GROUP + (w * NBM + (f >> 5)) * UNIT + piece_off(row & 15, f & 31);
      *reinterpret_cast<u32x2*>(p) = hi;
      *reinterpret_cast<u32x2*>(p + PIECE) = lo;
      if (w == 0) { dbsum.x += rd[i].x; dbsum.y += rd[i].y; dbsum.z += rd[i].z; dbsum.w += rd[i].w; }
    }
  };

  f32x16 acc[2][NBM];
#pragma unroll
  for (int sh = 0; sh < 2; ++sh)
#pragma unroll
    for (int i = 0; i < NBM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[sh][i][r] = 0.f;

  const int lane_off = h * 512 + ((lane >> 4) & 1) * 128 + (lane & 15) * 8;
  if (nsteps > 0) {
    load(0);
    store(smem);
    if (nsteps > 1) load(1);
  }
  __syncthreads();
  int cur = 0;
  for (int t = 0; t < nsteps; ++t) {
    const char* st = smem + cur * STAGE;
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {                  // the step's two 16-row groups
      u32x4 fa[2][NBM][2], fb[2];                     // [shift of the pair][block][hi / lo], [hi / lo]
#pragma unroll
      for (int sh = 0; sh < 2; ++sh)
#pragma unroll
        for (int i = 0; i < NBM; ++i) {
          const char* pu = st + gq * GROUP + ((2 * q + sh) * NBM + i) * UNIT + lane_off;
          fa[sh][i][0] = read_frag_tr(pu);
          fa[sh][i][1] = read_frag_tr(pu + PIECE);
        }
      const char* pb = st + gq * GROUP + (4 * NBM + cb) * UNIT + lane_off;
      fb[0] = read_frag_tr(pb);
      fb[1] = read_frag_tr(pb + PIECE);
      if (gq == 0) {
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < nsteps) store(smem + (cur ^ 1) * STAGE);
        if (t + 2 < nsteps) load(t + 2);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int sh = 0; sh < 2; ++sh)
#pragma unroll
        for (int i = 0; i < NBM; ++i) {
          acc[sh][i] = mfma(fa[sh][i][1], fb[0], acc[sh][i]);
          acc[sh][i] = mfma(fa[sh][i][0], fb[1], acc[sh][i]);
          acc[sh][i] = mfma(fa[sh][i][0], fb[0], acc[sh][i]);
        }
    }
    __syncthreads();
    cur ^= 1;
  }

  // slab: rows = output channel, columns = block (dy, dx) = shift index 2q + sh, then this wave's 32 of the tile's 128 S columns
  const float cs = 1.f / (sD * sS);
  float* __restrict__ slab = a.slab + (int64_t)split * COUT * a.K;
#pragma unroll
  for (int sh = 0; sh < 2; ++sh)
#pragma unroll
    for (int i = 0; i < NBM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        slab[(int64_t)co * a.K + (2 * q + sh) * a.C4 + ct * WG_COLS + cb * 32 + l31] = acc[sh][i][r] * cs;
      }
  if (a.dbslab && ct == 0) {
    // threads whose units lie in the unshifted window hold the column sums of their channel group (the others hold zero)
    reinterpret_cast<float4*>(s_db)[threadIdx.x] = dbsum;
    __syncthreads();
    if (threadIdx.x < COUT) {
      const int qg = threadIdx.x >> 2, e = threadIdx.x & 3;
      float v = 0.f;
      for (int u = qg; u < WG_THREADS; u += COUT / 4) v += s_db[u * 4 + e];
      a.dbslab[(int64_t)split * COUT + threadIdx.x] = v;
    }
  }
}

// out[e] (+)= sum over splits of slab[s][e] for the weight slabs (n1 elements) and the bias slabs (n2) in ONE launch: a workgroup owns 16
// float4 units, its 16 thread groups take the splits s = g (mod 16) and are combined through LDS in a fixed order -- deterministic.
// (64 units per workgroup and 4 groups: 64 workgroups for the widest stage, every wave a chain of 64 dependent-latency loads, 14 us.)
__global__ __launch_bounds__(256) void slab_sum_k(const float* __restrict__ slab1, int n1, float* __restrict__ out1, const float* __restrict__ slab2, int n2,
                                                  float* __restrict__ out2, int splits, int accumulate, int blocks1) {
  __shared__ float4 red[16][16];
  const bool second = (int)blockIdx.x >= blocks1;
  const float* __restrict__ slab = second ? slab2 : slab1;
  const int n = second ? n2 : n1;
  float* __restrict__ out = second ? out2 : out1;
  const int ul = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int u = ((int)blockIdx.x - (second ? blocks1 : 0)) * 16 + ul;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (4 * u < n) {
    for (int q0 = grp; q0 < splits; q0 += 64) {
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int q = q0 + 16 * i; v[i] = q < splits ? *reinterpret_cast<const float4*>(slab + (int64_t)q * n + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
      for (int i = 0; i < 4; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
    }
  }
  red[grp][ul] = s;
  __syncthreads();
  if (grp == 0 && 4 * u < n) {
    float4 t = red[0][ul];
#pragma unroll
    for (int q = 1; q < 16; ++q) { t.x += red[q][ul].x; t.y += red[q][ul].y; t.z += red[q][ul].z; t.w += red[q][ul].w; }
    float4* o = reinterpret_cast<float4*>(out + 4 * u);
    if (accumulate) { const float4 p = *o; t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w; }
    *o = t;
  }
}

// (also used by linear.hip's first-stage weight gradient)
void launch_slab_sum(const float* slab1, int n1, float* out1, const float* slab2, int n2, float* out2, int splits, int accumulate, hipStream_t st) {
  const int blocks1 = (int)ceil_div(n1 / 4, 16), blocks2 = out2 ? (int)ceil_div(n2 / 4, 16) : 0;
  hipLaunchKernelGGL(slab_sum_k, dim3((unsigned)(blocks1 + blocks2)), dim3(256), 0, st, slab1, n1, out1, slab2, n2, out2, splits, accumulate, blocks1);
}

struct WPlan { int splits; int64_t rows_per_split; };
static WPlan plan_wgrad(int64_t prows, int C4) {
  const int ctiles = C4 / WG_COLS;
  int64_t want = std::max<int64_t>(1, 2 * (int64_t)kNumCU / ctiles);       // ~two eight-wave workgroups per CU
  want = std::min<int64_t>(want, std::max<int64_t>(1, prows / (4 * WG_ROWS)));
  WPlan p;
  p.rows_per_split = ceil_div(ceil_div(prows, want), (int64_t)WG_ROWS) * WG_ROWS;
  p.splits = (int)ceil_div(prows, p.rows_per_split);
  return p;
}
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace conv16
}  // namespace clica

using namespace clica;
using namespace clica::conv16;

extern "C" int clica_conv16_pack(int32_t n, const float* const* src, const int32_t* src_count, const int32_t* const* map, uint16_t* const* dst,
                                 const int32_t* count, float* scales, clica_stream_t stream) {
  CLICA_CHECK_ARG(n >= 1 && n <= MAXPACK && src && src_count && map && dst && count && scales, "clica_conv16_pack: 1..%d matrices", MAXPACK);
  PackArgs P{};
  P.n = n; P.scale = scales;
  for (int i = 0; i < n; ++i) {
    CLICA_CHECK_ARG(src[i] && dst[i] && count[i] > 0 && src_count[i] > 0 && aligned16(src[i]), "clica_conv16_pack: matrix %d: bad argument (16-byte aligned sources)", i);
    P.src[i] = src[i]; P.map[i] = map[i]; P.dst[i] = dst[i]; P.count[i] = count[i]; P.src_count[i] = src_count[i];
  }
  hipLaunchKernelGGL(pack_k, dim3((unsigned)n * PACK_CHUNKS), dim3(1024), 0, as_stream(stream), P);
  return launch_status("clica_conv16_pack");
}

extern "C" int clica_conv16_amax(const float* x, int64_t n, uint32_t* slots, clica_stream_t stream) {
  CLICA_CHECK_ARG(x && slots && n > 0 && n % 4 == 0 && aligned16(x), "clica_conv16_amax: bad argument (n a multiple of 4, 16-byte aligned)");
  const int64_t blocks = std::min<int64_t>(ceil_div(n / 4, 256), 4 * kNumCU);
  hipLaunchKernelGGL(amax_k, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, n / 4, slots);
  return launch_status("clica_conv16_amax");
}

extern "C" int clica_conv16_zero_slots(uint32_t* slots, int32_t tensors, clica_stream_t stream) {
  CLICA_CHECK_ARG(slots && tensors > 0, "clica_conv16_zero_slots: bad argument");
  hipLaunchKernelGGL(zero_slots_k, dim3((unsigned)ceil_div((int64_t)tensors * kSlots, 256)), dim3(256), 0, as_stream(stream), slots, tensors * kSlots);
  return launch_status("clica_conv16_zero_slots");
}

extern "C" int clica_conv16_k4s2_fwd(const float* S, const uint16_t* Wg16, const float* wscale, const float* bias, int64_t images, int32_t C,
                                     int32_t Cout, int32_t hs, int32_t ws, int32_t relu, int32_t scatter, float* out, uint32_t* gate_bits,
                                     const uint32_t* amax_in, uint32_t* amax_out, clica_stream_t stream) {
  CLICA_CHECK_ARG(S && Wg16 && wscale && out && images > 0 && C >= 8 && C % 8 == 0 && Cout >= 32 && Cout % 32 == 0 && hs >= 2 && ws >= 2,
                  "clica_conv16_k4s2_fwd: bad argument (C a multiple of 8, Cout a multiple of 32)");
  CLICA_CHECK_ARG(scatter == 1 || scatter == 2, "clica_conv16_k4s2_fwd: scatter must be 1 (next stage's input) or 2 (rows on the grid)");
  CLICA_CHECK_ARG(scatter != 1 || ((hs - 1) % 2 == 0 && (ws - 1) % 2 == 0), "clica_conv16_k4s2_fwd: scatter = 1 needs an even output grid");
  CLICA_CHECK_ARG(aligned16(S) && aligned16(Wg16), "clica_conv16_k4s2_fwd: operands must be 16-byte aligned");
  CLICA_CHECK_ARG(!gate_bits || scatter == 1, "clica_conv16_k4s2_fwd: gate bits need scatter = 1");
  const int ho = hs - 1, wo = ws - 1;
  GArgs a{};
  a.A = S; a.lda = 4 * (int64_t)C; a.seg = 8 * C; a.jump = (ws - 2) * 4 * C;
  a.B = Wg16; a.bscale = wscale; a.amax_in = amax_in; a.amax_out = amax_out;
  a.C = out; a.bias = bias; a.relu = relu ? 1 : 0;
  a.M = images * ho * wo; a.N = Cout; a.K = 16 * C;
  a.gate_out = gate_bits;
  CLICA_CHECK_ARG(a.M < (1 << 24), "clica_conv16_k4s2_fwd: %lld rows (< 2^24 supported)", (long long)a.M);
  CLICA_CHECK_ARG(a.M * Cout < ((int64_t)1 << 30), "clica_conv16_k4s2_fwd: %lld output elements (< 2^30 supported: 32-bit offsets into the padded destination)",
                  (long long)(a.M * Cout));
  Geo& g = a.g;
  g.mode = scatter == 1 ? 1 : 3;
  g.hs = ho; g.ws = wo; g.ghs = hs; g.gws = ws; g.dhs = ho / 2 + 1; g.dws = wo / 2 + 1; g.dho = g.dwo = 0; g.c = Cout;
  g.inv_pix = 1.f / (float)(ho * wo); g.inv_ws = 1.f / (float)wo;
  return launch_conv16(a, as_stream(stream), "clica_conv16_k4s2_fwd");
}

extern "C" int clica_conv16_k4s2_dgrad(const float* dO, const uint16_t* WdT16, const float* wscale, int64_t images, int32_t C, int32_t Cout,
                                       int32_t hs, int32_t ws, float* dPrev, int32_t dhs, int32_t dws, const uint32_t* gate_bits,
                                       const uint32_t* amax_in, uint32_t* amax_out, clica_stream_t stream) {
  CLICA_CHECK_ARG(dO && WdT16 && wscale && dPrev && gate_bits && images > 0 && C >= 32 && C % 32 == 0 && Cout >= 8 && Cout % 8 == 0 && hs >= 2 && ws >= 2,
                  "clica_conv16_k4s2_dgrad: bad argument (C a multiple of 32, Cout a multiple of 8, gate bits required)");
  CLICA_CHECK_ARG(dhs >= 2 * (hs - 1) && dws >= 2 * (ws - 1), "clica_conv16_k4s2_dgrad: destination grid smaller than 2 (hs - 1) x 2 (ws - 1)");
  CLICA_CHECK_ARG(aligned16(dO) && aligned16(WdT16), "clica_conv16_k4s2_dgrad: operands must be 16-byte aligned");
  GArgs a{};
  a.A = dO - (int64_t)(ws + 1) * Cout; a.lda = Cout; a.seg = 2 * Cout; a.jump = (ws - 2) * Cout;
  a.B = WdT16; a.bscale = wscale; a.amax_in = amax_in; a.amax_out = amax_out;
  a.C = dPrev; a.bias = nullptr; a.relu = 0;
  a.M = images * hs * ws; a.N = 4 * C; a.K = 4 * Cout;
  a.gate_in = gate_bits;
  CLICA_CHECK_ARG(a.M < (1 << 24), "clica_conv16_k4s2_dgrad: %lld rows (< 2^24 supported)", (long long)a.M);
  CLICA_CHECK_ARG(images * dhs * dws * C < ((int64_t)1 << 31), "clica_conv16_k4s2_dgrad: %lld destination elements (< 2^31 supported: 32-bit offsets)",
                  (long long)(images * dhs * dws * C));
  Geo& g = a.g;
  g.mode = 2; g.hs = hs; g.ws = ws; g.ghs = hs; g.gws = ws; g.dhs = dhs; g.dws = dws; g.dho = 2 * (hs - 1); g.dwo = 2 * (ws - 1); g.c = C;
  g.inv_pix = 1.f / (float)(hs * ws); g.inv_ws = 1.f / (float)ws;
  return launch_conv16(a, as_stream(stream), "clica_conv16_k4s2_dgrad");
}

extern "C" int clica_conv16_k4s2_wgrad_workspace_bytes(int64_t rows, int32_t Cout, int32_t K, size_t* bytes) {
  CLICA_CHECK_ARG(bytes && rows > 0 && Cout >= 1 && K >= 4 * WG_COLS && K % (4 * WG_COLS) == 0, "clica_conv16_k4s2_wgrad_workspace_bytes: bad argument");
  const WPlan p = plan_wgrad(rows + 4 * 1024, K / 4);      // an upper bound over the grid widths (prows = rows + ws + 1)
  *bytes = align_up((size_t)(p.splits + 2) * Cout * K * sizeof(float), 256) + align_up((size_t)(p.splits + 2) * Cout * sizeof(float), 256);
  return CLICA_OK;
}

extern "C" int clica_conv16_k4s2_wgrad(const float* dO, const float* S, int64_t images, int32_t C, int32_t Cout, int32_t hs, int32_t ws,
                                       float* dWg, float* db, int32_t accumulate, const uint32_t* amax_dO, const uint32_t* amax_S,
                                       void* workspace, size_t workspace_bytes, clica_stream_t stream) {
  CLICA_CHECK_ARG(dO && S && dWg && workspace && images > 0 && C >= 32 && C % 32 == 0 && (Cout == 32 || Cout == 64) && hs >= 2 && ws >= 2 && ws < 4096,
                  "clica_conv16_k4s2_wgrad: bad argument (C a multiple of 32, Cout 32 or 64)");
  CLICA_CHECK_ARG(aligned16(dO) && aligned16(S) && aligned16(dWg) && (!db || aligned16(db)), "clica_conv16_k4s2_wgrad: operands must be 16-byte aligned");
  const int64_t rows = images * hs * ws;
  const int32_t K = 16 * C;
  const int64_t prows = rows + ws + 1;
  const WPlan p = plan_wgrad(prows, 4 * C);
  const size_t slab_bytes = align_up((size_t)p.splits * Cout * K * sizeof(float), 256);
  const size_t need = slab_bytes + align_up((size_t)p.splits * Cout * sizeof(float), 256);
  if (need > workspace_bytes) { set_error("clica_conv16_k4s2_wgrad: workspace %zu < %zu", workspace_bytes, need); return CLICA_E_WORKSPACE; }
  WArgs a{};
  a.dO = dO; a.S = S; a.Cout = Cout; a.C4 = 4 * C; a.ws = ws;
  a.rows = rows; a.prows = prows; a.rows_per_split = p.rows_per_split; a.K = K;
  a.amax_dO = amax_dO; a.amax_S = amax_S;
  a.slab = (float*)workspace; a.dbslab = db ? (float*)((char*)workspace + slab_bytes) : nullptr;
  hipStream_t st = as_stream(stream);
  const unsigned nwg = (unsigned)(p.splits * (4 * C / WG_COLS));
  const int nbm = Cout / 32;
  const size_t lds = (size_t)2 * 2 * (4 * nbm + 4) * (2 * PIECE + 64);
  if (Cout == 32) {
    auto k = wgrad16_k<1>;
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    hipLaunchKernelGGL(k, dim3(nwg), dim3(WG_THREADS), lds, st, a);
  } else {
    auto k = wgrad16_k<2>;
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    hipLaunchKernelGGL(k, dim3(nwg), dim3(WG_THREADS), lds, st, a);
  }
  int rc = launch_status("clica_conv16_k4s2_wgrad");
  if (rc) return rc;
  launch_slab_sum(a.slab, Cout * K, dWg, a.dbslab, (int)Cout, db, p.splits, accumulate ? 1 : 0, st);
  return launch_status("clica_conv16_k4s2_wgrad(reduce)");
}

extern "C" int clica_conv16_first_fwd(const float* x, const uint16_t* W16, const float* wscale, const float* bias, int64_t images, int32_t H, int32_t W,
                                      int32_t Cout, int32_t relu, float* out, uint32_t* gate_bits, const uint32_t* amax_in, uint32_t* amax_out,
                                      clica_stream_t stream) {
  CLICA_CHECK_ARG(x && W16 && wscale && out && images > 0 && H >= 4 && W >= 4 && H % 4 == 0 && W % 4 == 0 && Cout == 32,
                  "clica_conv16_first_fwd: bad argument (one input channel, Cout = 32, H and W multiples of 4)");
  CLICA_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 7) == 0 && aligned16(W16), "clica_conv16_first_fwd: x must be 8-byte, the packed weights 16-byte aligned");
  const int64_t pixels = images * (H / 2) * (W / 2);
  CLICA_CHECK_ARG(pixels < (1 << 24), "clica_conv16_first_fwd: %lld output pixels (< 2^24 supported)", (long long)pixels);
  const int64_t tiles = ceil_div(pixels, 32);
  hipLaunchKernelGGL(fwd_first16_k, dim3((unsigned)std::min<int64_t>(ceil_div(tiles, 4), 8 * kNumCU)), dim3(256), 0, as_stream(stream), x, W16, wscale, bias,
                     pixels, (int)H, (int)W, (int)relu, out, gate_bits, amax_in, amax_out);
  return launch_status("clica_conv16_first_fwd");
}
