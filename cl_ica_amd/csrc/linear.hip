// Fused Linear (+bias, +LeakyReLU) forward / dgrad / wgrad for the MLP encoder
// (get_mlp, /root/reference/encoders.py:36-48) on gfx950, exact fp32 on the matrix cores.
//
// One LDS-tiled GEMM template, C[M,N] = A_op[M,Kc] * B_op[Kc,N], instantiated for the three
// operand layouts the layer needs (all tensors row-major, contraction index Kc):
//   fwd   Y  = act(X W^T + b)        A = X  [M][Kc]  (Kc contiguous)   B = W  [N][Kc]  (Kc contiguous)
//   dgrad dX = (dY W) * act'(Xact)   A = dY [M][Kc]  (Kc contiguous)   B = W  [Kc][N]  (Kc strided)
//   wgrad dW = dY^T X, db = 1^T dY   A = dY [Kc][M]  (Kc strided)      B = X  [Kc][N]  (Kc strided)
//
// Math: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate; bit-identical to an fmaf chain, which is
// what keeps the 1e-5 parity bar -- bf16/xf32 inputs would not).  Each wave owns a (BM/WM)x(BN/WN)
// sub-tile as 32x32 accumulator blocks.  Because the contraction order is free, a lane reads FOUR
// consecutive k (one ds_read_b128) and the four MFMAs that follow pair k = 8s+4h+t of both
// operands (h = lane>>5), so a K-contiguous operand costs one LDS read per four MFMAs.
//
// Pipeline: global -> registers (next tile) overlaps the MFMAs of the current tile; registers ->
// LDS double buffer; one barrier per K tile.  wgrad splits the long contraction (Kc = rows of the
// batch) over blockIdx.z into fp32 slabs that a second tiny kernel sums deterministically.
#include "common.h"

namespace clica {
namespace gemm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int THREADS = 256;
constexpr int BK = 32;

enum Epi { EPI_BIAS_ACT = 0, EPI_DACT = 1, EPI_SLAB = 2 };

struct Args {
  const float* A; int64_t lda;
  const float* B; int64_t ldb;
  float* C; int64_t ldc;
  int64_t M, N, Kc;
  const float* bias;      // fwd: [N]
  const float* xact; int64_t ldxa;  // dgrad: saved activation [M][N]
  float slope; int leaky;
  int64_t k_per_split;    // wgrad: contraction rows per blockIdx.z
  float* dbias_slab;      // wgrad: [splits][M] column sums of A_op rows (db), or nullptr
};

// ---- tile loaders: global -> registers ------------------------------------------------------
// CONTIG: operand stored [rows][Kc]; tile = ROWS x BK, float4 along k.
// !CONTIG: operand stored [Kc][rows]; tile = BK x ROWS, float4 along rows.
template <int ROWS, bool CONTIG>
struct Tile {
  static constexpr int UNITS = ROWS * BK / 4;          // float4 units per tile
  static constexpr int PER_THREAD = UNITS / THREADS;
  static_assert(UNITS % THREADS == 0, "tile must divide over the workgroup");
  static constexpr int LDS_LD = CONTIG ? (BK + 4) : ROWS;   // +4 floats: conflict-free ds_read_b128
  static constexpr int LDS_FLOATS = CONTIG ? ROWS * (BK + 4) : BK * ROWS;

  template <bool VEC>
  static __device__ __forceinline__ void load(float4 (&r)[PER_THREAD], const float* __restrict__ src, int64_t ld,
                                              int64_t row0, int64_t nrows, int64_t k0, int64_t kend) {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int u = threadIdx.x + i * THREADS;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (CONTIG) {
        const int row = u / (BK / 4), kq = u % (BK / 4);
        const int64_t gr = row0 + row, gk = k0 + 4 * kq;
        if (gr < nrows) {
          const float* p = src + gr * ld + gk;
          if (VEC && gk + 3 < kend) v = *reinterpret_cast<const float4*>(p);
          else {
            if (gk < kend) v.x = p[0];
            if (gk + 1 < kend) v.y = p[1];
            if (gk + 2 < kend) v.z = p[2];
            if (gk + 3 < kend) v.w = p[3];
          }
        }
      } else {
        const int k = u / (ROWS / 4), rq = u % (ROWS / 4);
        const int64_t gk = k0 + k, gr = row0 + 4 * rq;
        if (gk < kend) {
          const float* p = src + gk * ld + gr;
          if (VEC && gr + 3 < nrows) v = *reinterpret_cast<const float4*>(p);
          else {
            if (gr < nrows) v.x = p[0];
            if (gr + 1 < nrows) v.y = p[1];
            if (gr + 2 < nrows) v.z = p[2];
            if (gr + 3 < nrows) v.w = p[3];
          }
        }
      }
      r[i] = v;
    }
  }

  static __device__ __forceinline__ void store(const float4 (&r)[PER_THREAD], float* lds) {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int u = threadIdx.x + i * THREADS;
      if (CONTIG) {
        const int row = u / (BK / 4), kq = u % (BK / 4);
        *reinterpret_cast<float4*>(&lds[row * LDS_LD + 4 * kq]) = r[i];
      } else {
        const int k = u / (ROWS / 4), rq = u % (ROWS / 4);
        *reinterpret_cast<float4*>(&lds[k * LDS_LD + 4 * rq]) = r[i];
      }
    }
  }

  // the four k-values (8s + 4h + t, t = 0..3) of row `row` for this lane's half h
  static __device__ __forceinline__ void frag(float (&f)[4], const float* lds, int row, int s, int h) {
    if (CONTIG) {
      const float4 v = *reinterpret_cast<const float4*>(&lds[row * LDS_LD + 8 * s + 4 * h]);
      f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) f[t] = lds[(8 * s + 4 * h + t) * LDS_LD + row];
    }
  }
};

template <int BM, int BN, int WM, int WN, bool A_CONTIG, bool B_CONTIG, int EPI, bool VEC>
__global__ __launch_bounds__(THREADS) void gemm_k(Args g) {
  using TA = Tile<BM, A_CONTIG>;
  using TB = Tile<BN, B_CONTIG>;
  static_assert(WM * WN == THREADS / 64, "4 waves");
  constexpr int TM = BM / WM, TN = BN / WN;        // wave tile
  constexpr int NBM = TM / 32, NBN = TN / 32;      // 32x32 accumulator blocks per wave
  static_assert(TM % 32 == 0 && TN % 32 == 0, "wave tile must be a multiple of 32");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int STAGE = TA::LDS_FLOATS + TB::LDS_FLOATS;   // one pipeline stage: [A tile][B tile]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, l31 = lane & 31;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  int64_t kbeg = 0, kend = g.Kc;
  if (EPI == EPI_SLAB) {
    kbeg = (int64_t)blockIdx.z * g.k_per_split;
    kend = min(g.Kc, kbeg + g.k_per_split);
  }

  f32x16 acc[NBM][NBN];
#pragma unroll
  for (int i = 0; i < NBM; ++i)
#pragma unroll
    for (int j = 0; j < NBN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float colsum = 0.f;  // wgrad: db partial for A_op row (threadIdx.x < BM), only blockIdx.x == 0

  float4 ra[TA::PER_THREAD], rb[TB::PER_THREAD];
  const int ntiles = (int)((kend - kbeg + BK - 1) / BK);
  if (ntiles > 0) {
    TA::template load<VEC>(ra, g.A, g.lda, m0, g.M, kbeg, kend);
    TB::template load<VEC>(rb, g.B, g.ldb, n0, g.N, kbeg, kend);
    TA::store(ra, smem);
    TB::store(rb, smem + TA::LDS_FLOATS);
  }
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntiles) {
      const int64_t k0 = kbeg + (int64_t)(t + 1) * BK;
      TA::template load<VEC>(ra, g.A, g.lda, m0, g.M, k0, kend);
      TB::template load<VEC>(rb, g.B, g.ldb, n0, g.N, k0, kend);
    }
    const float* a_s = smem + cur * STAGE;
    const float* b_s = a_s + TA::LDS_FLOATS;
#pragma unroll
    for (int s = 0; s < BK / 8; ++s) {
      float af[NBM][4], bf[NBN][4];
#pragma unroll
      for (int i = 0; i < NBM; ++i) TA::frag(af[i], a_s, wm * TM + i * 32 + l31, s, h);
#pragma unroll
      for (int j = 0; j < NBN; ++j) TB::frag(bf[j], b_s, wn * TN + j * 32 + l31, s, h);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int i = 0; i < NBM; ++i)
#pragma unroll
          for (int j = 0; j < NBN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][tt], bf[j][tt], acc[i][j], 0, 0, 0);
    }
    if (EPI == EPI_SLAB && g.dbias_slab && blockIdx.x == 0 && !A_CONTIG && threadIdx.x < BM) {
#pragma unroll 8
      for (int k = 0; k < BK; ++k) colsum += a_s[k * TA::LDS_LD + threadIdx.x];
    }
    if (t + 1 < ntiles) {
      TA::store(ra, smem + (cur ^ 1) * STAGE);
      TB::store(rb, smem + (cur ^ 1) * STAGE + TA::LDS_FLOATS);
    }
    __syncthreads();
  }

  // ---- epilogue: C/D layout of 32x32 blocks: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* Cbase = g.C;
  if (EPI == EPI_SLAB) Cbase += (int64_t)blockIdx.z * g.M * g.ldc;
#pragma unroll
  for (int i = 0; i < NBM; ++i) {
#pragma unroll
    for (int j = 0; j < NBN; ++j) {
      const int64_t col = n0 + wn * TN + j * 32 + l31;
      if (col >= g.N) continue;
      float bv = 0.f;
      if (EPI == EPI_BIAS_ACT && g.bias) bv = g.bias[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= g.M) continue;
        float v = acc[i][j][r];
        if (EPI == EPI_BIAS_ACT) {
          v += bv;
          if (g.leaky) v = v > 0.f ? v : v * g.slope;
        } else if (EPI == EPI_DACT) {
          if (g.xact) v *= (g.xact[row * g.ldxa + col] > 0.f ? 1.f : g.slope);
        }
        Cbase[row * g.ldc + col] = v;
      }
    }
  }
  if (EPI == EPI_SLAB && g.dbias_slab && blockIdx.x == 0 && !A_CONTIG && threadIdx.x < BM) {
    const int64_t row = m0 + threadIdx.x;
    if (row < g.M) g.dbias_slab[(int64_t)blockIdx.z * g.M + row] = colsum;
  }
}

// dW[i][j] (+)= sum_s slab[s][i][j];  db[i] (+)= sum_s dbslab[s][i]
__global__ __launch_bounds__(THREADS) void slab_reduce_k(const float* __restrict__ slab, int splits, int64_t M, int64_t N,
                                                        float* __restrict__ dW, int64_t lddw,
                                                        const float* __restrict__ dbslab, float* __restrict__ db,
                                                        int accumulate) {
  const int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  const int64_t total = M * N;
  if (idx < total) {
    float t = 0.f;
    for (int s = 0; s < splits; ++s) t += slab[(int64_t)s * total + idx];
    const int64_t i = idx / N, j = idx - i * N;
    float* dst = dW + i * lddw + j;
    *dst = accumulate ? (*dst + t) : t;
  }
  if (db && idx < M) {
    float t = 0.f;
    for (int s = 0; s < splits; ++s) t += dbslab[(int64_t)s * M + idx];
    db[idx] = accumulate ? (db[idx] + t) : t;
  }
}

template <int BM, int BN, bool A_CONTIG, bool B_CONTIG>
constexpr size_t lds_bytes() {
  return 2 * (Tile<BM, A_CONTIG>::LDS_FLOATS + Tile<BN, B_CONTIG>::LDS_FLOATS) * sizeof(float);
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int BM, int BN, int WM, int WN, bool A_CONTIG, bool B_CONTIG, int EPI>
static int launch(const Args& g, int splits, hipStream_t st, const char* who) {
  const bool vec = aligned16(g.A) && aligned16(g.B) && (g.lda % 4 == 0) && (g.ldb % 4 == 0);
  dim3 grid((unsigned)ceil_div(g.N, BN), (unsigned)ceil_div(g.M, BM), (unsigned)splits), block(THREADS);
  constexpr size_t lds = lds_bytes<BM, BN, A_CONTIG, B_CONTIG>();
  if (vec) {
    auto k = gemm_k<BM, BN, WM, WN, A_CONTIG, B_CONTIG, EPI, true>;
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    hipLaunchKernelGGL(k, grid, block, lds, st, g);
  } else {
    auto k = gemm_k<BM, BN, WM, WN, A_CONTIG, B_CONTIG, EPI, false>;
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    hipLaunchKernelGGL(k, grid, block, lds, st, g);
  }
  return launch_status(who);
}

// contraction split for wgrad: enough workgroups for ~2 per CU, each with >= 4 K tiles
static int wgrad_splits(int64_t M, int64_t N, int64_t Kc, int bm, int bn) {
  const int64_t tiles = ceil_div(M, bm) * ceil_div(N, bn);
  int64_t want = ceil_div((int64_t)kNumCU * 2, tiles);
  const int64_t max_s = ceil_div(Kc, (int64_t)BK * 4);
  if (want > max_s) want = max_s;
  if (want < 1) want = 1;
  return (int)want;
}

}  // namespace gemm
}  // namespace clica

using namespace clica;
using namespace clica::gemm;

extern "C" int clica_linear_fwd(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias,
                                float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                                int32_t leaky, float slope, clica_stream_t stream) {
  CLICA_CHECK_ARG(X && W && Y, "clica_linear_fwd: NULL pointer");
  CLICA_CHECK_ARG(M > 0 && N > 0 && K > 0, "clica_linear_fwd: M=%lld N=%lld K=%lld must be positive", (long long)M, (long long)N, (long long)K);
  CLICA_CHECK_ARG(ldx >= K && ldw >= K && ldy >= N, "clica_linear_fwd: leading dimension too small");
  Args g{}; g.A = X; g.lda = ldx; g.B = W; g.ldb = ldw; g.C = Y; g.ldc = ldy; g.M = M; g.N = N; g.Kc = K;
  g.bias = bias; g.slope = slope; g.leaky = leaky;
  return launch<128, 128, 2, 2, true, true, EPI_BIAS_ACT>(g, 1, as_stream(stream), "clica_linear_fwd");
}

extern "C" int clica_linear_dgrad(const float* dY, int64_t lddy, const float* W, int64_t ldw,
                                  const float* Xact, int64_t ldxa, float slope,
                                  float* dX, int64_t lddx, int64_t M, int64_t N, int64_t K,
                                  clica_stream_t stream) {
  CLICA_CHECK_ARG(dY && W && dX, "clica_linear_dgrad: NULL pointer");
  CLICA_CHECK_ARG(M > 0 && N > 0 && K > 0, "clica_linear_dgrad: sizes must be positive");
  CLICA_CHECK_ARG(lddy >= N && ldw >= K && lddx >= K && (!Xact || ldxa >= K), "clica_linear_dgrad: leading dimension too small");
  // dX[M,K] = dY[M,N] W[N,K]: contraction over N; B_op[kc=n][j=k] = W[n][k] (Kc strided)
  Args g{}; g.A = dY; g.lda = lddy; g.B = W; g.ldb = ldw; g.C = dX; g.ldc = lddx; g.M = M; g.N = K; g.Kc = N;
  g.xact = Xact; g.ldxa = ldxa; g.slope = slope;
  return launch<128, 128, 2, 2, true, false, EPI_DACT>(g, 1, as_stream(stream), "clica_linear_dgrad");
}

extern "C" int clica_linear_wgrad_workspace_bytes(int64_t M, int64_t N, int64_t K, size_t* bytes) {
  CLICA_CHECK_ARG(bytes && M > 0 && N > 0 && K > 0, "clica_linear_wgrad_workspace_bytes: bad argument");
  const int s = wgrad_splits(N, K, M, 128, 128);
  *bytes = align_up((size_t)s * N * K * sizeof(float), 256) + align_up((size_t)s * N * sizeof(float), 256);
  return CLICA_OK;
}

extern "C" int clica_linear_wgrad(const float* dY, int64_t lddy, const float* X, int64_t ldx,
                                  float* dW, int64_t lddw, float* db, int64_t M, int64_t N, int64_t K,
                                  int32_t accumulate, void* workspace, size_t workspace_bytes,
                                  clica_stream_t stream) {
  CLICA_CHECK_ARG(dY && X && dW && workspace, "clica_linear_wgrad: NULL pointer");
  CLICA_CHECK_ARG(M > 0 && N > 0 && K > 0, "clica_linear_wgrad: sizes must be positive");
  CLICA_CHECK_ARG(lddy >= N && ldx >= K && lddw >= K, "clica_linear_wgrad: leading dimension too small");
  size_t need = 0;
  clica_linear_wgrad_workspace_bytes(M, N, K, &need);
  if (need > workspace_bytes) { set_error("clica_linear_wgrad: workspace %zu < %zu", workspace_bytes, need); return CLICA_E_WORKSPACE; }
  const int s = wgrad_splits(N, K, M, 128, 128);
  float* slab = (float*)workspace;
  float* dbslab = (float*)((char*)workspace + align_up((size_t)s * N * K * sizeof(float), 256));
  // dW[N,K] = dY[M,N]^T X[M,K]: "M" = N, "N" = K, contraction over the batch rows M
  Args g{}; g.A = dY; g.lda = lddy; g.B = X; g.ldb = ldx; g.C = slab; g.ldc = K; g.M = N; g.N = K; g.Kc = M;
  g.k_per_split = ceil_div(ceil_div(M, (int64_t)s), (int64_t)BK) * BK;
  g.dbias_slab = db ? dbslab : nullptr;
  hipStream_t st = as_stream(stream);
  int rc = launch<128, 128, 2, 2, false, false, EPI_SLAB>(g, s, st, "clica_linear_wgrad");
  if (rc) return rc;
  const int64_t total = N * K;
  hipLaunchKernelGGL(slab_reduce_k, dim3((unsigned)ceil_div(total > N ? total : N, THREADS)), dim3(THREADS), 0, st,
                     (const float*)slab, s, N, K, dW, lddw, (const float*)dbslab, db, accumulate ? 1 : 0);
  return launch_status("clica_linear_wgrad(reduce)");
}
