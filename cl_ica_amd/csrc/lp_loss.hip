// C ABI of the Lp InfoNCE loss (include/clica.h) -- planning, workspace carve-up, launches.
// Kernels: lp_kernels.h; per-exponent instantiations: lp_loss_pk.hip.
#include "lp_kernels.h"
#include "lp_finalize.h"
#include "lp_mfma.h"
#include "lp_mfma_dev.h"

namespace clica {
namespace lp {

__global__ __launch_bounds__(64) void means_k(const float* __restrict__ blocksums, int nblocks, float inv_count,
                                              float* __restrict__ means) {
  // lane l sums slots l, l+64, ... ; fixed-order shuffle tree afterwards
  float v[3] = {0.f, 0.f, 0.f};
  for (int b = threadIdx.x; b < nblocks; b += 64) {
    v[0] += blocksums[b * 3 + 0]; v[1] += blocksums[b * 3 + 1]; v[2] += blocksums[b * 3 + 2];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[a] += __shfl_down(v[a], off, 64);
  }
  if (threadIdx.x == 0) { means[0] = v[0] * inv_count; means[1] = v[1] * inv_count; means[2] = v[2] * inv_count; }
}

// one rank, p = 2 on the matrix cores (lp_mfma.hip): the finalize block of 64 rows = two pool tiles also writes their feature planes
// (they carry u_j = C_j 2^-L_j, which this kernel has just computed) -- the separate plane launch of the backward call disappears
struct FeatOut { lp2::u32x4_t* FP = nullptr; const float* origin = nullptr; float pre2 = 0.f; int64_t pool_tiles = 0; };
// the matrix-core guard (lp_mfma.h): which of the two sweeps wrote the partials is decided on the device per call -- the finishing
// kernels read the same word and take the split count of the sweep that ran; the forward's also counts the fallbacks
struct Gate { float* words = nullptr; float limit = 0.f; int nsplit_alt = 0; };

__global__ __launch_bounds__(THREADS) void fwd_finalize_k(
    const float2* __restrict__ part, int nsplit, int64_t rows,
    const float* __restrict__ z1, int64_t ld1, const float* __restrict__ z2, int64_t ld2,
    Params q, float tau, float alpha, int compat, int frac, int dot, float log_b3,
    float* __restrict__ loss_i, float* __restrict__ pos_i, float* __restrict__ lse_i, Means M, TrainOut T, FeatOut F, Gate G) {
  if (G.words && lp2::guard_falls_back(G.words, G.limit)) {
    nsplit = G.nsplit_alt;
    if (blockIdx.x == 0 && threadIdx.x == 0) G.words[lp2::W_FALLBACKS] += 1.f;
  }
  __shared__ float sm[THREADS / FIN_ROWS][FIN_ROWS], ss[THREADS / FIN_ROWS][FIN_ROWS];
  __shared__ float ush[FIN_ROWS];
  constexpr int FIN_MAXN = 64;
  __shared__ float zrows[2][FIN_ROWS * FIN_MAXN];
  float v_loss, v_pos, v_lse;
  finalize_rows((int)blockIdx.x, part, nsplit, rows, z1, ld1, z2, ld2, q, tau, alpha, compat, frac, dot, log_b3, loss_i, pos_i, lse_i, T,
                FinScratch{&sm[0][0], &ss[0][0], zrows[0], zrows[1], FIN_MAXN, F.FP ? ush : nullptr}, v_loss, v_pos, v_lse);
  if (F.FP) {      // (uniform over the launch; `staged` holds: n <= 14)
    static_assert(THREADS == 2 * 128 && FIN_ROWS == 2 * lp2::ROWS, "a finalize block = two pool tiles of 128 feature vectors each");
    __syncthreads();
    const int id = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * FIN_ROWS;
    lp2::feat_vectors((int64_t)blockIdx.x * 2 + (id >> 7), (id >> 6) & 1, (id >> 5) & 1, id & 31, rows, q.n, F.origin, F.pre2,
                      [&](int64_t j, int f) { return zrows[0][(j - r0) * q.n + f]; },
                      [&](int64_t j) { return ush[j - r0]; }, F.FP);
    if (blockIdx.x == gridDim.x - 1) {      // padding tiles of the plan (masked rows: their e_ij are exact zeros, the planes must be finite)
      const lp2::u32x4_t z4 = {0u, 0u, 0u, 0u};
      for (int64_t v = (int64_t)gridDim.x * 2 * lp2::FEATVEC + id; v < F.pool_tiles * lp2::FEATVEC; v += THREADS) F.FP[v] = z4;
    }
  }
  reduce_means(v_loss, v_pos, v_lse, M, (int)blockIdx.x);
}

// ---- backward ------------------------------------------------------------------------------
// per-row coefficients + the positive-pair gradient
//   statL[i] = lse_i[i] (log2 domain);  statC[i] = -(C_i / tau)  with C_i the upstream weight of the row's log-sum-exp
__global__ __launch_bounds__(THREADS) void bwd_coef_k(
    int64_t rows, const float* __restrict__ z1, int64_t ld1, const float* __restrict__ z2, int64_t ld2,
    Params q, float tau, float alpha, int compat, int frac, int dot, const float* __restrict__ lse_i,
    const float* __restrict__ g_mean, const float* __restrict__ g_item,
    const float* __restrict__ g_pos, const float* __restrict__ g_neg,
    float* __restrict__ statL, float* __restrict__ statC,
    float* __restrict__ dz1, int64_t ldd1, float* __restrict__ dz2, int64_t ldd2) {
  const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  if (i >= rows) return;
  coef_row(i, rows, z1 + i * ld1, z2 + i * ld2, q, tau, alpha, compat, frac, dot, lse_i[i], g_mean, g_item, g_pos, g_neg, statL, statC,
           dz1, ldd1, dz2, ldd2);
}

// out[i,k] (+)= sum_split part[split][i][k]; FOUR threads per float4 of the padded row (each sums every fourth
// split with its loads in flight together, then a fixed-order shuffle tree): 4x the parallelism of a
// launch that is otherwise a few dozen latency-bound workgroups
struct MeansJob { const float* blocksums; int nblocks; float inv_count; float* means; int block; int32_t* tick; };   // block < 0: none
__global__ __launch_bounds__(THREADS) void bwd_reduce_k(const float* __restrict__ part, int nsplit, int64_t rows,
                                                       int np, int n, float* __restrict__ out, int64_t ldo,
                                                       int accumulate, MeansJob mj, Gate G) {
  if (G.words && lp2::guard_falls_back(G.words, G.limit)) nsplit = G.nsplit_alt;
  if ((int)blockIdx.x == mj.block) {        // training step: the forward's three means, off its critical path (see means_k)
    if (threadIdx.x < 64) {
      float v[3] = {0.f, 0.f, 0.f};
      for (int b = threadIdx.x; b < mj.nblocks; b += 64) {
        v[0] += mj.blocksums[b * 3 + 0]; v[1] += mj.blocksums[b * 3 + 1]; v[2] += mj.blocksums[b * 3 + 2];
      }
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[a] += __shfl_down(v[a], off, 64);
      if (threadIdx.x == 0) {
        mj.means[0] = v[0] * mj.inv_count; mj.means[1] = v[1] * mj.inv_count; mj.means[2] = v[2] * mj.inv_count;
        if (mj.tick) mj.tick[0] += 1;       // the step's samplers have consumed the counter; whoever reads it later sees step + 1
      }
    }
    return;
  }
  const int64_t tid = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  const int64_t idx = tid >> 2;
  const int sub = (int)(tid & 3);
  const int q4 = np / 4;
  const bool live = idx < rows * q4;
  const int64_t i = live ? idx / q4 : 0;
  const int k = live ? (int)(idx - i * q4) * 4 : 0;
  const int64_t stride = rows * (int64_t)np;
  const float* src = part + i * np + k;
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live && k < n) {
#pragma unroll 4
    for (int sp = sub; sp < nsplit; sp += 4) {
      const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)sp * stride);
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
  }
#pragma unroll
  for (int off = 1; off <= 2; off <<= 1) {
    t.x += __shfl_xor(t.x, off, 64); t.y += __shfl_xor(t.y, off, 64);
    t.z += __shfl_xor(t.z, off, 64); t.w += __shfl_xor(t.w, off, 64);
  }
  if (!live || k >= n || sub != 0) return;
  float* dst = out + i * ldo + k;
  const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (k + u < n) dst[u] = accumulate ? (dst[u] + tv[u]) : tv[u];
}

static Params make_params(const clica_lp_loss_desc* d, bool frac) {
  Params q;
  q.p = d->p; q.inv_p = 1.f / d->p; q.kscale = kLog2e / d->tau;
  q.sgn = frac ? -1.f : 1.f; q.eps = frac ? 1e-12f : 0.f; q.xs = -1.f;
  q.pow = d->pow ? 1 : 0; q.n = d->n;
  return q;
}
// training-pair forms (bits; all on): 1 = fixed-maximum forward sweep, 2 = folded backward coefficient (Params::train),
// 4 = p = 2 sweeps on the matrix cores where lp_mfma.hip's policy admits them (needs the semantics of bits 1 and 2: the pool contains the anchors)
static constexpr int train_flags() { return 7; }
static int exponent_kind(float p) { return p == 1.f ? 1 : (p == 2.f ? 2 : (p == 3.f ? 3 : 0)); }

static void launch_fwd_partial(const Plan& P, int pk, const float* own, int64_t ldo, int64_t n_own,
                               const float* str, int64_t lds, int64_t n_str, const Params& q,
                               float2* part, float* part_g, hipStream_t st) {
  switch (pk) {
    case 1: launch_fwd_partial_pk1(P, own, ldo, n_own, str, lds, n_str, q, part, part_g, st); break;
    case 2: launch_fwd_partial_pk2(P, own, ldo, n_own, str, lds, n_str, q, part, part_g, st); break;
    case 3: launch_fwd_partial_pk3(P, own, ldo, n_own, str, lds, n_str, q, part, part_g, st); break;
    case 4: launch_fwd_partial_pk4(P, own, ldo, n_own, str, lds, n_str, q, part, part_g, st); break;
    default: launch_fwd_partial_pk0(P, own, ldo, n_own, str, lds, n_str, q, part, part_g, st); break;
  }
}

// rowgrad[i,k] = sum_split G[split][i][k] * 2^(m[split][i] - lse_i*log2e): merge of the flash-style
// row-gradient partials (fwd_partial_k<..., ROWGRAD>) once the row's log-sum-exp is known
__global__ __launch_bounds__(THREADS) void rowgrad_combine_k(const float2* __restrict__ part, const float* __restrict__ part_g,
                                                            int nsplit, int64_t rows, int np, int n,
                                                            const float* __restrict__ lse_i, float* __restrict__ rowgrad,
                                                            int64_t ldrg) {
  const int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  const int q4 = np / 4;
  if (idx >= rows * q4) return;
  const int64_t i = idx / q4;
  const int k = (int)(idx - i * q4) * 4;
  if (k >= n) return;
  const float L2 = lse_i[i];
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int sp = 0; sp < nsplit; ++sp) {
    const float w = fexp2(part[(int64_t)sp * rows + i].x - L2);
    const float4 g = *reinterpret_cast<const float4*>(part_g + ((int64_t)sp * rows + i) * np + k);
    t.x = fmaf(g.x, w, t.x); t.y = fmaf(g.y, w, t.y); t.z = fmaf(g.z, w, t.z); t.w = fmaf(g.w, w, t.w);
  }
  const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (k + u < n) rowgrad[i * ldrg + k + u] = tv[u];
}

// out[i,k] (+)= statC[i] * rowgrad[i,k]   (backward of the row side without a pair pass)
__global__ __launch_bounds__(THREADS) void rowgrad_apply_k(const float* __restrict__ rowgrad, int64_t ldrg,
                                                          const float* __restrict__ statC, int64_t rows, int n,
                                                          float* __restrict__ out, int64_t ldo, int accumulate) {
  const int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  if (idx >= rows * n) return;
  const int64_t i = idx / n;
  const int k = (int)(idx - i * n);
  const float v = statC[i] * rowgrad[i * ldrg + k];
  float* dst = out + i * ldo + k;
  *dst = accumulate ? (*dst + v) : v;
}
static void launch_bwd_pairs(bool owner_stats, const Plan& P, int pk, const float* own, int64_t ldo, int64_t n_own,
                             const float* str, int64_t lds, int64_t n_str, const Params& q,
                             const float* statL, const float* statC, float* part, hipStream_t st) {
  switch (pk) {
    case 1: launch_bwd_pairs_pk1(P, owner_stats, own, ldo, n_own, str, lds, n_str, q, statL, statC, part, st); break;
    case 2: launch_bwd_pairs_pk2(P, owner_stats, own, ldo, n_own, str, lds, n_str, q, statL, statC, part, st); break;
    case 3: launch_bwd_pairs_pk3(P, owner_stats, own, ldo, n_own, str, lds, n_str, q, statL, statC, part, st); break;
    case 4: launch_bwd_pairs_pk4(P, owner_stats, own, ldo, n_own, str, lds, n_str, q, statL, statC, part, st); break;
    default: launch_bwd_pairs_pk0(P, owner_stats, own, ldo, n_own, str, lds, n_str, q, statL, statC, part, st); break;
  }
}

static void launch_bwd_sym(const Plan& P, int pk, const float* own, int64_t ldo, int64_t n_own, const float* str, int64_t lds,
                           int64_t n_str, const Params& q, const float* ownL, const float* ownC, const float* strL,
                           const float* strC, float* part, hipStream_t st) {
  switch (pk) {
    case 1: launch_bwd_sym_pk1(P, own, ldo, n_own, str, lds, n_str, q, ownL, ownC, strL, strC, part, st); break;
    case 2: launch_bwd_sym_pk2(P, own, ldo, n_own, str, lds, n_str, q, ownL, ownC, strL, strC, part, st); break;
    case 3: launch_bwd_sym_pk3(P, own, ldo, n_own, str, lds, n_str, q, ownL, ownC, strL, strC, part, st); break;
    case 4: launch_bwd_sym_pk4(P, own, ldo, n_own, str, lds, n_str, q, ownL, ownC, strL, strC, part, st); break;
    default: launch_bwd_sym_pk0(P, own, ldo, n_own, str, lds, n_str, q, ownL, ownC, strL, strC, part, st); break;
  }
}

// statistics of the pool rows for the symmetric sweep: every pool row carries the same upstream weight
// as the local rows (a mean loss on every rank), i.e. C = 2(1-alpha) g_mean / B + g_neg / B
__global__ __launch_bounds__(THREADS) void pool_stats_k(int64_t pool_rows, const float* __restrict__ pool_lse, int64_t local_rows,
                                                       float tau, float alpha, const float* __restrict__ g_mean,
                                                       const float* __restrict__ g_neg, float xs,
                                                       float* __restrict__ strL, float* __restrict__ strC) {
  const int64_t j = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  if (j >= pool_rows) return;
  const float inv = 1.f / (float)local_rows;
  const float C = 2.f * (1.f - alpha) * (g_mean ? g_mean[0] : 1.f) * inv + (g_neg ? g_neg[0] * inv : 0.f);
  strL[j] = pool_lse[j];
  strC[j] = xs * C / tau;
}

// workspace carve-up ------------------------------------------------------------------------------
struct FwdWs { float2* part; float* part_g; float* blocksums; size_t bytes; };
struct BwdWs { float* statL; float* statC; float* partR; float* partC; size_t bytes; };

// Header of every loss workspace: [0, 64) reserved, then the arrival counters of the one-launch forward (lp_finalize.h: one int per
// 64-row owner tile + one for the means).  ZERO before the first call on a workspace; every launch leaves them zero; nothing else in
// this file writes the header (the backward / training carves start behind it), so forward and backward calls may share a workspace.
constexpr size_t kHeaderBytes = 4096;
constexpr int64_t kArriveMaxTiles = (int64_t)(kHeaderBytes - 64) / 4 - 1;
static int* header_arrive(void* ws) { return reinterpret_cast<int*>((char*)ws + 64); }
static FwdWs carve_fwd(void* ws, const Plan& P, int64_t rows, bool rowgrad) {
  FwdWs w; char* p = (char*)ws; size_t off = 0;
  off += kHeaderBytes;
  w.blocksums = (float*)(p + off); off += align_up((size_t)ceil_div(rows, FIN_ROWS) * 3 * sizeof(float), 256);
  w.part = (float2*)(p + off); off += align_up((size_t)P.nsplit * rows * sizeof(float2), 256);
  w.part_g = nullptr;
  if (rowgrad) { w.part_g = (float*)(p + off); off += align_up((size_t)P.nsplit * rows * P.np * sizeof(float), 256); }
  w.bytes = off; return w;
}
static size_t fwd_carve_max(int64_t rows, int64_t cols, int n) {   // plain vs row-gradient forward, whichever is larger
  const size_t a = carve_fwd(nullptr, make_plan(rows, cols, n, false), rows, false).bytes;
  const size_t b = carve_fwd(nullptr, make_plan(rows, cols, n, true), rows, true).bytes;
  return a > b ? a : b;
}
static BwdWs carve_bwd(void* ws, const Plan& PR, const Plan& PC, int64_t rows, int64_t cols) {
  BwdWs w; char* p = (char*)ws; size_t off = kHeaderBytes;   // keep clear of the forward's arrival counters
  w.statL = (float*)(p + off); off += align_up((size_t)rows * sizeof(float), 256);
  w.statC = (float*)(p + off); off += align_up((size_t)rows * sizeof(float), 256);
  w.partR = (float*)(p + off); off += align_up((size_t)PR.nsplit * rows * PR.np * sizeof(float), 256);
  w.partC = (float*)(p + off); off += align_up((size_t)PC.nsplit * cols * PC.np * sizeof(float), 256);
  w.bytes = off; return w;
}

static int validate(const clica_lp_loss_desc* d, const char* who) {
  CLICA_CHECK_ARG(d != nullptr, "%s: desc is NULL", who);
  CLICA_CHECK_ARG(d->B > 0 && d->B3 > 0, "%s: B=%lld B3=%lld must be positive", who, (long long)d->B, (long long)d->B3);
  CLICA_CHECK_ARG(d->n >= 1, "%s: n=%d must be >= 1", who, d->n);
  CLICA_CHECK_ARG(pad_dim(d->n) > 0, "%s: n=%d > %d (register-resident kernels to 64, wide-row kernels beyond)", who, d->n, kMaxWideN);
  CLICA_CHECK_ARG(d->p > 0.f, "%s: p=%g must be > 0", who, d->p);
  CLICA_CHECK_ARG(d->tau > 0.f, "%s: tau=%g must be > 0", who, d->tau);
  if (d->p < 1.f && !d->no_eps)
    CLICA_CHECK_ARG(d->B == d->B3, "%s: p<1 uses the transposed branch (losses.py:433-442) and needs B3 == B", who);
  return CLICA_OK;
}

}  // namespace lp
}  // namespace clica

using namespace clica;
using namespace clica::lp;

// In the p < 1 branch the pair-matrix rows are z3 rows and the columns z1 rows.
extern "C" int clica_lp_loss_workspace_bytes(const clica_lp_loss_desc* d, size_t* fwd_bytes, size_t* bwd_bytes) {
  int rc = validate(d, "clica_lp_loss_workspace_bytes");
  if (rc) return rc;
  const bool frac = d->p < 1.f && !d->no_eps;
  const int64_t rows = frac ? d->B3 : d->B, cols = frac ? d->B : d->B3;
  Plan PF = make_plan(rows, cols, d->n, false);
  Plan PR = make_plan(rows, cols, d->n, true);
  Plan PC = make_plan(cols, rows, d->n, true);
  if (fwd_bytes) *fwd_bytes = fwd_carve_max(rows, cols, d->n);
  (void)PF;
  if (bwd_bytes) *bwd_bytes = carve_bwd(nullptr, PR, PC, rows, cols).bytes;
  return CLICA_OK;
}

// clica_set_tuning("lp_fused_finalize", 0): the forwards as sweep + fwd_finalize_k (+ means_k) launches (test / A-B hook -- same bits)
static int& fused_finalize_switch() { static int on = 1; return on; }
namespace clica { namespace lp { void set_fused_finalize(int on) { fused_finalize_switch() = on ? 1 : 0; } } }       // clica_set_tuning (linear.hip)
static bool launch_fwd_partial_fin(const Plan& P, int pk, const float* own, int64_t ldo, int64_t n_own, const float* str, int64_t lds,
                                   int64_t n_str, const Params& q, float2* part, const FinArgs& F, hipStream_t st) {
  switch (pk) {
    case 1: return launch_fwd_partial_fin_pk1(P, own, ldo, n_own, str, lds, n_str, q, part, F, st);
    case 2: return launch_fwd_partial_fin_pk2(P, own, ldo, n_own, str, lds, n_str, q, part, F, st);
    case 3: return launch_fwd_partial_fin_pk3(P, own, ldo, n_own, str, lds, n_str, q, part, F, st);
    default: return false;
  }
}
extern "C" int clica_lp_loss_fwd(const clica_lp_loss_desc* d,
                                 const float* z1, int64_t ld1, const float* z2, int64_t ld2,
                                 const float* z3, int64_t ld3,
                                 float* loss_i, float* pos_i, float* lse_i, float* means,
                                 float* rowgrad, int64_t ldrg,
                                 void* workspace, size_t workspace_bytes, clica_stream_t stream) {
  int rc = validate(d, "clica_lp_loss_fwd");
  if (rc) return rc;
  CLICA_CHECK_ARG(z1 && z2 && z3 && loss_i && pos_i && lse_i && means && workspace, "clica_lp_loss_fwd: NULL pointer");
  CLICA_CHECK_ARG(ld1 >= d->n && ld2 >= d->n && ld3 >= d->n, "clica_lp_loss_fwd: leading dimension < n");
  const bool frac = d->p < 1.f && !d->no_eps;
  const float* rows_p = frac ? z3 : z1; const int64_t ldr = frac ? ld3 : ld1; const int64_t rows = frac ? d->B3 : d->B;
  const float* cols_p = frac ? z1 : z3; const int64_t ldc = frac ? ld1 : ld3; const int64_t cols = frac ? d->B : d->B3;
  CLICA_CHECK_ARG(!rowgrad || ldrg >= d->n, "clica_lp_loss_fwd: rowgrad leading dimension < n");
  Plan P = make_plan(rows, cols, d->n, rowgrad != nullptr);
  FwdWs w = carve_fwd(workspace, P, rows, rowgrad != nullptr);
  if (w.bytes > workspace_bytes) { set_error("clica_lp_loss_fwd: workspace %zu < %zu", workspace_bytes, w.bytes); return CLICA_E_WORKSPACE; }
  Params q = make_params(d, frac);
  // The negatives ARE the anchors (same buffer: LpSimCLRLoss with z3_rec = roll(z1_rec), main_mlp.py:272): every logit is <= 0 and a
  // row's own pool entry gives exactly 0 -- the training pair's fixed-maximum sweep applies (Params::train bit 0), as in clica_lp_loss_fwd_train
  if (!frac && !rowgrad && z3 == z1 && ld3 == ld1 && d->B3 == d->B && d->pow && exponent_kind(d->p) != 0 && (train_flags() & 1)) {
    q.train = 1;
    if (exponent_kind(d->p) >= 2) q.pre = powf(q.kscale, 1.f / d->p);
  }
  hipStream_t st = as_stream(stream);
  // ONE launch where that form exists (p in {1, 2, 3}, rows of <= 16 padded coordinates, no row gradient, not the p < 1 branch): the last
  // workgroup of an owner tile finishes its rows, the last finisher the means (lp_finalize.h)
  if (fused_finalize_switch() && !frac && !rowgrad && P.tiles <= kArriveMaxTiles &&
      launch_fwd_partial_fin(P, exponent_kind(d->p), rows_p, ldr, rows, cols_p, ldc, cols, q, w.part,
                             FinArgs{z2, ld2, d->tau, d->alpha, d->compat ? 1 : 0, logf((float)cols), loss_i, pos_i, lse_i, Means{w.blocksums},
                                     TrainOut{}, header_arrive(workspace), means, 1.f / (float)rows}, st))
    return launch_status("clica_lp_loss_fwd");
  launch_fwd_partial(P, exponent_kind(d->p), rows_p, ldr, rows, cols_p, ldc, cols, q, w.part, w.part_g, st);
  Means M{w.blocksums};
  const int nfin = (int)ceil_div(rows, FIN_ROWS);
  hipLaunchKernelGGL(fwd_finalize_k, dim3((unsigned)nfin), dim3(THREADS), 0, st,
                     (const float2*)w.part, P.nsplit, rows, z1, ld1, z2, ld2, q, d->tau, d->alpha,
                     d->compat ? 1 : 0, frac ? 1 : 0, 0, logf((float)cols), loss_i, pos_i, lse_i, M, TrainOut{}, FeatOut{}, Gate{});
  hipLaunchKernelGGL(means_k, dim3(1), dim3(64), 0, st, (const float*)w.blocksums, nfin, 1.f / (float)rows, means);
  if (rowgrad)
    hipLaunchKernelGGL(rowgrad_combine_k, dim3((unsigned)ceil_div(rows * (P.np / 4), THREADS)), dim3(THREADS), 0, st,
                       (const float2*)w.part, (const float*)w.part_g, P.nsplit, rows, P.np, d->n, (const float*)lse_i, rowgrad, ldrg);
  return launch_status("clica_lp_loss_fwd");
}

extern "C" int clica_lp_loss_bwd(const clica_lp_loss_desc* d,
                                 const float* z1, int64_t ld1, const float* z2, int64_t ld2,
                                 const float* z3, int64_t ld3, const float* lse_i,
                                 const float* rowgrad, int64_t ldrg,
                                 const float* g_mean, const float* g_item, const float* g_pos, const float* g_neg,
                                 float* dz1, int64_t ldd1, float* dz2, int64_t ldd2,
                                 float* dz3, int64_t ldd3, int32_t accumulate_dz3,
                                 void* workspace, size_t workspace_bytes, clica_stream_t stream) {
  int rc = validate(d, "clica_lp_loss_bwd");
  if (rc) return rc;
  CLICA_CHECK_ARG(z1 && z2 && z3 && lse_i && workspace, "clica_lp_loss_bwd: NULL pointer");
  const bool frac = d->p < 1.f && !d->no_eps;
  const int64_t rows = frac ? d->B3 : d->B, cols = frac ? d->B : d->B3;
  const float* rows_p = frac ? z3 : z1; const int64_t ldr = frac ? ld3 : ld1;
  const float* cols_p = frac ? z1 : z3; const int64_t ldc = frac ? ld1 : ld3;
  // gradient destinations of the pair-matrix rows / columns
  float* d_rows = frac ? dz3 : dz1; const int64_t ld_dr = frac ? ldd3 : ldd1;
  float* d_cols = frac ? dz1 : dz3; const int64_t ld_dc = frac ? ldd1 : ldd3;
  Plan PR = make_plan(rows, cols, d->n, true);
  Plan PC = make_plan(cols, rows, d->n, true);
  BwdWs w = carve_bwd(workspace, PR, PC, rows, cols);
  if (w.bytes > workspace_bytes) { set_error("clica_lp_loss_bwd: workspace %zu < %zu", workspace_bytes, w.bytes); return CLICA_E_WORKSPACE; }
  Params q = make_params(d, frac);
  const int pk = exponent_kind(d->p);
  hipStream_t st = as_stream(stream);
  // 1. coefficients + positive-pair gradient: dz1 = gp (assign), dz2 = -gp
  hipLaunchKernelGGL(bwd_coef_k, dim3((unsigned)ceil_div(rows, THREADS)), dim3(THREADS), 0, st,
                     rows, z1, ld1, z2, ld2, q, d->tau, d->alpha, d->compat ? 1 : 0, frac ? 1 : 0, 0, lse_i,
                     g_mean, g_item, g_pos, g_neg, w.statL, w.statC, dz1, ldd1, dz2, ldd2);
  // in the p>=1 branch dz1 receives the positive term (assigned above) plus the row pass;
  // in the p<1 branch dz1 is the COLUMN gradient: positive term assigned, column pass added.
  if (d_rows && rowgrad) {   // the forward already produced the softmax-weighted row gradient
    const int acc = frac ? (accumulate_dz3 ? 1 : 0) : 1;
    hipLaunchKernelGGL(rowgrad_apply_k, dim3((unsigned)ceil_div(rows * d->n, THREADS)), dim3(THREADS), 0, st,
                       rowgrad, ldrg, (const float*)w.statC, rows, d->n, d_rows, ld_dr, acc);
  } else if (d_rows) {
    launch_bwd_pairs(true, PR, pk, rows_p, ldr, rows, cols_p, ldc, cols, q, w.statL, w.statC, w.partR, st);
    const int acc = frac ? (accumulate_dz3 ? 1 : 0) : 1;
    hipLaunchKernelGGL(bwd_reduce_k, dim3((unsigned)ceil_div(rows * PR.np, THREADS)), dim3(THREADS), 0, st,
                       (const float*)w.partR, PR.nsplit, rows, PR.np, d->n, d_rows, ld_dr, acc, MeansJob{nullptr, 0, 0.f, nullptr, -1, nullptr}, Gate{});
  }
  if (d_cols) {
    Params qc = q; qc.sgn = -q.sgn;   // e = -(owner - stream) + eps seen from the column side
    launch_bwd_pairs(false, PC, pk, cols_p, ldc, cols, rows_p, ldr, rows, qc, w.statL, w.statC, w.partC, st);
    const int acc = frac ? 1 : (accumulate_dz3 ? 1 : 0);
    hipLaunchKernelGGL(bwd_reduce_k, dim3((unsigned)ceil_div(cols * PC.np, THREADS)), dim3(THREADS), 0, st,
                       (const float*)w.partC, PC.nsplit, cols, PC.np, d->n, d_cols, ld_dc, acc, MeansJob{nullptr, 0, 0.f, nullptr, -1, nullptr}, Gate{});
  }
  return launch_status("clica_lp_loss_bwd");
}

extern "C" int clica_lp_loss_bwd_sym(const clica_lp_loss_desc* d,
                                     const float* z1, int64_t ld1, const float* z2, int64_t ld2,
                                     const float* pool, int64_t ldp, const float* lse_i, const float* pool_lse,
                                     const float* g_mean, const float* g_pos, const float* g_neg,
                                     float* dz1, int64_t ldd1, float* dz2, int64_t ldd2,
                                     void* workspace, size_t workspace_bytes, clica_stream_t stream) {
  int rc = validate(d, "clica_lp_loss_bwd_sym");
  if (rc) return rc;
  CLICA_CHECK_ARG(z1 && z2 && pool && lse_i && pool_lse && dz1 && workspace, "clica_lp_loss_bwd_sym: NULL pointer");
  CLICA_CHECK_ARG(d->p >= 1.f || d->no_eps, "clica_lp_loss_bwd_sym: the p<1 branch is not symmetric (eps inside the abs, losses.py:436)");
  CLICA_CHECK_ARG(d->B3 >= d->B, "clica_lp_loss_bwd_sym: the pool (B3=%lld rows) must contain the %lld local rows", (long long)d->B3, (long long)d->B);
  const int64_t rows = d->B, cols = d->B3;
  Plan PR = make_plan(rows, cols, d->n, true);
  Plan PC = make_plan(cols, rows, d->n, true);
  BwdWs w = carve_bwd(workspace, PR, PC, rows, cols);
  if (w.bytes > workspace_bytes) { set_error("clica_lp_loss_bwd_sym: workspace %zu < %zu", workspace_bytes, w.bytes); return CLICA_E_WORKSPACE; }
  float* strL = w.partC;            // the column-pass slab region is free in this mode (>= 2*cols floats)
  float* strC = w.partC + cols;
  Params q = make_params(d, false);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(bwd_coef_k, dim3((unsigned)ceil_div(rows, THREADS)), dim3(THREADS), 0, st,
                     rows, z1, ld1, z2, ld2, q, d->tau, d->alpha, d->compat ? 1 : 0, 0, 0, lse_i,
                     g_mean, (const float*)nullptr, g_pos, g_neg, w.statL, w.statC, dz1, ldd1, dz2, ldd2);
  if (pool_lse == lse_i && cols == rows) {   // single rank: the pool IS the local rows, their statistics are already there
    strL = w.statL; strC = w.statC;
  } else {
    hipLaunchKernelGGL(pool_stats_k, dim3((unsigned)ceil_div(cols, THREADS)), dim3(THREADS), 0, st,
                       cols, pool_lse, rows, d->tau, d->alpha, g_mean, g_neg, q.xs, strL, strC);
  }
  // the pool contains the owner rows (this entry point's contract): every row statistic is the log2 of a sum >= 1, so the pair sweep
  // takes the folded coefficient 2^x (u_i + u_j) with one exponential per pair (Params::train bit 1), as clica_lp_loss_bwd_sym_train does
  Params qs = q;
  if ((train_flags() & 2) && exponent_kind(d->p) != 0 && d->pow) {
    qs.train = 2;
    if (exponent_kind(d->p) >= 2) qs.pre = powf(qs.kscale, 1.f / d->p);
    qs.gfold = d->p / powf(qs.pre, d->p - 1.f);
  }
  launch_bwd_sym(PR, exponent_kind(d->p), z1, ld1, rows, pool, ldp, cols, qs, w.statL, w.statC, strL, strC, w.partR, st);
  hipLaunchKernelGGL(bwd_reduce_k, dim3((unsigned)ceil_div(rows * PR.np, THREADS)), dim3(THREADS), 0, st,
                     (const float*)w.partR, PR.nsplit, rows, PR.np, d->n, dz1, ldd1, 1, MeansJob{nullptr, 0, 0.f, nullptr, -1, nullptr}, Gate{});
  return launch_status("clica_lp_loss_bwd_sym");
}

// ---- training-step pair of entry points: forward with the coefficient step folded into its finalize, symmetric backward
// with the forward's means folded into its reduction (three launches fewer than fwd + bwd_sym) -------------------------
struct TrainWs { float* blocksums; int* arrive; float* statL; float* statC; float* strL; float* strC; char* scratch; size_t scratch_bytes; size_t bytes;
                 bool mfma; lp2::Plan P2; lp2::Ws w2; };
static bool train_mfma_shape(const clica_lp_loss_desc* d) { return (train_flags() & 7) == 7 && lp2::applies(d->n, d->p, d->pow); }
// (its fallback is the fixed-maximum / folded-coefficient difference sweep)
static bool train_mfma(const clica_lp_loss_desc* d) { return train_mfma_shape(d) && lp2::applies_to_pool(d->B, d->B3); }
static TrainWs carve_train(void* ws, const Plan& PF, const Plan& PR, int64_t rows, int64_t cols, bool mfma) {
  TrainWs w; char* p = (char*)ws; size_t off = kHeaderBytes;
  w.blocksums = (float*)(p + off); off += align_up((size_t)ceil_div(rows, FIN_ROWS) * 3 * sizeof(float), 256);
  w.arrive = (int*)(p + off); off += align_up((size_t)ceil_div(rows, FIN_ROWS) * sizeof(int), 256);    // fused finalize: arrival counters (zero between launches)
  w.statL = (float*)(p + off); off += align_up((size_t)rows * sizeof(float), 256);
  w.statC = (float*)(p + off); off += align_up((size_t)rows * sizeof(float), 256);
  w.strL = (float*)(p + off); off += align_up((size_t)cols * sizeof(float), 256);
  w.strC = (float*)(p + off); off += align_up((size_t)cols * sizeof(float), 256);
  w.mfma = mfma;
  int nsf = PF.nsplit, nsr = PR.nsplit;
  if (mfma) {       // operand planes of the matrix-core sweeps (written by the forward call, read by both)
    w.P2 = lp2::make_plan(rows, cols);
    w.w2 = lp2::carve(p + off, w.P2);
    off += w.w2.bytes;
    nsf = w.P2.nsplit > nsf ? w.P2.nsplit : nsf;       // either sweep may write the partials (the guard decides per call on the device)
    nsr = w.P2.nsplit > nsr ? w.P2.nsplit : nsr;
  }
  w.scratch = p + off;      // forward: per-split (max, sum) partials; backward: per-split gradient partials (the forward's are dead by then)
  const size_t f = align_up((size_t)nsf * rows * sizeof(float2), 256);
  const size_t b = align_up((size_t)nsr * rows * PR.np * sizeof(float), 256);
  w.scratch_bytes = f > b ? f : b;
  w.bytes = off + w.scratch_bytes; return w;
}

extern "C" int clica_lp_loss_train_workspace_bytes(const clica_lp_loss_desc* d, size_t* bytes) {
  int rc = validate(d, "clica_lp_loss_train_workspace_bytes");
  if (rc) return rc;
  CLICA_CHECK_ARG(bytes != nullptr, "clica_lp_loss_train_workspace_bytes: bytes is NULL");
  // (sized for the matrix-core sweeps wherever the shape admits them, whatever the pool policy in force: clica_lp_loss_set_matrix_cores
  //  may switch them on for this workspace later)
  *bytes = carve_train(nullptr, make_plan(d->B, d->B3, d->n, false), make_plan(d->B, d->B3, d->n, true), d->B, d->B3, train_mfma_shape(d)).bytes;
  return CLICA_OK;
}

extern "C" int clica_lp_loss_train_path(const clica_lp_loss_desc* d, int32_t* path) {
  int rc = validate(d, "clica_lp_loss_train_path");
  if (rc) return rc;
  CLICA_CHECK_ARG(path != nullptr, "clica_lp_loss_train_path: path is NULL");
  *path = train_mfma(d) ? 1 : 0;
  return CLICA_OK;
}

extern "C" int clica_lp_loss_set_matrix_cores(int32_t on) {
  lp2::set_enabled(on);
  return CLICA_OK;
}

extern "C" int clica_lp_loss_train_spread(const clica_lp_loss_desc* d, const void* workspace, size_t workspace_bytes, float* spread,
                                          clica_stream_t stream) {
  int rc = validate(d, "clica_lp_loss_train_spread");
  if (rc) return rc;
  CLICA_CHECK_ARG(workspace && spread, "clica_lp_loss_train_spread: NULL pointer");
  *spread = 0.f;
  if (!train_mfma(d)) return CLICA_OK;
  const int64_t rows = d->B, cols = d->B3;
  TrainWs w = carve_train(const_cast<void*>(workspace), make_plan(rows, cols, d->n, false), make_plan(rows, cols, d->n, true), rows, cols, true);
  if (w.bytes > workspace_bytes) { set_error("clica_lp_loss_train_spread: workspace %zu < %zu", workspace_bytes, w.bytes); return CLICA_E_WORKSPACE; }
  hipStream_t st = as_stream(stream);
  if (hipMemcpyAsync(spread, w.w2.spread, sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return launch_status("clica_lp_loss_train_spread");
  return CLICA_OK;
}

extern "C" int clica_lp_loss_set_spread_limit(float limit) {
  lp2::set_spread_limit(limit);
  return CLICA_OK;
}

extern "C" int clica_lp_loss_train_guard(const clica_lp_loss_desc* d, const void* workspace, size_t workspace_bytes, float* out4,
                                         clica_stream_t stream) {
  int rc = validate(d, "clica_lp_loss_train_guard");
  if (rc) return rc;
  CLICA_CHECK_ARG(workspace && out4, "clica_lp_loss_train_guard: NULL pointer");
  out4[0] = out4[1] = out4[3] = 0.f; out4[2] = lp2::spread_limit();
  if (!train_mfma(d)) return CLICA_OK;
  const int64_t rows = d->B, cols = d->B3;
  TrainWs w = carve_train(const_cast<void*>(workspace), make_plan(rows, cols, d->n, false), make_plan(rows, cols, d->n, true), rows, cols, true);
  if (w.bytes > workspace_bytes) { set_error("clica_lp_loss_train_guard: workspace %zu < %zu", workspace_bytes, w.bytes); return CLICA_E_WORKSPACE; }
  hipStream_t st = as_stream(stream);
  float words[16];
  if (hipMemcpyAsync(words, w.w2.spread, sizeof(words), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return launch_status("clica_lp_loss_train_guard");
  out4[0] = words[lp2::W_RUN_M]; out4[1] = words[lp2::W_M64]; out4[3] = words[lp2::W_FALLBACKS];      // (the low word of the tagged M holds the float)
  return CLICA_OK;
}

extern "C" int clica_lp_loss_fwd_train(const clica_lp_loss_desc* d,
                                       const float* z1, int64_t ld1, const float* z2, int64_t ld2, const float* pool, int64_t ldp,
                                       float* loss_i, float* pos_i, float* lse_i,
                                       float* dz1, int64_t ldd1, float* dz2, int64_t ldd2,
                                       void* workspace, size_t workspace_bytes, clica_stream_t stream) {
  int rc = validate(d, "clica_lp_loss_fwd_train");
  if (rc) return rc;
  CLICA_CHECK_ARG(z1 && z2 && pool && loss_i && pos_i && lse_i && dz1 && dz2 && workspace, "clica_lp_loss_fwd_train: NULL pointer");
  CLICA_CHECK_ARG(d->p >= 1.f, "clica_lp_loss_fwd_train: the p<1 branch is not symmetric (eps inside the abs, losses.py:436)");
  CLICA_CHECK_ARG(ld1 >= d->n && ld2 >= d->n && ldp >= d->n && ldd1 >= d->n && ldd2 >= d->n, "clica_lp_loss_fwd_train: leading dimension < n");
  const int64_t rows = d->B, cols = d->B3;
  Plan PF = make_plan(rows, cols, d->n, false), PR = make_plan(rows, cols, d->n, true);
  TrainWs w = carve_train(workspace, PF, PR, rows, cols, train_mfma(d));
  if (w.bytes > workspace_bytes) { set_error("clica_lp_loss_fwd_train: workspace %zu < %zu", workspace_bytes, w.bytes); return CLICA_E_WORKSPACE; }
  Params q = make_params(d, false);
  q.train = train_flags() & 1;       // the pool contains the owner rows: running maximum known (Params::train)
  if (q.train && exponent_kind(d->p) >= 2 && d->pow) q.pre = powf(q.kscale, 1.f / d->p);     // scaled coordinates: the distance sum is -logit (p = 2, 3)
  hipStream_t st = as_stream(stream);
  float2* part = reinterpret_cast<float2*>(w.scratch);
  int nsplit_f = PF.nsplit;
  Gate gate;
  if (w.mfma) {      // p = 2: logits as one augmented inner product on the matrix cores (lp_mfma.hip); same partial format
    const float limit = lp2::spread_limit();
    lp2::launch_prep(w.P2, w.w2, z1, ld1, rows, pool, ldp, cols, d->n, q.kscale, st);       // also measures this call's spread M
    // one launch: the matrix-core sweep, or -- decided by its workgroups from the guard words -- the difference sweep of plan PF
    lp2::launch_fwd(w.P2, w.w2, rows, part, limit, PF, z1, ld1, pool, ldp, cols, q, st);
    nsplit_f = w.P2.nsplit;
    gate = Gate{w.w2.spread, limit, PF.nsplit};
  } else {
    // one launch: the last workgroup of every owner tile finishes the tile's rows (lp_finalize.h); two where that form does not exist
    if (fused_finalize_switch() &&
        launch_fwd_partial_fin(PF, exponent_kind(d->p), z1, ld1, rows, pool, ldp, cols, q, part,
                               FinArgs{z2, ld2, d->tau, d->alpha, d->compat ? 1 : 0, logf((float)cols), loss_i, pos_i, lse_i, Means{w.blocksums},
                                       TrainOut{w.statL, w.statC, dz1, ldd1, dz2, ldd2}, w.arrive, nullptr, 0.f}, st))
      return launch_status("clica_lp_loss_fwd_train");
    launch_fwd_partial(PF, exponent_kind(d->p), z1, ld1, rows, pool, ldp, cols, q, part, nullptr, st);
  }
  Means M{w.blocksums};
  const int nfin = (int)ceil_div(rows, FIN_ROWS);
  FeatOut feat;
  if (w.mfma && pool == z1 && cols == rows && ldp == ld1)      // one rank: see FeatOut; bwd_sym_train makes the same test
    feat = FeatOut{(lp2::u32x4_t*)w.w2.pool_feat, w.w2.spread + lp2::W_ORIGIN_USED, sqrtf(2.f * q.kscale), w.P2.pool_tiles};
  hipLaunchKernelGGL(fwd_finalize_k, dim3((unsigned)nfin), dim3(THREADS), 0, st,
                     (const float2*)part, nsplit_f, rows, z1, ld1, z2, ld2, q, d->tau, d->alpha,
                     d->compat ? 1 : 0, 0, 0, logf((float)cols), loss_i, pos_i, lse_i, M, TrainOut{w.statL, w.statC, dz1, ldd1, dz2, ldd2}, feat, gate);
  return launch_status("clica_lp_loss_fwd_train");
}

static int bwd_sym_train_impl(const clica_lp_loss_desc* d,
                              const float* z1, int64_t ld1, const float* pool, int64_t ldp,
                              const float* lse_i, const float* pool_lse,
                              float* dz1, int64_t ldd1, float* means, int32_t* tick_counter,
                              void* workspace, size_t workspace_bytes, clica_lp_dy_parts* parts, clica_stream_t stream) {
  int rc = validate(d, "clica_lp_loss_bwd_sym_train");
  if (rc) return rc;
  CLICA_CHECK_ARG(z1 && pool && lse_i && pool_lse && (dz1 || parts) && means && workspace, "clica_lp_loss_bwd_sym_train: NULL pointer");
  CLICA_CHECK_ARG(d->p >= 1.f && d->B3 >= d->B, "clica_lp_loss_bwd_sym_train: needs p >= 1 and a pool that contains the local rows");
  const int64_t rows = d->B, cols = d->B3;
  Plan PF = make_plan(rows, cols, d->n, false), PR = make_plan(rows, cols, d->n, true);
  TrainWs w = carve_train(workspace, PF, PR, rows, cols, train_mfma(d));
  if (w.bytes > workspace_bytes) { set_error("clica_lp_loss_bwd_sym_train: workspace %zu < %zu", workspace_bytes, w.bytes); return CLICA_E_WORKSPACE; }
  Params q = make_params(d, false);
  hipStream_t st = as_stream(stream);
  const float* strL = w.statL; const float* strC = w.statC;
  if (!(pool_lse == lse_i && cols == rows)) {        // several ranks: statistics of the whole pool from the gathered lse
    hipLaunchKernelGGL(pool_stats_k, dim3((unsigned)ceil_div(cols, THREADS)), dim3(THREADS), 0, st,
                       cols, pool_lse, rows, d->tau, d->alpha, (const float*)nullptr, (const float*)nullptr, q.xs, w.strL, w.strC);
    strL = w.strL; strC = w.strC;
  }
  float* partR = reinterpret_cast<float*>(w.scratch);
  q.train = train_flags() & 2;       // every row statistic is the log2 of a sum >= 1: folded coefficient (Params::train)
  if (q.train && exponent_kind(d->p) != 0 && d->pow) {
    if (exponent_kind(d->p) >= 2) q.pre = powf(q.kscale, 1.f / d->p);     // (p = 1 keeps unscaled coordinates: sign(d) must be exact)
    q.gfold = d->p / powf(q.pre, d->p - 1.f);
  }
  int nsplit_r = PR.nsplit;
  Gate gate;
  if (w.mfma) {      // the planes of the forward call are still in the workspace (same z1 / pool, as for the row statistics), and so is its spread
    const float limit = lp2::spread_limit();
    lp2::launch_bwd(w.P2, w.w2, z1, ld1, rows, pool, ldp, cols, d->n, PR.np, q.kscale, w.statL, w.statC, strL, strC, partR,
                    /*feat_ready=*/pool == z1 && cols == rows && ldp == ld1 && pool_lse == lse_i, limit, PR, q, st);
    nsplit_r = w.P2.nsplit;
    gate = Gate{w.w2.spread, limit, PR.nsplit};
  } else {
    launch_bwd_sym(PR, exponent_kind(d->p), z1, ld1, rows, pool, ldp, cols, q, w.statL, w.statC, strL, strC, partR, st);
  }
  if (parts) {      // the consumer (clica_mlp_dgrad_split_tail) does the reduction, the means and the tick itself
    *parts = clica_lp_dy_parts{partR, nsplit_r, gate.nsplit_alt, PR.np, d->n, rows, gate.words, gate.limit,
                               w.blocksums, (int32_t)ceil_div(rows, FIN_ROWS), 1.f / (float)rows, means, tick_counter};
    return launch_status("clica_lp_loss_bwd_sym_train_parts");
  }
  const int blocks = (int)ceil_div(rows * PR.np, THREADS);
  hipLaunchKernelGGL(bwd_reduce_k, dim3((unsigned)(blocks + 1)), dim3(THREADS), 0, st,
                     (const float*)partR, nsplit_r, rows, PR.np, d->n, dz1, ldd1, 1,
                     MeansJob{w.blocksums, (int)ceil_div(rows, FIN_ROWS), 1.f / (float)rows, means, blocks, tick_counter}, gate);
  return launch_status("clica_lp_loss_bwd_sym_train");
}
extern "C" int clica_lp_loss_bwd_sym_train(const clica_lp_loss_desc* d,
                                           const float* z1, int64_t ld1, const float* pool, int64_t ldp,
                                           const float* lse_i, const float* pool_lse,
                                           float* dz1, int64_t ldd1, float* means, int32_t* tick_counter,
                                           void* workspace, size_t workspace_bytes, clica_stream_t stream) {
  return bwd_sym_train_impl(d, z1, ld1, pool, ldp, lse_i, pool_lse, dz1, ldd1, means, tick_counter, workspace, workspace_bytes, nullptr, stream);
}
extern "C" int clica_lp_loss_bwd_sym_train_parts(const clica_lp_loss_desc* d,
                                                 const float* z1, int64_t ld1, const float* pool, int64_t ldp,
                                                 const float* lse_i, const float* pool_lse, float* means, int32_t* tick_counter,
                                                 void* workspace, size_t workspace_bytes, clica_lp_dy_parts* parts, clica_stream_t stream) {
  CLICA_CHECK_ARG(parts != nullptr, "clica_lp_loss_bwd_sym_train_parts: parts is NULL");
  return bwd_sym_train_impl(d, z1, ld1, pool, ldp, lse_i, pool_lse, nullptr, 0, means, tick_counter, workspace, workspace_bytes, parts, stream);
}

// =====================================================================================
// Dot-product InfoNCE -- SimCLRLoss.loss, /root/reference/losses.py:177-202.
// Same tiled online-LSE skeleton with the pair term o*s (kind PK_DOT); the optional row
// L2-normalisation (losses.py:180-185) is a pre-pass into workspace plus its chain rule.
// =====================================================================================
namespace clica {
namespace lp {

// u = z / ||z||_2 ; inv[i] = 1/||z_i||
__global__ __launch_bounds__(THREADS) void rownorm_fwd_k(const float* __restrict__ z, int64_t ldz, int64_t rows, int n,
                                                        float* __restrict__ u, int64_t ldu, float* __restrict__ inv) {
  const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  if (i >= rows) return;
  float ss = 0.f;
  for (int k = 0; k < n; ++k) { const float v = z[i * ldz + k]; ss += v * v; }
  const float r = 1.f / sqrtf(ss);
  inv[i] = r;
  for (int k = 0; k < n; ++k) u[i * ldu + k] = z[i * ldz + k] * r;
}
// dz (+)= (du - u <du,u>) * inv
__global__ __launch_bounds__(THREADS) void rownorm_bwd_k(const float* __restrict__ u, const float* __restrict__ du, int64_t ldu,
                                                        const float* __restrict__ inv, int64_t rows, int n,
                                                        float* __restrict__ dz, int64_t lddz, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  if (i >= rows) return;
  float dot = 0.f;
  for (int k = 0; k < n; ++k) dot += du[i * ldu + k] * u[i * ldu + k];
  const float r = inv[i];
  for (int k = 0; k < n; ++k) {
    const float g = (du[i * ldu + k] - u[i * ldu + k] * dot) * r;
    float* dst = dz + i * lddz + k;
    *dst = accumulate ? (*dst + g) : g;
  }
}

// wide rows (n >= 64): one WAVE per row, coalesced, shuffle-reduced -- the thread-per-row kernels above walk n strided
// coordinates per thread (2.4 ms of a 2.5 ms SimCLRLoss step at n = 512, B = 1024)
constexpr int kWideRowN = 64;
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__global__ __launch_bounds__(256) void rownorm_fwd_wide_k(const float* __restrict__ z, int64_t ldz, int64_t rows, int n,
                                                         float* __restrict__ u, int64_t ldu, float* __restrict__ inv) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= rows) return;
  float ss = 0.f;
  for (int k = lane; k < n; k += 64) { const float v = z[i * ldz + k]; ss = fmaf(v, v, ss); }
  const float r = 1.f / sqrtf(wave_sum(ss));
  if (lane == 0) inv[i] = r;
  for (int k = lane; k < n; k += 64) u[i * ldu + k] = z[i * ldz + k] * r;
}
__global__ __launch_bounds__(256) void rownorm_bwd_wide_k(const float* __restrict__ u, const float* __restrict__ du, int64_t ldu,
                                                         const float* __restrict__ inv, int64_t rows, int n,
                                                         float* __restrict__ dz, int64_t lddz, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= rows) return;
  float dot = 0.f;
  for (int k = lane; k < n; k += 64) dot = fmaf(du[i * ldu + k], u[i * ldu + k], dot);
  dot = wave_sum(dot);
  const float r = inv[i];
  for (int k = lane; k < n; k += 64) {
    const float g = (du[i * ldu + k] - u[i * ldu + k] * dot) * r;
    float* dst = dz + i * lddz + k;
    *dst = accumulate ? (*dst + g) : g;
  }
}
__global__ __launch_bounds__(256) void rowdot_wide_k(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                                                    int64_t rows, int n, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= rows) return;
  float s = 0.f;
  for (int k = lane; k < n; k += 64) s = fmaf(a[i * lda + k], b[i * ldb + k], s);
  s = wave_sum(s);
  if (lane == 0) out[i] = s;
}
// dz1[i,:] = dpos_i z2[i,:],  dz2[i,:] = dpos_i z1[i,:]   (positive-pair gradient of the dot kind, losses.py:188)
__global__ __launch_bounds__(THREADS) void dpos_apply_k(const float* __restrict__ dpos, const float* __restrict__ a, int64_t lda,
                                                       const float* __restrict__ b, int64_t ldb, int64_t rows, int n,
                                                       float* __restrict__ dz1, int64_t ldd1, float* __restrict__ dz2, int64_t ldd2) {
  const int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  if (idx >= rows * n) return;
  const int64_t i = idx / n; const int k = (int)(idx - i * n);
  const float g = dpos[i];
  if (dz1) dz1[i * ldd1 + k] = g * b[i * ldb + k];
  if (dz2) dz2[i * ldd2 + k] = g * a[i * lda + k];
}

struct DotWs {
  float *u1, *u2, *u3, *i1, *i2, *i3, *du1, *du2, *du3, *pd, *dp;
  size_t bytes;
};
static DotWs carve_dot(void* ws, size_t off, int64_t B, int64_t B3, int n, bool normalize, bool bwd) {
  DotWs w{}; char* p = (char*)ws;
  auto take = [&](size_t count) { float* r = (float*)(p + off); off += align_up(count * sizeof(float), 256); return r; };
  if (normalize) {
    w.u1 = take((size_t)B * n); w.u2 = take((size_t)B * n); w.u3 = take((size_t)B3 * n);
    w.i1 = take(B); w.i2 = take(B); w.i3 = take(B3);
    if (bwd) { w.du1 = take((size_t)B * n); w.du2 = take((size_t)B * n); w.du3 = take((size_t)B3 * n); }
  }
  if (n >= kWideRowN) { w.pd = take(B); w.dp = take(B); }
  w.bytes = off; return w;
}
static int validate_dot(const clica_dot_loss_desc* d, const char* who) {
  CLICA_CHECK_ARG(d != nullptr, "%s: desc is NULL", who);
  CLICA_CHECK_ARG(d->B > 0 && d->B3 > 0, "%s: B=%lld B3=%lld must be positive", who, (long long)d->B, (long long)d->B3);
  CLICA_CHECK_ARG(d->n >= 1 && pad_dim(d->n) > 0, "%s: n=%d must be in 1..%d", who, d->n, kMaxWideN);
  CLICA_CHECK_ARG(d->tau > 0.f, "%s: tau=%g must be > 0", who, d->tau);
  return CLICA_OK;
}
static Params dot_params(const clica_dot_loss_desc* d) {
  Params q;
  q.p = 1.f; q.inv_p = 1.f; q.kscale = kLog2e / d->tau; q.sgn = 1.f; q.eps = 0.f; q.xs = 1.f; q.pow = 1; q.n = d->n;
  return q;
}
static void normalize_rows(const float* z, int64_t ld, int64_t rows, int n, float* u, float* inv, hipStream_t st) {
  if (n >= kWideRowN)
    hipLaunchKernelGGL(rownorm_fwd_wide_k, dim3((unsigned)ceil_div(rows, (int64_t)4)), dim3(256), 0, st, z, ld, rows, n, u, (int64_t)n, inv);
  else
    hipLaunchKernelGGL(rownorm_fwd_k, dim3((unsigned)ceil_div(rows, THREADS)), dim3(THREADS), 0, st, z, ld, rows, n, u, (int64_t)n, inv);
}
static void normalize_rows_bwd(const float* u, const float* du, const float* inv, int64_t rows, int n, float* dz, int64_t lddz, int acc,
                               hipStream_t st) {
  if (n >= kWideRowN)
    hipLaunchKernelGGL(rownorm_bwd_wide_k, dim3((unsigned)ceil_div(rows, (int64_t)4)), dim3(256), 0, st, u, du, (int64_t)n, inv, rows, n, dz, lddz, acc);
  else
    hipLaunchKernelGGL(rownorm_bwd_k, dim3((unsigned)ceil_div(rows, THREADS)), dim3(THREADS), 0, st, u, du, (int64_t)n, inv, rows, n, dz, lddz, acc);
}
static void row_dots(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows, int n, float* out, hipStream_t st) {
  hipLaunchKernelGGL(rowdot_wide_k, dim3((unsigned)ceil_div(rows, (int64_t)4)), dim3(256), 0, st, a, lda, b, ldb, rows, n, out);
}

// ---- SimCLRLoss on the matrix cores for wide rows (n >= 96) ------------------------------------------------------------
// losses.py:187 is a contraction, neg = z1 z3^T.  The pair sweep above evaluates it on the vector ALU (n FMAs per pair, fine at
// n = 10, 512 deep at the widths G19 covers); from n = 96 on the three products of the loss are GEMMs on the fp32 MFMA kernels
// of linear.hip, with the B x B3 logit matrix materialised ONCE per pass in the workspace (4 B B3 bytes against 2 B B3 n flops:
// 48 flops per byte at n = 96, compute-bound on the matrix pipe):
//   forward   S = U1 U3^T                  clica_linear_fwd   (M = B,  N = B3, K = n)
//             (max, sum 2^(. - max)) of S * log2(e)/tau per row and 2048-column chunk -> the partial format of fwd_partial_k,
//             finished by the SAME fwd_finalize_k (positive pair, loss, means)
//   backward  S again, then in place  W_ij = statC_i 2^(S_ij log2(e)/tau - statL_i)   (bwd_coef_k's row statistics)
//             dU1 += W U3                  clica_linear_dgrad (M = B,  N = B3, K = n)
//             dU3 (+)= W^T U1              clica_linear_wgrad (M = B,  N = B3, K = n)
// clica_set_tuning("dot_mfma", 0) keeps the pair sweep at every width (test hook: both paths are product paths, chosen by width).
constexpr int kDotMfmaMinN = 96;      // measured crossover (tools/simclr_bench.py, B = 4096): n = 64 472 vs 512 us, n = 128 925 vs 554 us
constexpr int DOT_CHUNK = 2048;            // columns per (max, sum) partial: 32 values per lane
static int& dot_mfma_switch() { static int on = 1; return on; }
void set_dot_mfma(int on) { dot_mfma_switch() = on ? 1 : 0; }       // clica_set_tuning (linear.hip)
static bool dot_mfma(const clica_dot_loss_desc* d) { return dot_mfma_switch() && d->n >= kDotMfmaMinN; }
struct DotMfmaWs { float* S; int64_t ldS; float2* part; int nchunk; float* T; void* wg; size_t wg_bytes; size_t bytes; };
static DotMfmaWs carve_dot_mfma(void* ws, size_t off, int64_t B, int64_t B3, int n, bool bwd) {
  DotMfmaWs w{}; char* p = (char*)ws;
  auto take = [&](size_t bytes) { void* r = p + off; off += align_up(bytes, 256); return r; };
  w.ldS = (B3 + 3) & ~(int64_t)3;
  w.nchunk = (int)ceil_div(B3, (int64_t)DOT_CHUNK);
  w.S = (float*)take((size_t)B * w.ldS * sizeof(float));
  w.part = (float2*)take((size_t)w.nchunk * B * sizeof(float2));
  if (bwd) {
    w.T = (float*)take((size_t)B * n * sizeof(float));
    size_t wb = 0;
    (void)clica_linear_wgrad_workspace_bytes(B, B3, n, &wb);
    w.wg_bytes = wb; w.wg = take(wb);
  }
  w.bytes = off; return w;
}
// one wave per (row, chunk): the row's logits of the chunk stay in registers between the max and the sum
__global__ __launch_bounds__(256) void dot_rows_partial_k(const float* __restrict__ S, int64_t ldS, int64_t rows, int64_t cols,
                                                         float kscale, float2* __restrict__ part) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= rows) return;
  const int64_t c0 = (int64_t)blockIdx.y * DOT_CHUNK;
  const float* row = S + i * ldS;
  float x[DOT_CHUNK / 64];
  float m = -1e30f;
#pragma unroll
  for (int u = 0; u < DOT_CHUNK / 256; ++u) {
    const int64_t c = c0 + u * 256 + lane * 4;
    float4 v = make_float4(-1e30f, -1e30f, -1e30f, -1e30f);
    if (c + 3 < cols) v = *reinterpret_cast<const float4*>(row + c);        // ldS and c are multiples of 4
    else { if (c < cols) v.x = row[c]; if (c + 1 < cols) v.y = row[c + 1]; if (c + 2 < cols) v.z = row[c + 2]; }
    const bool ok[4] = {c < cols, c + 1 < cols, c + 2 < cols, c + 3 < cols};
    const float t[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { x[u * 4 + e] = ok[e] ? t[e] * kscale : -1e30f; m = fmaxf(m, x[u * 4 + e]); }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < DOT_CHUNK / 64; ++u) s += fexp2(x[u] - m);          // 2^(-1e30 - m) = 0 for the padding
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) part[(int64_t)blockIdx.y * rows + i] = make_float2(m, s);
}
// S_ij <- c_i 2^(S_ij kscale - L_i) in place; c_i = statC[i] (backward) or 1 (forward-time row gradient)
__global__ __launch_bounds__(256) void dot_weights_k(float* __restrict__ S, int64_t ldS, int64_t rows, int64_t cols, float kscale,
                                                    const float* __restrict__ statL, const float* __restrict__ statC) {
  const int64_t i = blockIdx.x;
  const int64_t c = ((int64_t)blockIdx.y * 256 + threadIdx.x) * 4;
  if (c >= cols) return;
  const float L = statL[i], cf = statC ? statC[i] : 1.f;
  float* p = S + i * ldS + c;
  float4 v = *reinterpret_cast<float4*>(p);                                  // the row is padded to ldS: whole float4s are in range
  v.x = c < cols ? cf * fexp2(v.x * kscale - L) : 0.f;
  v.y = c + 1 < cols ? cf * fexp2(v.y * kscale - L) : 0.f;
  v.z = c + 2 < cols ? cf * fexp2(v.z * kscale - L) : 0.f;
  v.w = c + 3 < cols ? cf * fexp2(v.w * kscale - L) : 0.f;
  *reinterpret_cast<float4*>(p) = v;
}
__global__ __launch_bounds__(THREADS) void add_rows_k(const float* __restrict__ T, int64_t ldt, int64_t rows, int n,
                                                     float* __restrict__ out, int64_t ldo) {
  const int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  if (idx >= rows * n) return;
  const int64_t i = idx / n; const int k = (int)(idx - i * n);
  out[i * ldo + k] += T[i * ldt + k];
}
static int dot_logits(const float* u1, int64_t ld1, const float* u3, int64_t ld3, const DotMfmaWs& mw, int64_t B, int64_t B3, int n,
                      clica_stream_t stream) {
  return clica_linear_fwd(u1, ld1, u3, ld3, nullptr, mw.S, mw.ldS, B, B3, n, 0, 0.f, stream);
}
static void dot_weights(const DotMfmaWs& mw, int64_t B, int64_t B3, float kscale, const float* statL, const float* statC, hipStream_t st) {
  hipLaunchKernelGGL(dot_weights_k, dim3((unsigned)B, (unsigned)ceil_div(B3, (int64_t)1024)), dim3(256), 0, st,
                     mw.S, mw.ldS, B, B3, kscale, statL, statC);
}

}  // namespace lp
}  // namespace clica

extern "C" int clica_dot_loss_workspace_bytes(const clica_dot_loss_desc* d, size_t* fwd_bytes, size_t* bwd_bytes) {
  int rc = validate_dot(d, "clica_dot_loss_workspace_bytes");
  if (rc) return rc;
  Plan PF = make_plan(d->B, d->B3, d->n, false), PR = make_plan(d->B, d->B3, d->n, true), PC = make_plan(d->B3, d->B, d->n, true);
  if (fwd_bytes) *fwd_bytes = carve_dot(nullptr, fwd_carve_max(d->B, d->B3, d->n), d->B, d->B3, d->n, d->normalize, false).bytes;
  (void)PF;
  if (bwd_bytes) *bwd_bytes = carve_dot(nullptr, carve_bwd(nullptr, PR, PC, d->B, d->B3).bytes, d->B, d->B3, d->n, d->normalize, true).bytes;
  if (dot_mfma(d)) {      // logit matrix (+ backward: row-gradient staging and the split-K slabs of the column GEMM) behind everything else
    if (fwd_bytes) *fwd_bytes = carve_dot_mfma(nullptr, *fwd_bytes, d->B, d->B3, d->n, false).bytes;
    if (bwd_bytes) *bwd_bytes = carve_dot_mfma(nullptr, *bwd_bytes, d->B, d->B3, d->n, true).bytes;
  }
  return CLICA_OK;
}

extern "C" int clica_dot_loss_fwd(const clica_dot_loss_desc* d,
                                  const float* z1, int64_t ld1, const float* z2, int64_t ld2,
                                  const float* z3, int64_t ld3,
                                  float* loss_i, float* pos_i, float* lse_i, float* means,
                                  float* rowgrad, int64_t ldrg,
                                  void* workspace, size_t workspace_bytes, clica_stream_t stream) {
  int rc = validate_dot(d, "clica_dot_loss_fwd");
  if (rc) return rc;
  CLICA_CHECK_ARG(z1 && z2 && z3 && loss_i && pos_i && lse_i && means && workspace, "clica_dot_loss_fwd: NULL pointer");
  CLICA_CHECK_ARG(ld1 >= d->n && ld2 >= d->n && ld3 >= d->n, "clica_dot_loss_fwd: leading dimension < n");
  CLICA_CHECK_ARG(!rowgrad || ldrg >= d->n, "clica_dot_loss_fwd: rowgrad leading dimension < n");
  Plan P = make_plan(d->B, d->B3, d->n, rowgrad != nullptr);
  // the U/inv buffers sit behind the LARGER forward carve so fwd and bwd agree on their offsets
  FwdWs w = carve_fwd(workspace, P, d->B, rowgrad != nullptr);
  const size_t dot_off = fwd_carve_max(d->B, d->B3, d->n);
  DotWs dw = carve_dot(workspace, dot_off, d->B, d->B3, d->n, d->normalize, false);
  if (dw.bytes > workspace_bytes) { set_error("clica_dot_loss_fwd: workspace %zu < %zu", workspace_bytes, dw.bytes); return CLICA_E_WORKSPACE; }
  hipStream_t st = as_stream(stream);
  Params q = dot_params(d);
  if (d->normalize) {
    normalize_rows(z1, ld1, d->B, d->n, dw.u1, dw.i1, st);
    normalize_rows(z2, ld2, d->B, d->n, dw.u2, dw.i2, st);
    normalize_rows(z3, ld3, d->B3, d->n, dw.u3, dw.i3, st);
    z1 = dw.u1; z2 = dw.u2; z3 = dw.u3; ld1 = ld2 = ld3 = d->n;
  }
  if (dw.pd) { row_dots(z1, ld1, z2, ld2, d->B, d->n, dw.pd, st); q.posdot = dw.pd; }
  Means M{w.blocksums};
  const int nfin = (int)ceil_div(d->B, FIN_ROWS);
  if (dot_mfma(d)) {
    const DotMfmaWs mw = carve_dot_mfma(workspace, dw.bytes, d->B, d->B3, d->n, false);
    if (mw.bytes > workspace_bytes) { set_error("clica_dot_loss_fwd: workspace %zu < %zu", workspace_bytes, mw.bytes); return CLICA_E_WORKSPACE; }
    rc = dot_logits(z1, ld1, z3, ld3, mw, d->B, d->B3, d->n, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(dot_rows_partial_k, dim3((unsigned)ceil_div(d->B, (int64_t)4), (unsigned)mw.nchunk), dim3(256), 0, st,
                       (const float*)mw.S, mw.ldS, d->B, d->B3, q.kscale, mw.part);
    hipLaunchKernelGGL(fwd_finalize_k, dim3((unsigned)nfin), dim3(THREADS), 0, st,
                       (const float2*)mw.part, mw.nchunk, d->B, z1, ld1, z2, ld2, q, d->tau, d->alpha,
                       1, 0, 1, 0.f, loss_i, pos_i, lse_i, M, TrainOut{}, FeatOut{}, Gate{});
    hipLaunchKernelGGL(means_k, dim3(1), dim3(64), 0, st, (const float*)w.blocksums, nfin, 1.f / (float)d->B, means);
    if (rowgrad) {          // softmax-weighted sum of the z3 rows: one more GEMM on the weights (coefficient 1)
      dot_weights(mw, d->B, d->B3, q.kscale, lse_i, nullptr, st);
      rc = clica_linear_dgrad(mw.S, mw.ldS, z3, ld3, nullptr, 0, 0.f, rowgrad, ldrg, d->B, d->B3, d->n, stream);
      if (rc) return rc;
    }
    return launch_status("clica_dot_loss_fwd(mfma)");
  }
  launch_fwd_partial(P, PK_DOT, z1, ld1, d->B, z3, ld3, d->B3, q, w.part, w.part_g, st);
  hipLaunchKernelGGL(fwd_finalize_k, dim3((unsigned)nfin), dim3(THREADS), 0, st,
                     (const float2*)w.part, P.nsplit, d->B, z1, ld1, z2, ld2, q, d->tau, d->alpha,
                     1, 0, 1, 0.f, loss_i, pos_i, lse_i, M, TrainOut{}, FeatOut{}, Gate{});
  hipLaunchKernelGGL(means_k, dim3(1), dim3(64), 0, st, (const float*)w.blocksums, nfin, 1.f / (float)d->B, means);
  if (rowgrad)   // gradient w.r.t. the (normalised, if requested) rows
    hipLaunchKernelGGL(rowgrad_combine_k, dim3((unsigned)ceil_div(d->B * (P.np / 4), THREADS)), dim3(THREADS), 0, st,
                       (const float2*)w.part, (const float*)w.part_g, P.nsplit, d->B, P.np, d->n, (const float*)lse_i, rowgrad, ldrg);
  return launch_status("clica_dot_loss_fwd");
}

extern "C" int clica_dot_loss_bwd(const clica_dot_loss_desc* d,
                                  const float* z1, int64_t ld1, const float* z2, int64_t ld2,
                                  const float* z3, int64_t ld3, const float* lse_i,
                                  const float* rowgrad, int64_t ldrg,
                                  const float* g_mean, const float* g_item, const float* g_pos, const float* g_neg,
                                  float* dz1, int64_t ldd1, float* dz2, int64_t ldd2,
                                  float* dz3, int64_t ldd3, int32_t accumulate_dz3,
                                  void* workspace, size_t workspace_bytes, clica_stream_t stream) {
  int rc = validate_dot(d, "clica_dot_loss_bwd");
  if (rc) return rc;
  CLICA_CHECK_ARG(z1 && z2 && z3 && lse_i && workspace, "clica_dot_loss_bwd: NULL pointer");
  const int64_t B = d->B, B3 = d->B3; const int n = d->n;
  Plan PR = make_plan(B, B3, n, true), PC = make_plan(B3, B, n, true);
  BwdWs w = carve_bwd(workspace, PR, PC, B, B3);
  DotWs dw = carve_dot(workspace, w.bytes, B, B3, n, d->normalize, true);
  if (dw.bytes > workspace_bytes) { set_error("clica_dot_loss_bwd: workspace %zu < %zu", workspace_bytes, dw.bytes); return CLICA_E_WORKSPACE; }
  hipStream_t st = as_stream(stream);
  Params q = dot_params(d);
  // with normalisation the pair gradients land in du* and go through the chain rule afterwards
  float *o1 = dz1, *o2 = dz2, *o3 = dz3; int64_t lo1 = ldd1, lo2 = ldd2, lo3 = ldd3; int acc3 = accumulate_dz3 ? 1 : 0;
  if (d->normalize) {
    normalize_rows(z1, ld1, B, n, dw.u1, dw.i1, st);
    normalize_rows(z2, ld2, B, n, dw.u2, dw.i2, st);
    normalize_rows(z3, ld3, B3, n, dw.u3, dw.i3, st);
    z1 = dw.u1; z2 = dw.u2; z3 = dw.u3; ld1 = ld2 = ld3 = n;
    o1 = dz1 ? dw.du1 : nullptr; o2 = dz2 ? dw.du2 : nullptr; o3 = dz3 ? dw.du3 : nullptr; lo1 = lo2 = lo3 = n; acc3 = 0;
  }
  if (dw.pd) { row_dots(z1, ld1, z2, ld2, B, n, dw.pd, st); q.posdot = dw.pd; q.dpos = dw.dp; }
  hipLaunchKernelGGL(bwd_coef_k, dim3((unsigned)ceil_div(B, THREADS)), dim3(THREADS), 0, st,
                     B, z1, ld1, z2, ld2, q, d->tau, d->alpha, 1, 0, 1, lse_i,
                     g_mean, g_item, g_pos, g_neg, w.statL, w.statC, o1, lo1, o2, lo2);
  if (dw.pd && (o1 || o2))
    hipLaunchKernelGGL(dpos_apply_k, dim3((unsigned)ceil_div(B * n, THREADS)), dim3(THREADS), 0, st,
                       (const float*)dw.dp, z1, ld1, z2, ld2, B, n, o1, lo1, o2, lo2);
  const bool mfma = dot_mfma(d) && ((o1 && !rowgrad) || o3);
  DotMfmaWs mw{};
  if (mfma) {
    mw = carve_dot_mfma(workspace, dw.bytes, B, B3, n, true);
    if (mw.bytes > workspace_bytes) { set_error("clica_dot_loss_bwd: workspace %zu < %zu", workspace_bytes, mw.bytes); return CLICA_E_WORKSPACE; }
    rc = dot_logits(z1, ld1, z3, ld3, mw, B, B3, n, stream);
    if (rc) return rc;
    dot_weights(mw, B, B3, q.kscale, w.statL, w.statC, st);
  }
  if (o1 && rowgrad) {
    hipLaunchKernelGGL(rowgrad_apply_k, dim3((unsigned)ceil_div(B * n, THREADS)), dim3(THREADS), 0, st,
                       rowgrad, ldrg, (const float*)w.statC, B, n, o1, lo1, 1);
  } else if (o1 && mfma) {
    rc = clica_linear_dgrad(mw.S, mw.ldS, z3, ld3, nullptr, 0, 0.f, mw.T, n, B, B3, n, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(add_rows_k, dim3((unsigned)ceil_div(B * n, THREADS)), dim3(THREADS), 0, st, (const float*)mw.T, (int64_t)n, B, n, o1, lo1);
  } else if (o1) {
    launch_bwd_pairs(true, PR, PK_DOT, z1, ld1, B, z3, ld3, B3, q, w.statL, w.statC, w.partR, st);
    hipLaunchKernelGGL(bwd_reduce_k, dim3((unsigned)ceil_div(B * PR.np, THREADS)), dim3(THREADS), 0, st,
                       (const float*)w.partR, PR.nsplit, B, PR.np, n, o1, lo1, 1, MeansJob{nullptr, 0, 0.f, nullptr, -1, nullptr}, Gate{});
  }
  if (o3 && mfma) {
    rc = clica_linear_wgrad(mw.S, mw.ldS, z1, ld1, o3, lo3, nullptr, B, B3, n, acc3, mw.wg, mw.wg_bytes, stream);
    if (rc) return rc;
  } else if (o3) {
    launch_bwd_pairs(false, PC, PK_DOT, z3, ld3, B3, z1, ld1, B, q, w.statL, w.statC, w.partC, st);
    hipLaunchKernelGGL(bwd_reduce_k, dim3((unsigned)ceil_div(B3 * PC.np, THREADS)), dim3(THREADS), 0, st,
                       (const float*)w.partC, PC.nsplit, B3, PC.np, n, o3, lo3, acc3, MeansJob{nullptr, 0, 0.f, nullptr, -1, nullptr}, Gate{});
  }
  if (d->normalize) {
    if (dz1) normalize_rows_bwd(dw.u1, dw.du1, dw.i1, B, n, dz1, ldd1, 0, st);
    if (dz2) normalize_rows_bwd(dw.u2, dw.du2, dw.i2, B, n, dz2, ldd2, 0, st);
    if (dz3) normalize_rows_bwd(dw.u3, dw.du3, dw.i3, B3, n, dz3, ldd3, accumulate_dz3 ? 1 : 0, st);
  }
  return launch_status("clica_dot_loss_bwd");
}
