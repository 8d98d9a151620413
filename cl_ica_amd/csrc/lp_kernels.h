// Lp InfoNCE loss (LpSimCLRLoss.loss, /root/reference/losses.py:430-477) for gfx950.
//
// The reference materialises the (B, B3, n) broadcast difference and the (B, B3) distance
// matrix (1.5 GB + 151 MB at B = 6144, n = 10).  Here the pair space is tiled flash-style:
//
//   forward : grid (owner tiles) x (stream splits).  A thread OWNS R rows ("owners") whose n
//             coordinates live in registers; the other operand ("stream") is staged through
//             LDS 64 rows at a time and read with wave-uniform (broadcast) ds_read_b128.
//             Each thread keeps a running (max, sum) in the log2 domain per owner; per-split
//             partials go to a small workspace and `finalize` merges them, adds the positive
//             pair, and emits loss_i / pos_i / lse_i and the three means (deterministic
//             last-block reduction, no float atomics).
//   backward: distances are recomputed.  d/dz1 is a row reduction (owners = z1 rows, softmax
//             statistics per owner); d/dz3 is a column reduction over the ROW-normalised
//             weights (owners = z3 rows, statistics per stream row).  Same kernel, roles
//             swapped.  No B x B3 storage, no atomics.
//
// This is an all-pairs VALU/transcendental-bound kernel (not a contraction; MFMA does not
// apply for p != 2 and the ||a||^2+||b||^2-2ab expansion loses 3 digits at the exact-zero
// pairs that z3 = roll(z1) guarantees).  HBM traffic is ~(2B + B3) n 4 bytes.
#pragma once
#include "common.h"
#include <math.h>

namespace clica {
namespace lp {

constexpr int THREADS = 256;
constexpr int TS = 64;  // stream rows per LDS tile
constexpr int JB = 4;   // stream rows processed together (independent FMA chains)

struct Params {
  float p;        // exponent
  float inv_p;
  float kscale;   // log2(e) / tau
  float sgn;      // e = sgn * (owner - stream) + eps      (p < 1 branch: sgn = -1, eps = 1e-12)
  float eps;
  float xs;       // logit = xs * neg * kscale : -1 for Lp distances, +1 for the dot-product kind
  int pow;        // 1: use sum |e|^p ; 0: its 1/p-th root
  int n;          // true embedding dim (<= NP)
};

// ---- per-coordinate pair term and its owner-derivative ----------------------------------------
// PK: 1,2,3 = integer Lp fast paths, 0 = generic p > 0 (incl. the p < 1 eps branch),
//     4 = dot product (SimCLRLoss, losses.py:187): term = o*s, d/do = s
constexpr int PK_DOT = 4;
template <int PK>
__device__ __forceinline__ float pair_e(float o, float s, const Params& q) {
  if constexpr (PK == 0) return q.sgn * (o - s) + q.eps;
  return o - s;
}
template <int PK>
__device__ __forceinline__ float term(float o, float s, const Params& q) {
  if constexpr (PK == PK_DOT) return o * s;
  const float e = pair_e<PK>(o, s, q);
  if constexpr (PK == 1) return fabsf(e);
  if constexpr (PK == 2) return e * e;
  if constexpr (PK == 3) return fabsf(e) * e * e;
  const float a = fabsf(e);
  return a > 0.f ? exp2f(q.p * log2f(a)) : 0.f;
}
// (1/p) d term / d owner -- the factor p is folded into the pair coefficient (droot_of), the
// sign of the generic branch into the coefficient as well.  Zero at e == 0 (torch.norm's backward
// masks the zero-norm entries; sign(0) = 0).
template <int PK>
__device__ __forceinline__ float dterm(float o, float s, const Params& q) {
  if constexpr (PK == PK_DOT) return s;
  const float e = pair_e<PK>(o, s, q);
  if constexpr (PK == 1) return (e > 0.f ? 1.f : 0.f) - (e < 0.f ? 1.f : 0.f);
  if constexpr (PK == 2) return e;
  if constexpr (PK == 3) return e * fabsf(e);
  const float a = fabsf(e);
  const float v = a > 0.f ? exp2f((q.p - 1.f) * log2f(a)) : 0.f;
  return e < 0.f ? -v : v;
}
// neg value from the sum of powers, and d neg / d sum (times p, see dterm)
__device__ __forceinline__ float root_of(float s, const Params& q) {
  if (q.pow) return s;
  if (q.p == 2.f) return sqrtf(s);
  if (q.p == 1.f) return s;
  return s > 0.f ? exp2f(q.inv_p * log2f(s)) : 0.f;
}
__device__ __forceinline__ float droot_of(float s, const Params& q) {  // p * d root / d s
  if (q.pow) return q.p;
  if (q.p == 1.f) return 1.f;
  return s > 0.f ? exp2f((q.inv_p - 1.f) * log2f(s)) : 0.f;
}

// ---- staging -----------------------------------------------------------------------------
template <int NP>
__device__ __forceinline__ void stage_tile(float* tile, const float* __restrict__ str, int64_t lds,
                                           int64_t j0, int cnt, int n) {
  for (int idx = threadIdx.x; idx < TS * NP; idx += THREADS) {
    int row = idx / NP, k = idx - row * NP;
    float v = 0.f;
    if (row < cnt && k < n) v = str[(j0 + row) * lds + k];
    tile[idx] = v;
  }
}

template <int NP, int R>
__device__ __forceinline__ void load_owners(float (&o)[R][NP], const float* __restrict__ own, int64_t ldo,
                                            int64_t own0, int64_t n_own, int n) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int64_t i = own0 + (int64_t)r * THREADS + threadIdx.x;
    bool ok = i < n_own;
#pragma unroll
    for (int k = 0; k < NP; ++k) o[r][k] = (ok && k < n) ? own[i * ldo + k] : 0.f;
  }
}

// sum_k |e_k|^p for one owner against JB stream rows of the LDS tile
template <int NP, int PK>
__device__ __forceinline__ void dist_group(const float (&o)[NP], const float* tile, int jj, const Params& q,
                                           float (&acc)[JB]) {
#pragma unroll
  for (int c = 0; c < JB; ++c) acc[c] = 0.f;
#pragma unroll
  for (int k4 = 0; k4 < NP / 4; ++k4) {
#pragma unroll
    for (int c = 0; c < JB; ++c) {
      const float4 sv = *reinterpret_cast<const float4*>(&tile[(jj + c) * NP + 4 * k4]);
      const float s4[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (PK == 0 && 4 * k4 + u >= q.n) continue;  // zero padding is not neutral with eps
        acc[c] += term<PK>(o[4 * k4 + u], s4[u], q);
      }
    }
  }
}

// ---- forward: per-split (max, sum) partials in the log2 domain -----------------------------
template <int NP, int PK, int R>
__global__ __launch_bounds__(THREADS) void fwd_partial_k(
    const float* __restrict__ own, int64_t ldo, int64_t n_own,
    const float* __restrict__ str, int64_t lds, int64_t n_str,
    Params q, float2* __restrict__ part, int chunk) {
  __shared__ __attribute__((aligned(16))) float tile[TS * NP];
  const int64_t own0 = (int64_t)blockIdx.x * (THREADS * R);
  float o[R][NP];
  load_owners<NP, R>(o, own, ldo, own0, n_own, q.n);
  float m[R], s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { m[r] = -INFINITY; s[r] = 0.f; }

  const int64_t jb = (int64_t)blockIdx.y * chunk;
  const int64_t je = min(n_str, jb + (int64_t)chunk);
  for (int64_t j0 = jb; j0 < je; j0 += TS) {
    const int cnt = (int)min((int64_t)TS, je - j0);
    __syncthreads();
    stage_tile<NP>(tile, str, lds, j0, cnt, q.n);
    __syncthreads();
    for (int jj = 0; jj < cnt; jj += JB) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float acc[JB];
        dist_group<NP, PK>(o[r], tile, jj, q, acc);
        float x[JB];
        float bm = -INFINITY;
#pragma unroll
        for (int c = 0; c < JB; ++c) {
          x[c] = (jj + c < cnt) ? q.xs * root_of(acc[c], q) * q.kscale : -INFINITY;
          bm = fmaxf(bm, x[c]);
        }
        // clamp keeps (-inf) - (-inf) out of the exponent when nothing valid was seen yet
        const float mn = fmaxf(fmaxf(m[r], bm), -1e30f);
        float add = 0.f;
#pragma unroll
        for (int c = 0; c < JB; ++c) add += exp2f(x[c] - mn);
        s[r] = s[r] * exp2f(m[r] - mn) + add;
        m[r] = mn;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int64_t i = own0 + (int64_t)r * THREADS + threadIdx.x;
    if (i < n_own) part[(int64_t)blockIdx.y * n_own + i] = make_float2(m[r], s[r]);
  }
}

// ---- backward ------------------------------------------------------------------------------
// Owner-gradient partials.  OWNER_STATS: softmax statistics belong to the owner rows (d/dz1);
// otherwise to the stream rows (d/dz3: column reduction over row-normalised weights).
template <int NP, int PK, int R, bool OWNER_STATS>
__global__ __launch_bounds__(THREADS) void bwd_pairs_k(
    const float* __restrict__ own, int64_t ldo, int64_t n_own,
    const float* __restrict__ str, int64_t lds, int64_t n_str,
    Params q, const float* __restrict__ statL, const float* __restrict__ statC,
    float* __restrict__ part, int chunk) {
  __shared__ __attribute__((aligned(16))) float tile[TS * NP];
  __shared__ float tL[TS], tC[TS];
  const int64_t own0 = (int64_t)blockIdx.x * (THREADS * R);
  float o[R][NP], g[R][NP];
  load_owners<NP, R>(o, own, ldo, own0, n_own, q.n);
  float oL[R], oC[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int64_t i = own0 + (int64_t)r * THREADS + threadIdx.x;
    oL[r] = 0.f; oC[r] = 0.f;
    if (OWNER_STATS && i < n_own) { oL[r] = statL[i]; oC[r] = statC[i]; }
#pragma unroll
    for (int k = 0; k < NP; ++k) g[r][k] = 0.f;
  }
  const int64_t jb = (int64_t)blockIdx.y * chunk;
  const int64_t je = min(n_str, jb + (int64_t)chunk);
  for (int64_t j0 = jb; j0 < je; j0 += TS) {
    const int cnt = (int)min((int64_t)TS, je - j0);
    __syncthreads();
    stage_tile<NP>(tile, str, lds, j0, cnt, q.n);
    if (!OWNER_STATS && threadIdx.x < TS) {
      const bool ok = threadIdx.x < cnt;
      tL[threadIdx.x] = ok ? statL[j0 + threadIdx.x] : 0.f;
      tC[threadIdx.x] = ok ? statC[j0 + threadIdx.x] : 0.f;
    }
    __syncthreads();
    for (int jj = 0; jj < cnt; jj += JB) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float acc[JB];
        dist_group<NP, PK>(o[r], tile, jj, q, acc);
        float coef[JB];
#pragma unroll
        for (int c = 0; c < JB; ++c) {
          const float L = OWNER_STATS ? oL[r] : tL[jj + c];
          const float C = OWNER_STATS ? oC[r] : tC[jj + c];
          const float w = exp2f(q.xs * root_of(acc[c], q) * q.kscale - L);
          float cf = C * w * droot_of(acc[c], q);
          if (PK == 0) cf *= q.sgn;
          coef[c] = (jj + c < cnt) ? cf : 0.f;
        }
#pragma unroll
        for (int k4 = 0; k4 < NP / 4; ++k4) {
#pragma unroll
          for (int c = 0; c < JB; ++c) {
            const float4 sv = *reinterpret_cast<const float4*>(&tile[(jj + c) * NP + 4 * k4]);
            const float s4[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              if (PK == 0 && 4 * k4 + u >= q.n) continue;
              g[r][4 * k4 + u] += coef[c] * dterm<PK>(o[r][4 * k4 + u], s4[u], q);
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int64_t i = own0 + (int64_t)r * THREADS + threadIdx.x;
    if (i < n_own) {
      float4* dst = reinterpret_cast<float4*>(part + ((int64_t)blockIdx.y * n_own + i) * NP);
#pragma unroll
      for (int k4 = 0; k4 < NP / 4; ++k4)
        dst[k4] = make_float4(g[r][4 * k4], g[r][4 * k4 + 1], g[r][4 * k4 + 2], g[r][4 * k4 + 3]);
    }
  }
}

// ---- host-side planning -----------------------------------------------------------------------
inline int pad_dim(int n) {
  static const int dims[] = {4, 8, 12, 16, 24, 32, 40, 64};
  for (int d : dims) if (n <= d) return d;
  return -1;
}
constexpr int owners_fwd(int np) { return np <= 24 ? 2 : 1; }
constexpr int owners_bwd(int np) { return np <= 16 ? 2 : 1; }

struct Plan {
  int np, R;
  int64_t tiles;  // owner tiles
  int nsplit, chunk;
};
inline Plan make_plan(int64_t n_own, int64_t n_str, int n, bool bwd) {
  Plan P;
  P.np = pad_dim(n);
  P.R = bwd ? owners_bwd(P.np) : owners_fwd(P.np);
  P.tiles = ceil_div(n_own, (int64_t)THREADS * P.R);
  // aim for ~4 workgroups per CU; each split covers a whole number of LDS tiles
  int64_t want = ceil_div((int64_t)kNumCU * 4, P.tiles);
  int64_t max_split = ceil_div(n_str, (int64_t)TS);
  int64_t ns = want < 1 ? 1 : (want > max_split ? max_split : want);
  if (ns < 1) ns = 1;
  int64_t chunk = ceil_div(ceil_div(n_str, ns), (int64_t)TS) * TS;
  if (chunk < TS) chunk = TS;
  P.chunk = (int)chunk;
  P.nsplit = (int)(n_str > 0 ? ceil_div(n_str, chunk) : 1);
  return P;
}


// per-exponent-kind launchers, one translation unit each (lp_loss_pk.hip with -DCLICA_PK=k)
#define CLICA_LP_DECLARE(PKV)                                                                          \
  void launch_fwd_partial_pk##PKV(const Plan& P, const float* own, int64_t ldo, int64_t n_own,        \
                                  const float* str, int64_t lds, int64_t n_str, const Params& q,      \
                                  float2* part, hipStream_t st);                                      \
  void launch_bwd_pairs_pk##PKV(const Plan& P, bool owner_stats, const float* own, int64_t ldo,       \
                                int64_t n_own, const float* str, int64_t lds, int64_t n_str,          \
                                const Params& q, const float* statL, const float* statC, float* part, \
                                hipStream_t st);
CLICA_LP_DECLARE(0) CLICA_LP_DECLARE(1) CLICA_LP_DECLARE(2) CLICA_LP_DECLARE(3) CLICA_LP_DECLARE(4)

}  // namespace lp
}  // namespace clica
