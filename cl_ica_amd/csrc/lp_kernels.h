// Lp InfoNCE loss (LpSimCLRLoss.loss, /root/reference/losses.py:430-477) for gfx950.
//
// The reference materialises the (B, B3, n) broadcast difference and the (B, B3) distance
// matrix (1.5 GB + 151 MB at B = 6144, n = 10).  Here the pair space is tiled flash-style:
//
//   forward : grid (owner tiles) x (stream splits).  A workgroup owns 32 R rows ("owners") whose n coordinates
//             live in registers; the other operand ("stream") is staged through LDS TS rows at a time.  The
//             workgroup's stream chunk is cut into PARTS = 8 partitions on chip: wave w, half-wave h works on
//             rows [q TS/8, (q+1) TS/8) of every tile (q = 2w + h) -- lanes l and l + 32 hold the SAME owner and
//             read DIFFERENT stream rows (one ds_read_b128 serves both halves: two broadcast addresses cost the
//             same four LDS cycles as one).  Each lane keeps a running (max, sum) in the log2 domain per owner;
//             the eight partitions are merged with one shuffle + one pass through LDS, so only
//             (stream splits) = ~8 partials per row reach HBM (the first version kept 512 owners per workgroup
//             and needed ~85 HBM-level splits to fill the chip: 15-39x the algorithmic bytes in partial traffic
//             and two latency-bound merge kernels).  `finalize` merges the splits, adds the positive pair, and
//             emits loss_i / pos_i / lse_i and the three means (deterministic, no float atomics).
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
#include <stdlib.h>
#include <type_traits>

namespace clica {
namespace lp {

constexpr int THREADS = 256;
constexpr int WAVES = THREADS / 64;
constexpr int HALF = 32;            // owner rows per wave: lanes l and l + 32 share an owner and split the stream
constexpr int PARTS = 2 * WAVES;    // stream partitions inside a workgroup (merged on chip, never through HBM)
constexpr int JB = 4;               // stream rows processed together (independent FMA chains)
#ifndef CLICA_LP_JBW
#define CLICA_LP_JBW 2      // measured (tools/loss_train_probe.py): 2 beats 4 by 3 % (n = 10) to 7 % (n = 40), 1 and 8 are slower
#endif
// ... and for the forward sweep: four up to the 16-wide layout, two above (measured at n = 10: 44 / 47 / 76 us for 4 / 2 / 8 rows,
// at n = 40: 180 / 168 us for 4 / 2)
constexpr int jb_fwd(int np) { return np <= 16 ? 4 : 2; }
constexpr int JBW = CLICA_LP_JBW;   // the same for the backward sweep (RPP = 16 rows per partition and tile: 2, 4, 8 or 16)
// stream rows per LDS tile: every partition gets TS / PARTS of them (32 / 16 / 8 rows)
#ifndef CLICA_LP_TS16
#define CLICA_LP_TS16 128      // measured (tools/loss_train_probe.py): 128 beats 256 by 14 % on the backward sweep (registers: the
#endif                        // staging prefetch holds TS NP / 256 values per thread across the tile's arithmetic)
#ifndef CLICA_LP_TS40
#define CLICA_LP_TS40 128
#endif
constexpr int tile_rows(int np) { return np <= 16 ? CLICA_LP_TS16 : (np <= 40 ? CLICA_LP_TS40 : 64); }

struct Params {
  float p;        // exponent
  float inv_p;
  float kscale;   // log2(e) / tau
  float sgn;      // e = sgn * (owner - stream) + eps      (p < 1 branch: sgn = -1, eps = 1e-12)
  float eps;
  float xs;       // logit = xs * neg * kscale : -1 for Lp distances, +1 for the dot-product kind
  int pow;        // 1: use sum |e|^p ; 0: its 1/p-th root
  int n;          // true embedding dim (<= NP)
  // training entry points only (clica_lp_loss_fwd_train / bwd_sym_train: the pool CONTAINS the owner rows, integer p, pow):
  //   bit 0: forward without the running maximum -- logits of an Lp distance are <= 0 and the owner's own pool row gives exactly 0, so
  //          sum_j 2^x_ij >= 1 needs no rescaling (per pair: multiply, exp, add instead of max / subtract / exp / rescale)
  //   bit 1: backward coefficient with ONE exponential, 2^x (u_i + u_j) with u = C 2^-L per row (L = log2 of a sum >= 1, so 2^-L is in
  //          (0, 1]) instead of C_i 2^(x - L_i) + C_j 2^(x - L_j)
  int train = 0;
  // coordinate scale applied to owner AND stream rows as they are loaded (the same fp32 multiply on both sides: identical rows stay
  // identical, so exact-zero pairs stay exact).  1 outside the training sweeps; there pre = (log2(e) / tau)^(1/p), which makes the
  // scaled distance sum the logit itself (x = -acc: no per-pair multiply), see fwd_partial_k<ZMAX> / bwd_pairs_k<FOLD>.  NOT for p = 1:
  // its gradient is sign(o - s), and a scale that is not a power of two can round two coordinates a few ulps apart onto each other
  // (measured: 1.8e-4 of the gradient scale in the n = 40 full-size test); p = 1 keeps unscaled rows and the multiply
  float pre = 1.f;
  float gfold = 1.f;   // FOLD sweeps: p / pre^(p-1), the factor between sum_j w_ij term'(d_scaled) and the gradient in unscaled coordinates
  // dot kind, wide rows (n >= 64) only: <z1_i, z2_i> computed beforehand by a wave-per-row kernel (the finishing kernels run one
  // THREAD per row and would walk 512 strided coordinates each), and where the coefficient step leaves d loss / d pos_i for
  // an element-wise kernel instead of writing the two gradient rows itself
  const float* posdot = nullptr;
  float* dpos = nullptr;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// raw v_exp_f32 / v_log_f32 (base 2).  exp2f()/log2f() wrap these in denormal-range fix-ups
// (compare + select + ldexp per call: 6 instructions instead of 1); a softmax term below 2^-126
// relative to the row maximum is zero for our purposes.
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float flog2(float x) { return __builtin_amdgcn_logf(x); }

// ---- per-coordinate pair term and its owner-derivative ----------------------------------------
// PK: 1,2,3 = integer Lp fast paths, 0 = generic p > 0 (incl. the p < 1 eps branch),
//     4 = dot product (SimCLRLoss, losses.py:187): term = o*s, d/do = s
constexpr int PK_DOT = 4;

// acc + |d| as ONE instruction (the source-modifier form of v_add_f32).  Written out because the compiler prefers to clear the sign
// bits with v_and_b32 (one slot per element) and add the pair with v_pk_add_f32 (half a slot): 1.5 slots per element instead of
// 1 -- 40 of the ~175 issue slots per pair of the n = 40, p = 1 forward sweep.
__device__ __forceinline__ float add_abs(float acc, float d) {
  float r;
  asm("v_add_f32_e64 %0, %1, |%2|" : "=v"(r) : "v"(acc), "v"(d));
  return r;
}

// acc += term(o, s) for two adjacent coordinates (k, k+1); packed fp32 math where the ISA has it
// (v_pk_add_f32 / v_pk_fma_f32: two lanes of work per issue slot)
template <int PK>
__device__ __forceinline__ void accum2(f32x2& acc, f32x2 o, f32x2 s, const Params& q, int k) {
  if constexpr (PK == PK_DOT) {
    acc = __builtin_elementwise_fma(o, s, acc);
  } else if constexpr (PK == 2) {
    const f32x2 d = o - s;
    acc = __builtin_elementwise_fma(d, d, acc);
  } else if constexpr (PK == 1) {
    const f32x2 d = o - s;
    acc.x = add_abs(acc.x, d.x); acc.y = add_abs(acc.y, d.y);
  } else if constexpr (PK == 3) {
    const f32x2 d = o - s, t = d * d;
    acc.x = fmaf(fabsf(d.x), t.x, acc.x); acc.y = fmaf(fabsf(d.y), t.y, acc.y);
  } else {   // generic exponent; zero padding is NOT neutral with the eps of the p < 1 branch
    if (k < q.n) { const float a = fabsf(q.sgn * (o.x - s.x) + q.eps); acc.x += a > 0.f ? fexp2(q.p * flog2(a)) : 0.f; }
    if (k + 1 < q.n) { const float a = fabsf(q.sgn * (o.y - s.y) + q.eps); acc.y += a > 0.f ? fexp2(q.p * flog2(a)) : 0.f; }
  }
}
// g += coef * (1/p) d term / d owner.  The factor p lives in the pair coefficient (droot_of), the
// sign of the generic branch too.  Zero at e == 0 (torch.norm's backward masks the zero-norm
// entries; sign(0) = 0).
template <int PK>
__device__ __forceinline__ void gaccum2(f32x2& g, float coef, f32x2 o, f32x2 s, const Params& q, int k) {
  const f32x2 c2 = {coef, coef};
  if constexpr (PK == PK_DOT) {
    g = __builtin_elementwise_fma(c2, s, g);
  } else if constexpr (PK == 2) {
    g = __builtin_elementwise_fma(c2, o - s, g);
  } else if constexpr (PK == 1) {
    // g += coef * sign(d), sign(0) = 0: sign(d) = med3(d * 2^126, -1, 1) -- exactly +-1 for every normal difference (the product
    // is >= 1 or overflows to inf), exactly 0 for d = 0 -- as packed multiply, two v_med3 and a packed fma: 2.5 issue slots per
    // coordinate with the subtraction, where two compares + two selects + an add took 5.5 (the p = 1 backward sweep at n = 40
    // was 160 of its ~220 instructions per pair).  g +- coef by fma with +-1 is the same rounding as the add it replaces.
    const f32x2 d = o - s;
    const f32x2 big = {0x1p126f, 0x1p126f};
    f32x2 t = d * big;
    t.x = __builtin_amdgcn_fmed3f(t.x, -1.f, 1.f); t.y = __builtin_amdgcn_fmed3f(t.y, -1.f, 1.f);
    g = __builtin_elementwise_fma(c2, t, g);
  } else if constexpr (PK == 3) {
    const f32x2 d = o - s;
    const f32x2 t = {d.x * fabsf(d.x), d.y * fabsf(d.y)};
    g = __builtin_elementwise_fma(c2, t, g);
  } else {
    if (k < q.n) {
      const float e = q.sgn * (o.x - s.x) + q.eps, a = fabsf(e);
      const float v = a > 0.f ? fexp2((q.p - 1.f) * flog2(a)) : 0.f;
      g.x += coef * (e < 0.f ? -v : v);
    }
    if (k + 1 < q.n) {
      const float e = q.sgn * (o.y - s.y) + q.eps, a = fabsf(e);
      const float v = a > 0.f ? fexp2((q.p - 1.f) * flog2(a)) : 0.f;
      g.y += coef * (e < 0.f ? -v : v);
    }
  }
}
// neg value from the sum of powers, and p * d neg / d sum.  ROOT = false is the reference's default
// pow=True (neg = sum); ROOT = true takes the 1/p-th root (pow=False).
template <bool ROOT>
__device__ __forceinline__ float root_of(float s, const Params& q) {
  if constexpr (!ROOT) return s;
  if (q.p == 2.f) return sqrtf(s);
  if (q.p == 1.f) return s;
  return s > 0.f ? fexp2(q.inv_p * flog2(s)) : 0.f;
}
template <bool ROOT>
__device__ __forceinline__ float droot_of(float s, const Params& q) {
  if constexpr (!ROOT) return q.p;
  if (q.p == 1.f) return 1.f;
  return s > 0.f ? fexp2((q.inv_p - 1.f) * flog2(s)) : 0.f;
}

// ---- staging -----------------------------------------------------------------------------
// Global -> registers -> LDS, split in two so the global loads of tile t+1 are in flight while tile t
// is being consumed (double-buffered LDS, one barrier per tile).  Branch-free: masked-out elements
// read element 0 and are zeroed by a select at store time.
template <int NP>
struct Stager {
  static constexpr int TS = tile_rows(NP);
  static constexpr int ITERS = TS * NP / THREADS;
  static_assert((TS * NP) % THREADS == 0, "tile must divide over the workgroup");
  float v[ITERS];
  __device__ __forceinline__ void load(const float* __restrict__ str, int64_t lds, int64_t j0, int cnt, int n, float pre = 1.f) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = threadIdx.x + it * THREADS;
      const int row = idx / NP, k = idx - row * NP;
      const bool ok = row < cnt && k < n;
      const float x = str[ok ? (j0 + row) * lds + k : 0];
      v[it] = ok ? x * pre : 0.f;
    }
  }
  __device__ __forceinline__ void store(float* tile) const {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) tile[threadIdx.x + it * THREADS] = v[it];
  }
};

// owner r of this lane: row own0 + r * HALF + (lane & 31) -- both half-waves load the same rows
template <int NP, int R>
__device__ __forceinline__ void load_owners(f32x2 (&o)[R][NP / 2], const float* __restrict__ own, int64_t ldo,
                                            int64_t own0, int64_t n_own, int n, float pre = 1.f) {
  const int li = threadIdx.x & (HALF - 1);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t i = own0 + (int64_t)r * HALF + li;
    const bool ok = i < n_own;
#pragma unroll
    for (int k2 = 0; k2 < NP / 2; ++k2) {
      const bool a = ok && 2 * k2 < n, b = ok && 2 * k2 + 1 < n;
      const float x = own[a ? i * ldo + 2 * k2 : 0], y = own[b ? i * ldo + 2 * k2 + 1 : 0];
      o[r][k2].x = a ? x * pre : 0.f; o[r][k2].y = b ? y * pre : 0.f;
    }
  }
}

// sum_k term for one owner against JB stream rows of the LDS tile (wave-uniform ds_read_b128 broadcasts)
// NQ = number of dimension PAIRS that can hold real data (ceil(n / 2) <= NP / 2): a pair that is all padding
// (n = 9, 10 in the 12-wide layout) contributes exact zeros and is skipped at compile time -- same bits, two packed
// instructions fewer per pair and sweep.
template <int NP, int PK, int NQ = NP / 2, int JBT = JB>
__device__ __forceinline__ void dist_group(const f32x2 (&o)[NP / 2], const float* tile, int jj, const Params& q,
                                           float (&acc)[JBT]) {
  f32x2 a2[JBT];
#pragma unroll
  for (int c = 0; c < JBT; ++c) a2[c] = (f32x2){0.f, 0.f};
#pragma unroll
  for (int k4 = 0; k4 < NP / 4; ++k4) {
#pragma unroll
    for (int c = 0; c < JBT; ++c) {
      const float4 sv = *reinterpret_cast<const float4*>(&tile[(jj + c) * NP + 4 * k4]);
      if (2 * k4 < NQ) accum2<PK>(a2[c], o[2 * k4], (f32x2){sv.x, sv.y}, q, 4 * k4);
      if (2 * k4 + 1 < NQ) accum2<PK>(a2[c], o[2 * k4 + 1], (f32x2){sv.z, sv.w}, q, 4 * k4 + 2);
    }
  }
#pragma unroll
  for (int c = 0; c < JBT; ++c) acc[c] = a2[c].x + a2[c].y;
}

// The same, KEEPING the coordinate differences d = o - s of the JBT pairs in registers for the gradient pass of the backward sweep
// (difference-based kinds 1, 2, 3 only): the second pass then neither re-reads the stream rows from LDS nor repeats the packed
// subtraction -- 5 of the ~42 vector instructions per pair at n = 10.  2 * JBT * NQ more registers: narrow layouts only.
template <int NP, int PK, int NQ = NP / 2, int JBT = JB>
__device__ __forceinline__ void dist_group_keep(const f32x2 (&o)[NP / 2], const float* tile, int jj, float (&acc)[JBT], f32x2 (&d)[JBT][NQ]) {
  static_assert(PK >= 1 && PK <= 3, "difference-based pair kinds");
  f32x2 a2[JBT];
#pragma unroll
  for (int c = 0; c < JBT; ++c) a2[c] = (f32x2){0.f, 0.f};
#pragma unroll
  for (int k4 = 0; k4 < NP / 4; ++k4) {
#pragma unroll
    for (int c = 0; c < JBT; ++c) {
      const float4 sv = *reinterpret_cast<const float4*>(&tile[(jj + c) * NP + 4 * k4]);
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int k2 = 2 * k4 + h2;
        if (k2 < NQ) {
          const f32x2 dd = o[k2] - (h2 ? (f32x2){sv.z, sv.w} : (f32x2){sv.x, sv.y});
          d[c][k2] = dd;
          if constexpr (PK == 2) a2[c] = __builtin_elementwise_fma(dd, dd, a2[c]);
          else if constexpr (PK == 1) { a2[c].x = add_abs(a2[c].x, dd.x); a2[c].y = add_abs(a2[c].y, dd.y); }
          else { const f32x2 t = dd * dd; a2[c].x = fmaf(fabsf(dd.x), t.x, a2[c].x); a2[c].y = fmaf(fabsf(dd.y), t.y, a2[c].y); }
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < JBT; ++c) acc[c] = a2[c].x + a2[c].y;
}
// g += coef * (1/p) d term / d owner from the kept difference (same arithmetic as gaccum2 behind its subtraction)
template <int PK>
__device__ __forceinline__ void gaccum2_d(f32x2& g, float coef, f32x2 d) {
  const f32x2 c2 = {coef, coef};
  if constexpr (PK == 2) {
    g = __builtin_elementwise_fma(c2, d, g);
  } else if constexpr (PK == 1) {
    const f32x2 big = {0x1p126f, 0x1p126f};
    f32x2 t = d * big;
    t.x = __builtin_amdgcn_fmed3f(t.x, -1.f, 1.f); t.y = __builtin_amdgcn_fmed3f(t.y, -1.f, 1.f);
    g = __builtin_elementwise_fma(c2, t, g);
  } else {
    const f32x2 t = {d.x * fabsf(d.x), d.y * fabsf(d.y)};
    g = __builtin_elementwise_fma(c2, t, g);
  }
}

// ---- forward: per-split (max, sum) partials in the log2 domain -----------------------------
// ROWGRAD: also accumulate G_k = sum_j 2^(x_ij - m) * p * droot * (1/p) d term / d owner_k with the same
// running-max rescaling (the flash-attention forward with "V" = the pair's distance derivative).  After
// the splits are merged, G / 2^(lse) is the softmax-weighted row gradient sum_j w_ij d neg_ij / d owner_i,
// so the backward needs NO row pass: dz1 = pos-term + (-C_i / tau) * rowgrad_i for any upstream gradient.
// register budget hints (minimum waves per SIMD the kernel should fit): the sweeps are latency-tolerant only with >= 3-4 waves
constexpr int fwd_min_waves(int np, bool rowgrad) { return rowgrad ? (np <= 16 ? 3 : 2) : (np <= 16 ? 4 : (np <= 24 ? 3 : 2)); }
constexpr int bwd_min_waves(int np) { return np <= 16 ? 3 : 2; }

// (the sweep as a device function of the workgroup's (owner tile, stream split) = (bx, by): fwd_partial_k below is its kernel; lp_mfma.hip
//  instantiates it a second time INSIDE its matrix-core forward kernel as the guard's fallback -- one launch, no empty gated launch)
template <int NP, int PK, int R, bool ROOT, bool ROWGRAD, int NQ = NP / 2, bool ZMAX = false>
__device__ __forceinline__ void fwd_partial_body(
    const float* __restrict__ own, int64_t ldo, int64_t n_own,
    const float* __restrict__ str, int64_t lds, int64_t n_str,
    const Params& q, float2* __restrict__ part, float* __restrict__ part_g, int chunk, const int bx, const int by,
    float** scratch_out = nullptr /* fused finalize: the tile buffers (2 * TS * NP floats), free once every thread is past the partial store */) {
  constexpr int TS = tile_rows(NP), RPP = TS / PARTS, JBF = jb_fwd(NP);
  static_assert(RPP % JBF == 0, "partition rows must be whole JB groups");
  static_assert(WAVES * HALF * R * NP <= 2 * TS * NP, "the cross-wave merge reuses the tile buffers");
  __shared__ __attribute__((aligned(16))) float tiles[2][TS * NP];
  __shared__ float2 wred[WAVES][HALF * R];
  if (scratch_out) *scratch_out = &tiles[0][0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & (HALF - 1), hf = lane >> 5;
  const int pq = wave * 2 + hf;                       // this half-wave's partition of every tile
  const int64_t own0 = (int64_t)bx * (HALF * R);
  f32x2 o[R][NP / 2];
  f32x2 G[ROWGRAD ? R : 1][NP / 2];
  load_owners<NP, R>(o, own, ldo, own0, n_own, q.n, q.pre);
  float m[R], s[R];
  const float csgn = (PK == 0) ? q.sgn : 1.f;
#pragma unroll
  for (int r = 0; r < (ROWGRAD ? R : 1); ++r)
#pragma unroll
    for (int k2 = 0; k2 < NP / 2; ++k2) G[r][k2] = (f32x2){0.f, 0.f};
  static_assert(!(ZMAX && ROWGRAD), "the fixed-maximum forward is the training sweep (no row gradient)");
#pragma unroll
  for (int r = 0; r < R; ++r) { m[r] = ZMAX ? 0.f : -INFINITY; s[r] = 0.f; }
  const float xk = q.xs * q.kscale;

  const int64_t jb = (int64_t)by * chunk;
  const int64_t je = min(n_str, jb + (int64_t)chunk);
  Stager<NP> st;
  if (jb < je) {
    st.load(str, lds, jb, (int)min((int64_t)TS, je - jb), q.n, q.pre);
    st.store(tiles[0]);
  }
  __syncthreads();
  int cur = 0;
  for (int64_t j0 = jb; j0 < je; j0 += TS, cur ^= 1) {
    const int cnt = (int)min((int64_t)TS, je - j0);
    const bool more = j0 + TS < je;
    if (more) st.load(str, lds, j0 + TS, (int)min((int64_t)TS, je - j0 - TS), q.n, q.pre);   // in flight during the tile
    const float* tile = tiles[cur] + pq * RPP * NP;
    const int cq = min(RPP, max(0, cnt - pq * RPP));    // valid rows of this partition (ragged last tile only)
    // full partitions (every tile but a ragged last one) skip the per-row tail mask: same values, two instructions fewer per pair
    auto sweep = [&](auto ragged_tag) {
    constexpr bool RAGGED = decltype(ragged_tag)::value;
    for (int jj = 0; jj < cq; jj += JBF) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float acc[JBF];
        dist_group<NP, PK, NQ, JBF>(o[r], tile, jj, q, acc);
        if constexpr (ZMAX) {      // training sweep: maximum known to be 0 (Params::train), s >= 1 from the owner's own pool row; the
          float add = 0.f;         // coordinates arrive scaled (Params::pre), so the distance sum IS the negated logit
#pragma unroll
          for (int c = 0; c < JBF; ++c) add += (RAGGED && jj + c >= cq) ? 0.f : fexp2(PK == 1 ? acc[c] * xk : -acc[c]);     // (p = 1: unscaled, see Params::pre)
          s[r] += add;
          continue;
        }
        float x[JBF];
#pragma unroll
        for (int c = 0; c < JBF; ++c) {
          x[c] = root_of<ROOT>(acc[c], q) * xk;
          if (RAGGED && jj + c >= cq) x[c] = -INFINITY;      // ragged tail of the stream
        }
        // clamp keeps (-inf) - (-inf) out of the exponent when nothing valid was seen yet
        float xm = x[0];
#pragma unroll
        for (int c = 1; c < JBF; ++c) xm = fmaxf(xm, x[c]);
        const float mn = fmaxf(xm, fmaxf(m[r], -1e30f));
        float add = 0.f, e[JBF];
#pragma unroll
        for (int c = 0; c < JBF; ++c) { e[c] = fexp2(x[c] - mn); add += e[c]; }
        const float resc = fexp2(m[r] - mn);
        s[r] = fmaf(s[r], resc, add);
        m[r] = mn;
        if (ROWGRAD) {
#pragma unroll
          for (int c = 0; c < JBF; ++c) e[c] *= droot_of<ROOT>(acc[c], q) * csgn;
          asm volatile("" ::: "memory");   // re-read the tile for the second sweep instead of keeping it in VGPRs
          const f32x2 r2 = {resc, resc};
#pragma unroll
          for (int k4 = 0; k4 < NP / 4; ++k4) {
            G[r][2 * k4] *= r2; G[r][2 * k4 + 1] *= r2;
#pragma unroll
            for (int c = 0; c < JBF; ++c) {
              const float4 sv = *reinterpret_cast<const float4*>(&tile[(jj + c) * NP + 4 * k4]);
              if (2 * k4 < NQ) gaccum2<PK>(G[r][2 * k4], e[c], o[r][2 * k4], (f32x2){sv.x, sv.y}, q, 4 * k4);
              if (2 * k4 + 1 < NQ) gaccum2<PK>(G[r][2 * k4 + 1], e[c], o[r][2 * k4 + 1], (f32x2){sv.z, sv.w}, q, 4 * k4 + 2);
            }
          }
        }
      }
    }
    };
    if (cq == RPP) sweep(std::false_type{}); else sweep(std::true_type{});
    if (more) st.store(tiles[cur ^ 1]);     // last read one iteration ago, behind the previous barrier
    __syncthreads();
  }
  // ---- merge the workgroup's eight partitions on chip: half-waves by shuffle, waves through LDS (fixed order) ----
  float* gred = &tiles[0][0];               // [WAVES][HALF * R][NP]; every wave is past its last tile read (barrier above)
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float mo = __shfl_xor(m[r], HALF, 64), so = __shfl_xor(s[r], HALF, 64);
    const float mn = fmaxf(fmaxf(m[r], mo), -1e30f);
    const float fa = fexp2(m[r] - mn), fb = fexp2(mo - mn);
    // both halves compute the same merged value, in the same operand order (lower half first)
    const float s_lo = hf ? so : s[r], s_hi = hf ? s[r] : so, f_lo = hf ? fb : fa, f_hi = hf ? fa : fb;
    s[r] = fmaf(s_lo, f_lo, s_hi * f_hi);
    if (ROWGRAD) {
#pragma unroll
      for (int k2 = 0; k2 < NP / 2; ++k2) {
        const float gxo = __shfl_xor(G[r][k2].x, HALF, 64), gyo = __shfl_xor(G[r][k2].y, HALF, 64);
        const float gx_lo = hf ? gxo : G[r][k2].x, gx_hi = hf ? G[r][k2].x : gxo;
        const float gy_lo = hf ? gyo : G[r][k2].y, gy_hi = hf ? G[r][k2].y : gyo;
        G[r][k2].x = fmaf(gx_lo, f_lo, gx_hi * f_hi); G[r][k2].y = fmaf(gy_lo, f_lo, gy_hi * f_hi);
      }
    }
    m[r] = mn;
    if (hf == 0) {
      wred[wave][r * HALF + li] = make_float2(m[r], s[r]);
      if (ROWGRAD) {
        float4* dst = reinterpret_cast<float4*>(gred + ((size_t)wave * (HALF * R) + r * HALF + li) * NP);
#pragma unroll
        for (int k4 = 0; k4 < NP / 4; ++k4)
          dst[k4] = make_float4(G[r][2 * k4].x, G[r][2 * k4].y, G[r][2 * k4 + 1].x, G[r][2 * k4 + 1].y);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < HALF * R) {
    const int t = threadIdx.x;
    const int64_t i = own0 + t;
    float mm = -1e30f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) mm = fmaxf(mm, wred[w][t].x);
    float ss = 0.f, f[WAVES];
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { f[w] = fexp2(wred[w][t].x - mm); ss = fmaf(wred[w][t].y, f[w], ss); }
    if (i < n_own) {
      if (scratch_out) {      // fused finalize: another CU (possibly another XCD) reads this -- an agent-scope store (written through the XCD's
                              // L2), so that the arrival needs no release fence (a buffer_wbl2 per workgroup: +42 us on a 29 us sweep, measured)
        const float2 ms = make_float2(mm, ss);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(part + (int64_t)by * n_own + i), __builtin_bit_cast(unsigned long long, ms),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        part[(int64_t)by * n_own + i] = make_float2(mm, ss);
      }
      if (ROWGRAD) {
        float4* dst = reinterpret_cast<float4*>(part_g + ((int64_t)by * n_own + i) * NP);
#pragma unroll
        for (int k4 = 0; k4 < NP / 4; ++k4) {
          float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int w = 0; w < WAVES; ++w) {
            const float4 g4 = *reinterpret_cast<const float4*>(gred + ((size_t)w * (HALF * R) + t) * NP + 4 * k4);
            a4.x = fmaf(g4.x, f[w], a4.x); a4.y = fmaf(g4.y, f[w], a4.y); a4.z = fmaf(g4.z, f[w], a4.z); a4.w = fmaf(g4.w, f[w], a4.w);
          }
          dst[k4] = a4;
        }
      }
    }
  }
}

template <int NP, int PK, int R, bool ROOT, bool ROWGRAD, int NQ = NP / 2, bool ZMAX = false>
__global__ __launch_bounds__(THREADS, fwd_min_waves(NP, ROWGRAD)) void fwd_partial_k(
    const float* __restrict__ own, int64_t ldo, int64_t n_own,
    const float* __restrict__ str, int64_t lds, int64_t n_str,
    Params q, float2* __restrict__ part, float* __restrict__ part_g, int chunk) {
  fwd_partial_body<NP, PK, R, ROOT, ROWGRAD, NQ, ZMAX>(own, ldo, n_own, str, lds, n_str, q, part, part_g, chunk, blockIdx.x, blockIdx.y);
}

// ---- backward ------------------------------------------------------------------------------
// Owner-gradient partials.  STATS bit 0: softmax statistics of the OWNER rows weigh the pair (d/dz1: row
// reduction); bit 1: statistics of the STREAM rows do (d/dz3: column reduction over row-normalised
// weights); both (3) = the symmetric sweep used when the stream is the pool the owners belong to
// (z3 = all z1, main_mlp.py:272): d_ij = d_ji, so coef = C_i 2^(x - L_i) + C_j 2^(x - L_j) yields row AND
// column contributions to dz_i in one pass.  Same owner / partition layout as the forward; the eight partitions'
// gradient partials are summed on chip (shuffle, then LDS in wave order) before one NP-float row per owner and
// stream split goes to HBM.
template <int NP, int PK, int R, int STATS, bool ROOT, int NQ = NP / 2, bool FOLD = false>
__device__ __forceinline__ void bwd_pairs_body(
    const float* __restrict__ own, int64_t ldo, int64_t n_own,
    const float* __restrict__ str, int64_t lds, int64_t n_str,
    const Params& q, const float* __restrict__ statL, const float* __restrict__ statC,
    const float* __restrict__ strL, const float* __restrict__ strC,
    float* __restrict__ part, int chunk, const int bx, const int by) {
  constexpr bool OWNER_STATS = (STATS & 1) != 0, STREAM_STATS = (STATS & 2) != 0;
  constexpr int TS = tile_rows(NP), RPP = TS / PARTS;
  static_assert(WAVES * HALF * R * NP <= 2 * TS * NP, "the cross-wave merge reuses the tile buffers");
  __shared__ __attribute__((aligned(16))) float tiles[2][TS * NP];
  __shared__ float tLs[2][TS], tCs[2][TS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & (HALF - 1), hf = lane >> 5;
  const int pq = wave * 2 + hf;
  const int64_t own0 = (int64_t)bx * (HALF * R);
  f32x2 o[R][NP / 2], g[R][NP / 2];
  load_owners<NP, R>(o, own, ldo, own0, n_own, q.n, q.pre);
  float oL[R], oC[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t i = own0 + (int64_t)r * HALF + li;
    oL[r] = 0.f; oC[r] = 0.f;
    if (OWNER_STATS && i < n_own) { oL[r] = statL[i]; oC[r] = statC[i]; }
    if (FOLD) oC[r] *= fexp2(-oL[r]) * q.gfold;     // u_i = C_i 2^-L_i x (p / pre^(p-1)) (Params::train bit 1: L_i >= 0; gfold see Params)
#pragma unroll
    for (int k2 = 0; k2 < NP / 2; ++k2) g[r][k2] = (f32x2){0.f, 0.f};
  }
  const float xk = q.xs * q.kscale;
  const float csgn = (PK == 0) ? q.sgn : 1.f;
  const int64_t jb = (int64_t)by * chunk;
  const int64_t je = min(n_str, jb + (int64_t)chunk);
  Stager<NP> st;
  float rl = 0.f, rc = 0.f;     // staged stream statistics (threads < TS)
  auto load_stats = [&](int64_t j0, int cnt) {
    if (STREAM_STATS && (int)threadIdx.x < TS) {
      const bool ok = (int)threadIdx.x < cnt;
      const float l = strL[ok ? j0 + threadIdx.x : 0], c = strC[ok ? j0 + threadIdx.x : 0];
      rl = ok ? l : 0.f;
      rc = ok ? c : 0.f;         // zero coefficient masks the ragged tail
    }
  };
  auto store_stats = [&](int b) {
    if (STREAM_STATS && (int)threadIdx.x < TS) { tLs[b][threadIdx.x] = rl; tCs[b][threadIdx.x] = FOLD ? rc * fexp2(-rl) * q.gfold : rc; }
  };
  if (jb < je) {
    const int c0 = (int)min((int64_t)TS, je - jb);
    st.load(str, lds, jb, c0, q.n, q.pre); load_stats(jb, c0);
    st.store(tiles[0]); store_stats(0);
  }
  __syncthreads();
  int cur = 0;
  for (int64_t j0 = jb; j0 < je; j0 += TS, cur ^= 1) {
    const int cnt = (int)min((int64_t)TS, je - j0);
    const bool more = j0 + TS < je;
    if (more) {
      const int c1 = (int)min((int64_t)TS, je - j0 - TS);
      st.load(str, lds, j0 + TS, c1, q.n, q.pre); load_stats(j0 + TS, c1);
    }
    const float* tile = tiles[cur] + pq * RPP * NP;
    const float* tL = tLs[cur] + pq * RPP;
    const float* tC = tCs[cur] + pq * RPP;
    const int cq = min(RPP, max(0, cnt - pq * RPP));
    // (a tail-mask-free copy of this loop for full partitions, as in the forward, was measured 4 % SLOWER here: not kept)
#ifndef CLICA_LP_KEEPD_WIDE
#define CLICA_LP_KEEPD_WIDE 1
#endif
    // keep the coordinate differences for the gradient pass (dist_group_keep), ONE pair in flight (n = 40 sits at 255 registers with
    // two) -- measured 337 -> 243 us for the n = 40 symmetric sweep: the gradient pass's LDS re-read + second subtraction cost far
    // more than the second pair in flight hid
    constexpr bool KEEPD = PK >= 1 && PK <= 3 && (NP <= 16 || (CLICA_LP_KEEPD_WIDE && NP <= 40));
    constexpr int JW = KEEPD ? 1 : JBW;      // (narrow rows with kept differences: 1 / 2 / 4 pairs in flight = 50.5 / 51.2 / 57.8 us at n = 10)
    for (int jj = 0; jj < cq; jj += JW) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float acc[JW];
        f32x2 dk[JW][KEEPD ? NQ : 1];
        if constexpr (KEEPD) dist_group_keep<NP, PK, NQ, JW>(o[r], tile, jj, acc, dk);
        else dist_group<NP, PK, NQ, JW>(o[r], tile, jj, q, acc);
        float coef[JW];
#pragma unroll
        for (int c = 0; c < JW; ++c) {
          if constexpr (FOLD) {      // scaled coordinates: x = -acc; p, the sign and the gradient's 1 / pre^(p-1) live in the row factors
            static_assert(!FOLD || (STATS == 3 && !ROOT), "folded coefficient: symmetric training sweep only");
            float cf = fexp2(PK == 1 ? acc[c] * xk : -acc[c]) * (oC[r] + tC[jj + c]);   // one exponential, one add, one multiply per pair
            if (jj + c >= cq) cf = 0.f;
            coef[c] = cf;
            continue;
          }
          const float x = root_of<ROOT>(acc[c], q) * xk;
          float w = 0.f;
          if constexpr (FOLD) {
          } else {
            if (OWNER_STATS) w = oC[r] * fexp2(x - oL[r]);
            if (STREAM_STATS) w = fmaf(tC[jj + c], fexp2(x - tL[jj + c]), w);   // tC = 0 masks the ragged tail
          }
          float cf = w * droot_of<ROOT>(acc[c], q) * csgn;
          if (OWNER_STATS && jj + c >= cq) cf = 0.f;
          coef[c] = cf;
        }
        if constexpr (KEEPD) {           // narrow rows: the differences of the JW pairs are still in registers
#pragma unroll
          for (int k2 = 0; k2 < NQ; ++k2)
#pragma unroll
            for (int c = 0; c < JW; ++c) gaccum2_d<PK>(g[r][k2], coef[c], dk[c][k2]);
        } else {
          asm volatile("" ::: "memory");   // re-read the tile for the second sweep instead of keeping it in VGPRs
#pragma unroll
          for (int k4 = 0; k4 < NP / 4; ++k4) {
#pragma unroll
            for (int c = 0; c < JW; ++c) {
              const float4 sv = *reinterpret_cast<const float4*>(&tile[(jj + c) * NP + 4 * k4]);
              if (2 * k4 < NQ) gaccum2<PK>(g[r][2 * k4], coef[c], o[r][2 * k4], (f32x2){sv.x, sv.y}, q, 4 * k4);
              if (2 * k4 + 1 < NQ) gaccum2<PK>(g[r][2 * k4 + 1], coef[c], o[r][2 * k4 + 1], (f32x2){sv.z, sv.w}, q, 4 * k4 + 2);
            }
          }
        }
      }
    }
    if (more) { st.store(tiles[cur ^ 1]); store_stats(cur ^ 1); }
    __syncthreads();
  }
  // ---- sum the eight partitions on chip (fixed order: lower half + upper half, then waves 0..3) ----
  float* gred = &tiles[0][0];               // [WAVES][HALF * R][NP]
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int k2 = 0; k2 < NP / 2; ++k2) {
      const float gxo = __shfl_xor(g[r][k2].x, HALF, 64), gyo = __shfl_xor(g[r][k2].y, HALF, 64);
      g[r][k2].x = hf ? gxo + g[r][k2].x : g[r][k2].x + gxo;
      g[r][k2].y = hf ? gyo + g[r][k2].y : g[r][k2].y + gyo;
    }
    if (hf == 0) {
      float4* dst = reinterpret_cast<float4*>(gred + ((size_t)wave * (HALF * R) + r * HALF + li) * NP);
#pragma unroll
      for (int k4 = 0; k4 < NP / 4; ++k4)
        dst[k4] = make_float4(g[r][2 * k4].x, g[r][2 * k4].y, g[r][2 * k4 + 1].x, g[r][2 * k4 + 1].y);
    }
  }
  __syncthreads();
  constexpr int Q4 = NP / 4;
  for (int u = threadIdx.x; u < HALF * R * Q4; u += THREADS) {
    const int t = u / Q4, k4 = u - t * Q4;
    const int64_t i = own0 + t;
    if (i >= n_own) continue;
    float4 a4 = *reinterpret_cast<const float4*>(gred + (size_t)t * NP + 4 * k4);
#pragma unroll
    for (int w = 1; w < WAVES; ++w) {
      const float4 g4 = *reinterpret_cast<const float4*>(gred + ((size_t)w * (HALF * R) + t) * NP + 4 * k4);
      a4.x += g4.x; a4.y += g4.y; a4.z += g4.z; a4.w += g4.w;
    }
    *reinterpret_cast<float4*>(part + ((int64_t)by * n_own + i) * NP + 4 * k4) = a4;
  }
}

template <int NP, int PK, int R, int STATS, bool ROOT, int NQ = NP / 2, bool FOLD = false>
__global__ __launch_bounds__(THREADS, bwd_min_waves(NP)) void bwd_pairs_k(
    const float* __restrict__ own, int64_t ldo, int64_t n_own,
    const float* __restrict__ str, int64_t lds, int64_t n_str,
    Params q, const float* __restrict__ statL, const float* __restrict__ statC,
    const float* __restrict__ strL, const float* __restrict__ strC,
    float* __restrict__ part, int chunk) {
  bwd_pairs_body<NP, PK, R, STATS, ROOT, NQ, FOLD>(own, ldo, n_own, str, lds, n_str, q, statL, statC, strL, strC, part, chunk, blockIdx.x, blockIdx.y);
}

// ---- wide rows: 64 < n <= 512 ------------------------------------------------------------------
// The kernels above keep an owner's n coordinates (and its n gradient accumulators) in one lane's registers, which stops
// paying at n = 64.  Wider rows use a different cut of the same sweep: a workgroup owns W_OWN = 16 rows, SIXTEEN lanes
// share an owner and each holds every sixteenth coordinate (KC = 8 / 16 / 32 registers for n <= 128 / 256 / 512), the
// per-pair sum is closed with a four-step xor butterfly inside the 16-lane group (fixed order: deterministic), and every
// lane then updates its own slice of the gradient.  Stream rows are staged through LDS a tile at a time and read
// conflict-free (the 16 coordinate lanes hit 16 consecutive banks, the four owner groups of a wave broadcast).  One
// kernel template covers the five sweeps; the partial formats are those of fwd_partial_k / bwd_pairs_k, so finalize,
// rowgrad_combine_k and bwd_reduce_k are shared.  None of the reference's configurations is this wide (n <= 40 there);
// the path exists so that the loss has no dimension cliff, and is sized for correctness first.
constexpr int W_OWN = 16, W_KL = 16;
constexpr int kMaxWideN = 512;
constexpr int wide_kc(int n) { return n <= 128 ? 8 : (n <= 256 ? 16 : 32); }
constexpr int wide_ts(int kc) { return kc == 32 ? 16 : 32; }            // 32 KB tile at KC >= 16
enum WideMode { W_FWD = 0, W_FWD_ROWGRAD = 1, W_BWD_OWNER = 2, W_BWD_STREAM = 3, W_BWD_SYM = 4 };

template <int PK>
__device__ __forceinline__ float term1(float o, float s, const Params& q) {
  if constexpr (PK == PK_DOT) return o * s;
  else if constexpr (PK == 2) { const float d = o - s; return d * d; }
  else if constexpr (PK == 1) return fabsf(o - s);
  else if constexpr (PK == 3) { const float d = o - s; return fabsf(d) * d * d; }
  else { const float a = fabsf(q.sgn * (o - s) + q.eps); return a > 0.f ? fexp2(q.p * flog2(a)) : 0.f; }
}
// (1/p) d term / d owner, sign of the generic branch's `sgn` left to the pair coefficient (as in gaccum2)
template <int PK>
__device__ __forceinline__ float dterm1(float o, float s, const Params& q) {
  if constexpr (PK == PK_DOT) return s;
  else if constexpr (PK == 2) return o - s;
  else if constexpr (PK == 1) { const float d = o - s; return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
  else if constexpr (PK == 3) { const float d = o - s; return d * fabsf(d); }
  else {
    const float e = q.sgn * (o - s) + q.eps, a = fabsf(e);
    const float v = a > 0.f ? fexp2((q.p - 1.f) * flog2(a)) : 0.f;
    return e < 0.f ? -v : v;
  }
}

template <int KC, int PK, int MODE>
__global__ __launch_bounds__(THREADS) void wide_k(
    const float* __restrict__ own, int64_t ldo, int64_t n_own,
    const float* __restrict__ str, int64_t lds, int64_t n_str, Params q,
    const float* __restrict__ ownL, const float* __restrict__ ownC,
    const float* __restrict__ strL, const float* __restrict__ strC,
    float2* __restrict__ part, float* __restrict__ part_g, int np, int chunk) {
  constexpr bool FWD = MODE == W_FWD || MODE == W_FWD_ROWGRAD, ROWGRAD = MODE == W_FWD_ROWGRAD;
  constexpr bool OWNER_STATS = MODE == W_BWD_OWNER || MODE == W_BWD_SYM, STREAM_STATS = MODE == W_BWD_STREAM || MODE == W_BWD_SYM;
  constexpr bool GRAD = MODE != W_FWD;
  constexpr int TS = wide_ts(KC), LDT = KC * W_KL;
  __shared__ __attribute__((aligned(16))) float tile[TS * LDT];
  __shared__ float tL[TS], tC[TS];
  const int oi = threadIdx.x >> 4, kl = threadIdx.x & (W_KL - 1);
  const int64_t i = (int64_t)blockIdx.x * W_OWN + oi;
  const bool own_ok = i < n_own;
  float o[KC], g[GRAD ? KC : 1];
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const int k = kl + W_KL * c;
    const bool ok = own_ok && k < q.n;
    const float x = own[ok ? i * ldo + k : 0];
    o[c] = ok ? x : 0.f;
  }
#pragma unroll
  for (int c = 0; c < (GRAD ? KC : 1); ++c) g[c] = 0.f;
  float oL = 0.f, oC = 0.f;
  if (OWNER_STATS && own_ok) { oL = ownL[i]; oC = ownC[i]; }
  float m = -INFINITY, s = 0.f;
  const float xk = q.xs * q.kscale;
  const float csgn = (PK == 0) ? q.sgn : 1.f;
  const bool root = !q.pow;
  const int64_t jb = (int64_t)blockIdx.y * chunk;
  const int64_t je = min(n_str, jb + (int64_t)chunk);
  for (int64_t j0 = jb; j0 < je; j0 += TS) {
    const int cnt = (int)min((int64_t)TS, je - j0);
    __syncthreads();                                  // the previous tile has been consumed
    for (int idx = threadIdx.x; idx < TS * LDT; idx += THREADS) {
      const int row = idx / LDT, k = idx - row * LDT;
      const bool ok = row < cnt && k < q.n;
      const float x = str[ok ? (j0 + row) * lds + k : 0];
      tile[idx] = ok ? x : 0.f;
    }
    if (STREAM_STATS && (int)threadIdx.x < TS) {
      const bool ok = (int)threadIdx.x < cnt;
      const float l = strL[ok ? j0 + threadIdx.x : 0], c = strC[ok ? j0 + threadIdx.x : 0];
      tL[threadIdx.x] = ok ? l : 0.f; tC[threadIdx.x] = ok ? c : 0.f;
    }
    __syncthreads();
    for (int jj = 0; jj < cnt; ++jj) {
      const float* row = tile + jj * LDT + kl;
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < KC; ++c)
        if (PK != 0 || kl + W_KL * c < q.n) acc += term1<PK>(o[c], row[W_KL * c], q);   // zero padding is neutral except with eps
#pragma unroll
      for (int off = W_KL / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
      const float x = (root ? root_of<true>(acc, q) : acc) * xk;
      const float dr = (root ? droot_of<true>(acc, q) : q.p) * csgn;
      float cf;
      if (FWD) {
        const float mn = fmaxf(x, fmaxf(m, -1e30f));
        const float e = fexp2(x - mn), resc = fexp2(m - mn);
        s = fmaf(s, resc, e);
        m = mn;
        cf = e * dr;
        if (ROWGRAD) {
#pragma unroll
          for (int c = 0; c < KC; ++c) g[c] *= resc;
        }
      } else {
        float w = 0.f;
        if (OWNER_STATS) w = oC * fexp2(x - oL);
        if (STREAM_STATS) w = fmaf(tC[jj], fexp2(x - tL[jj]), w);
        cf = w * dr;
      }
      if (GRAD) {
#pragma unroll
        for (int c = 0; c < KC; ++c)
          if (PK != 0 || kl + W_KL * c < q.n) g[c] = fmaf(cf, dterm1<PK>(o[c], row[W_KL * c], q), g[c]);
      }
    }
  }
  if (!own_ok) return;
  const int64_t slot = (int64_t)blockIdx.y * n_own + i;
  if (FWD && kl == 0) part[slot] = make_float2(fmaxf(m, -1e30f), s);
  if (GRAD) {
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const int k = kl + W_KL * c;
      if (k < np) part_g[slot * np + k] = k < q.n ? g[c] : 0.f;
    }
  }
}

// ---- host-side planning -----------------------------------------------------------------------
inline int pad_dim(int n) {
  static const int dims[] = {4, 8, 12, 16, 24, 32, 40, 64};
  for (int d : dims) if (n <= d) return d;
  if (n <= kMaxWideN) return (n + 3) / 4 * 4;      // wide rows (wide_k): partial rows padded to whole float4s
  return -1;
}
constexpr int owners_fwd(int np) { return np <= 24 ? 2 : 1; }
constexpr int owners_bwd(int np) { return np <= 16 ? 2 : 1; }

struct Plan {
  int np, R;
  int64_t tiles;  // owner tiles (32 R owners each)
  int nsplit, chunk;
};
inline Plan make_plan(int64_t n_own, int64_t n_str, int n, bool bwd, int resident = 0) {
  Plan P;
  P.np = pad_dim(n);
  if (P.np > 64) {          // wide rows: 16 owners per workgroup, about four workgroups per CU, splits in whole LDS tiles
    const int64_t TSW = wide_ts(wide_kc(n));
    P.R = 0;
    P.tiles = ceil_div(n_own, (int64_t)W_OWN);
    int64_t ns = ceil_div((int64_t)kNumCU * 4, P.tiles);
    const int64_t cap = ceil_div(n_str, TSW);
    if (ns > cap) ns = cap;
    if (ns < 1) ns = 1;
    const int64_t chunk = ceil_div(ceil_div(n_str, ns), TSW) * TSW;
    P.chunk = (int)chunk;
    P.nsplit = (int)ceil_div(n_str, chunk);
    return P;
  }
  P.R = bwd ? owners_bwd(P.np) : owners_fwd(P.np);
  P.tiles = ceil_div(n_own, (int64_t)HALF * P.R);
  const int64_t TS = tile_rows(P.np);
  // HBM-level stream splits.  A workgroup already cuts its chunk eight ways on chip, so HBM-level splits only exist to fill
  // the chip (cost model below).  Splits are whole LDS tiles.  E.g. B = B3 = 6144, n = 10: 96 tiles x 8 splits of 768 rows
  // = exactly 3 workgroups per CU; the 8-rank pool (B3 = 49 152): 96 x 24 splits of 2048 rows = three rounds of that.
  const int64_t max_split = ceil_div(n_str, TS);
  auto finish = [&](int64_t ns) {
    if (ns < 1) ns = 1;
    if (ns > max_split) ns = max_split;
    int64_t chunk = ceil_div(ceil_div(n_str, ns), TS) * TS;
    if (chunk < TS) chunk = TS;
    P.chunk = (int)chunk;
    P.nsplit = (int)(n_str > 0 ? ceil_div(n_str, chunk) : 1);
  };
  // Cost model (fitted to tools/loss_train_probe.py sweeps, B = 6144, pools of 6144 and 49 152 rows, n = 10 and n = 40):
  // a CU holds RES workgroups at a time (register-limited: 3 waves per SIMD up to the 16-wide layout, 2 above), which share
  // it, so a "round" of RES workgroups per CU lasts RES x (rows per split + a fixed prologue / merge cost of ~192 rows'
  // worth).  Workgroups do not leave a round together -- the CU runs under-occupied while the stragglers finish -- which
  // costs about a quarter of a round per launch however many rounds there are: more, shorter rounds amortise it (pool of
  // 49 152 rows: 24 splits of 2048 rows instead of 8 of 6144 is 7 % / 5 % faster forward / backward), until the fixed
  // cost per workgroup wins (pool of 6144 rows: 8 splits of 768 stay best).
  double best = 1e300; int64_t best_ns = 1;
  if (resident < 0) {
    // one-round model (the nearest-neighbour sweep, nn_search.hip: few query tiles, a very long stream): every workgroup of
    // the launch is taken as resident, 2..6 per CU, the launch lasts as long as its most loaded CU
    const int64_t lo1 = ceil_div((int64_t)kNumCU * 2, P.tiles), hi1 = ceil_div((int64_t)kNumCU * 6, P.tiles);
    const int64_t h1 = hi1 > max_split ? max_split : (hi1 < 1 ? 1 : hi1);
    const int64_t l1 = lo1 < 1 ? 1 : (lo1 > h1 ? h1 : lo1);
    for (int64_t ns = l1; ns <= h1; ++ns) {
      const int64_t chunk = ceil_div(ceil_div(n_str, ns), TS) * TS;
      const int64_t nsp = ceil_div(n_str, chunk);
      const int64_t rounds = ceil_div(P.tiles * nsp, (int64_t)kNumCU);
      const double cost = (double)rounds * (double)(chunk + 192) * (1.0 + 0.03 * (rounds < 3 ? 3 - rounds : 0)) + 4.0 * (double)nsp;
      if (cost < best - 1e-9) { best = cost; best_ns = nsp; }
    }
    finish(best_ns);
    return P;
  }
  const int64_t RES = resident > 0 ? resident : (P.np <= 16 ? 3 : 2);
  const int64_t slots = (int64_t)kNumCU * RES;
  const int64_t lo = ceil_div(slots, P.tiles), hi = ceil_div(slots * 6, P.tiles);
  const int64_t ns_hi = hi > max_split ? max_split : (hi < 1 ? 1 : hi);
  const int64_t ns_lo = lo < 1 ? 1 : (lo > ns_hi ? ns_hi : lo);        // short streams: as many splits as there are tiles of rows
  for (int64_t ns = ns_lo; ns <= ns_hi; ++ns) {
    const int64_t chunk = ceil_div(ceil_div(n_str, ns), TS) * TS;
    const int64_t nsp = ceil_div(n_str, chunk);
    const int64_t rounds = ceil_div(P.tiles * nsp, slots);
    const double cost = ((double)rounds + 0.25) * (double)RES * (double)(chunk + 192) + 4.0 * (double)nsp;
    if (cost < best - 1e-9) { best = cost; best_ns = nsp; }
  }
  finish(best_ns);
  return P;
}


// per-exponent-kind launchers, one translation unit each (lp_loss_pk.hip with -DCLICA_PK=k)
#define CLICA_LP_DECLARE(PKV)                                                                          \
  void launch_fwd_partial_pk##PKV(const Plan& P, const float* own, int64_t ldo, int64_t n_own,        \
                                  const float* str, int64_t lds, int64_t n_str, const Params& q,      \
                                  float2* part, float* part_g, hipStream_t st);                       \
  void launch_bwd_pairs_pk##PKV(const Plan& P, bool owner_stats, const float* own, int64_t ldo,       \
                                int64_t n_own, const float* str, int64_t lds, int64_t n_str,          \
                                const Params& q, const float* statL, const float* statC, float* part, \
                                hipStream_t st);                                                      \
  void launch_bwd_sym_pk##PKV(const Plan& P, const float* own, int64_t ldo, int64_t n_own,            \
                              const float* str, int64_t lds, int64_t n_str, const Params& q,          \
                              const float* ownL, const float* ownC, const float* strL, const float* strC, \
                              float* part, hipStream_t st);
CLICA_LP_DECLARE(0) CLICA_LP_DECLARE(1) CLICA_LP_DECLARE(2) CLICA_LP_DECLARE(3) CLICA_LP_DECLARE(4)

}  // namespace lp
}  // namespace clica
