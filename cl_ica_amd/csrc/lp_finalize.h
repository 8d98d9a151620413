// Finishing step of the Lp / dot InfoNCE forward, shared between its own launch (lp_loss.hip: fwd_finalize_k) and the fused training
// sweep (lp_loss_pk.hip: fwd_partial_fin_k).  Row-wise restatement of /root/reference/losses.py:458-477 (loss_pos, logsumexp /
// _logmeanexp over the negatives (+ the positive pair in simclr_compatibility_mode), loss_i, the three means).
#pragma once
#include "lp_kernels.h"

namespace clica {
namespace lp {

// ---- positive pair ---------------------------------------------------------------------------
// returns sum of powers; frac (p<1) branch: (|z1-z2| + 1e-12)^p  (losses.py:439-441)
__device__ __forceinline__ float pos_sum(const float* a, const float* b, int n, const Params& q, bool frac) {
  float s = 0.f;
  for (int k = 0; k < n; ++k) {
    float d = a[k] - b[k];
    float t;
    if (frac) t = fexp2(q.p * flog2(fabsf(d) + 1e-12f));
    else if (q.p == 2.f) t = d * d;
    else if (q.p == 1.f) t = fabsf(d);
    else if (q.p == 3.f) t = fabsf(d) * d * d;
    else t = fabsf(d) > 0.f ? fexp2(q.p * flog2(fabsf(d))) : 0.f;
    s += t;
  }
  return s;
}

struct Means {
  float* blocksums;      // [gridDim.x][3] per-block partial sums
};

// block tree -> one slot per block; `means_k` (next launch on the stream) sums the slots in index
// order, so the three means are deterministic and need no atomics or fences.
// (`blk`: the finalize block's index -- blockIdx.x of fwd_finalize_k, the owner tile of the fused sweep)
__device__ __forceinline__ void reduce_means(float v0, float v1, float v2, const Means& M, const int blk, const bool coherent = false) {
  __shared__ float red[3][THREADS / 64];
  float v[3] = {v0, v1, v2};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[a] += __shfl_down(v[a], off, 64);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[0][wave] = v[0]; red[1][wave] = v[1]; red[2][wave] = v[2]; }
  __syncthreads();
  if (threadIdx.x < 3) {
    float t = 0.f;
    for (int w = 0; w < THREADS / 64; ++w) t += red[threadIdx.x][w];
    if (coherent) __hip_atomic_store(&M.blocksums[blk * 3 + threadIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // read by another workgroup of this launch
    else M.blocksums[blk * 3 + threadIdx.x] = t;
  }
}


constexpr int FIN_ROWS = 64;   // rows per finalize block; the 4 waves split the per-split partials
// training forward: the finalize thread of a row also does that row's coefficient step (statistics for the pair
// sweep + the positive-pair gradient, upstream gradient = d(mean loss) = 1), saving the bwd_coef_k launch
struct TrainOut { float* statL; float* statC; float* dz1; int64_t ldd1; float* dz2; int64_t ldd2; };

// one row of the coefficient step (shared by bwd_coef_k and the training forward's finalize)
__device__ __forceinline__ void coef_row(
    const int64_t i, const int64_t rows, const float* a /* row i of z1 */, const float* b /* row i of z2 */,
    const Params& q, float tau, float alpha, int compat, int frac, int dot, const float L2,
    const float* __restrict__ g_mean, const float* __restrict__ g_item,
    const float* __restrict__ g_pos, const float* __restrict__ g_neg,
    float* __restrict__ statL, float* __restrict__ statC,
    float* __restrict__ dz1, int64_t ldd1, float* __restrict__ dz2, int64_t ldd2, float* statC_value = nullptr) {
  const float inv_rows = 1.f / (float)rows;
  const float gi = (g_mean ? g_mean[0] : 1.f) * inv_rows + (g_item ? g_item[i] : 0.f);
  const float A = 2.f * alpha * gi + (g_pos ? g_pos[0] * inv_rows : 0.f);
  const float C = 2.f * (1.f - alpha) * gi + (g_neg ? g_neg[0] * inv_rows : 0.f);
  statL[i] = L2;      // row statistic stays in the log2 domain end to end: no ln <-> log2 round trip of a number that is
                      // ~10^3 in saturated rows (each rounding of it is a 1e-4 relative error on every weight of the row)
  const float sc = q.xs * C / tau;
  statC[i] = sc;
  if (statC_value) *statC_value = sc;       // (the caller's copy: reading statC[i] back is a load behind this thread's own stores -- a vmcnt(0))
  if (!dz1 && !dz2) return;
  if (dot) {
    float pos = 0.f;
    if (q.posdot) pos = q.posdot[i];
    else for (int k = 0; k < q.n; ++k) pos += a[k] * b[k];
    const float dpos = -A / tau + (C / tau) * fexp2(pos * q.kscale - L2);
    if (q.dpos) { q.dpos[i] = dpos; return; }       // wide rows: dpos_apply_k writes dz1 = dpos z2, dz2 = dpos z1
    for (int k = 0; k < q.n; ++k) {
      if (dz1) dz1[i * ldd1 + k] = dpos * b[k];
      if (dz2) dz2[i * ldd2 + k] = dpos * a[k];
    }
    return;
  }
  const float sp_ = pos_sum(a, b, q.n, q, frac != 0);
  const float pos = q.pow ? sp_ : root_of<true>(sp_, q);
  float cpos = A / tau;
  if (compat) cpos -= (C / tau) * fexp2(-pos * q.kscale - L2);
  cpos *= q.pow ? q.p : droot_of<true>(sp_, q);  // includes the factor p
  for (int k = 0; k < q.n; ++k) {
    const float d = a[k] - b[k];
    float dt;
    if (frac) {
      const float v = fexp2((q.p - 1.f) * flog2(fabsf(d) + 1e-12f));
      dt = d > 0.f ? v : (d < 0.f ? -v : 0.f);
    } else if (q.p == 2.f) dt = d;
    else if (q.p == 1.f) dt = (d > 0.f ? 1.f : 0.f) - (d < 0.f ? 1.f : 0.f);
    else if (q.p == 3.f) dt = d * fabsf(d);
    else {
      const float ad = fabsf(d);
      const float v = ad > 0.f ? fexp2((q.p - 1.f) * flog2(ad)) : 0.f;
      dt = d < 0.f ? -v : v;
    }
    const float g = cpos * dt;
    if (dz1) dz1[i * ldd1 + k] = g;
    if (dz2) dz2[i * ldd2 + k] = -g;
  }
}


// The per-row part of the finalize for block `blk` (FIN_ROWS rows, THREADS threads): merge of the per-split (max, sum) partials, the
// positive pair, loss_i / pos_i / lse_i, and -- TrainOut -- the coefficient step.  Shared by fwd_finalize_k (its own launch) and by the
// fused training sweep (fwd_partial_fin_k: the LAST workgroup to deliver a partial of an owner tile runs it for that tile).
// Scratch in LDS: sm, ss [THREADS / FIN_ROWS][FIN_ROWS]; zr0, zr1 [FIN_ROWS * zmaxn] (rows of z1 / z2 staged when n <= zmaxn);
// ush [FIN_ROWS] or nullptr (u_i = C_i 2^-L_i for the caller's feature planes).  Returns through v_*: this thread's summands of the means.
// COHERENT: the partials were written by other workgroups of THIS launch (agent-scope stores): read them with agent-scope loads.
struct FinScratch { float* sm; float* ss; float* zr0; float* zr1; int zmaxn; float* ush; };
template <bool COHERENT = false>
__device__ __forceinline__ void finalize_rows(
    const int blk, const float2* __restrict__ part, const int nsplit, const int64_t rows,
    const float* __restrict__ z1, int64_t ld1, const float* __restrict__ z2, int64_t ld2,
    const Params& q, float tau, float alpha, int compat, int frac, int dot, float log_b3,
    float* __restrict__ loss_i, float* __restrict__ pos_i, float* __restrict__ lse_i, const TrainOut& T, const FinScratch& S,
    float& v_loss, float& v_pos, float& v_lse) {
  // the block's z1 / z2 rows, staged by all 256 threads (coalesced) while the partials are in flight: the finishing
  // threads then read their row's coordinates from LDS instead of starting two more dependent global round trips
  const int lane_row = threadIdx.x & (FIN_ROWS - 1), grp = threadIdx.x / FIN_ROWS;
  const int64_t i = (int64_t)blk * FIN_ROWS + lane_row;
  const bool staged = q.n <= S.zmaxn;
  if (staged) {
    const int64_t r0 = (int64_t)blk * FIN_ROWS;
    const int cnt = (int)min((int64_t)FIN_ROWS, rows - r0) * q.n;
    for (int idx = threadIdx.x; idx < cnt; idx += THREADS) {
      const int r = idx / q.n, k = idx - r * q.n;
      S.zr0[idx] = z1[(r0 + r) * ld1 + k];
      S.zr1[idx] = z2[(r0 + r) * ld2 + k];
    }
  }
  float m = -1e30f, s = 0.f;
  if (i < rows) {
    // four partials in flight per round (a one-at-a-time loop is a chain of dependent L2 round trips);
    // missing ones are (m = -1e30, s = 0): they leave the running pair unchanged
    constexpr int G = THREADS / FIN_ROWS;
    for (int sp0 = grp; sp0 < nsplit; sp0 += 4 * G) {
      float2 ps[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int sp = sp0 + u * G;
        if (!(sp < nsplit)) ps[u] = make_float2(-1e30f, 0.f);
        else if (COHERENT)
          ps[u] = __builtin_bit_cast(float2, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(part + (int64_t)sp * rows + i),
                                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        else ps[u] = part[(int64_t)sp * rows + i];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float mn = fmaxf(m, ps[u].x);
        s = s * fexp2(m - mn) + ps[u].y * fexp2(ps[u].x - mn);
        m = mn;
      }
    }
  }
  S.sm[grp * FIN_ROWS + lane_row] = m; S.ss[grp * FIN_ROWS + lane_row] = s;
  __syncthreads();
  v_loss = 0.f; v_pos = 0.f; v_lse = 0.f;
  if (grp == 0 && i < rows) {
#pragma unroll
    for (int w = 1; w < THREADS / FIN_ROWS; ++w) {
      const float pm = S.sm[w * FIN_ROWS + lane_row], psum = S.ss[w * FIN_ROWS + lane_row];
      const float mn = fmaxf(m, pm);
      s = s * fexp2(m - mn) + psum * fexp2(pm - mn);
      m = mn;
    }
    float pos, xp;
    const float* ra = staged ? &S.zr0[lane_row * q.n] : z1 + i * ld1;
    const float* rb = staged ? &S.zr1[lane_row * q.n] : z2 + i * ld2;
    if (dot) {   // SimCLRLoss: pos = <z1, z2> and the logit is +pos/tau (losses.py:188-193)
      pos = 0.f;
      if (q.posdot) pos = q.posdot[i];
      else for (int k = 0; k < q.n; ++k) pos += ra[k] * rb[k];
      xp = pos * q.kscale;
      pos = -pos;  // loss_pos = -pos/tau (losses.py:192)
    } else {
      const float sp_ = pos_sum(ra, rb, q.n, q, frac != 0);
      pos = q.pow ? sp_ : root_of<true>(sp_, q);
      xp = -pos * q.kscale;
    }
    if (compat) {  // positive pair joins the softmax denominator (losses.py:459-462)
      const float mn = fmaxf(m, xp);
      s = s * fexp2(m - mn) + fexp2(xp - mn);
      m = mn;
    }
    const float L2 = m + flog2(s);                // log2-domain log-sum-exp of the scaled logits: THE saved row statistic
    const float lse_raw = L2 * kLn2;
    const float lse = compat ? lse_raw : lse_raw - log_b3;   // _logmeanexp, losses.py:506-510
    const float lp = pos / tau;
    const float li = 2.f * (alpha * lp + (1.f - alpha) * lse);
    loss_i[i] = li; pos_i[i] = lp; lse_i[i] = L2;
    float sc_i = 0.f;
    if (T.statL)
      coef_row(i, rows, ra, rb, q, tau, alpha, compat, frac, dot, L2, nullptr, nullptr, nullptr, nullptr, T.statL, T.statC,
               T.dz1, T.ldd1, T.dz2, T.ldd2, &sc_i);
    v_loss = li; v_pos = lp; v_lse = lse;
    if (S.ush) S.ush[lane_row] = sc_i * fexp2(-L2);            // u_i = C_i 2^-L_i (the value coef_row has just stored in statC[i])
  } else if (S.ush && grp == 0) {
    S.ush[lane_row] = 0.f;
  }
}

// ---- the training step's forward sweep with its finalize folded in (round 6: one launch fewer) ----------------------------------------
// Grid = (owner tiles, stream splits) as fwd_partial_k.  Every workgroup delivers its (max, sum) partials (agent-scope stores), then --
// one relaxed agent-scope atomic on the tile's arrival counter -- the LAST of the tile's `nsplit` workgroups runs finalize_rows +
// reduce_means for the tile's 64 rows (agent-scope loads: the other partials came from other CUs, possibly other XCDs) and puts the
// counter back to zero for the next launch.  The merge order of the partials is finalize_rows' (split index), whoever arrives last: results are those
// of the two-launch form bit for bit.  Scratch: the sweep's tile buffers (z rows) + 2 KB of its own.
// Counters: `arrive[tiles]`, all zero before the first launch and between launches (the launch leaves them zero).
// `means` (generic forward, nullptr in the training pair whose means the backward chain finishes): the LAST tile finisher of the launch
// (second arrival counter, arrive[gridDim.x]) sums the block sums in means_k's order -- lane l takes slots l, l + 64, ..., then the
// shuffle tree -- and writes the three means: same bits as the means_k launch.
struct FinArgs {
  const float* z2; int64_t ld2; float tau, alpha; int compat; float log_b3;
  float* loss_i; float* pos_i; float* lse_i; Means M; TrainOut T; int* arrive;
  float* means; float inv_count;
};
template <int NP, int PK, int R, bool ROOT, int NQ, bool ZMAX>
__global__ __launch_bounds__(THREADS, fwd_min_waves(NP, false)) void fwd_partial_fin_k(
    const float* __restrict__ own, int64_t ldo, int64_t n_own,
    const float* __restrict__ str, int64_t lds, int64_t n_str,
    Params q, float2* __restrict__ part, int chunk, FinArgs F) {
  static_assert(HALF * R == FIN_ROWS && NP <= 16, "an owner tile of the sweep = one finalize block; z rows staged in the tile buffers");
  float* scratch = nullptr;
  fwd_partial_body<NP, PK, R, ROOT, false, NQ, ZMAX>(own, ldo, n_own, str, lds, n_str, q, part, nullptr, chunk, blockIdx.x, blockIdx.y, &scratch);
  __shared__ float sm[THREADS / FIN_ROWS][FIN_ROWS], ss[THREADS / FIN_ROWS][FIN_ROWS];
  __shared__ int s_last;
  // No fences: a release fence at agent scope is a write-back scan of the XCD's L2 (buffer_wbl2) per workgroup -- measured +42 us on a
  // 29 us sweep.  Instead the partials are agent-scope stores (written through), each storing thread waits for its store's
  // acknowledgement, the barrier collects the workgroup, and only then does thread 0 arrive; the finishing workgroup reads the partials
  // with agent-scope loads (past its L1 and the XCD's L2 copies).
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this thread's partial store has been acknowledged
  __syncthreads();                         // (also: every thread is past its last use of the tile buffers)
  if (threadIdx.x == 0) {
    const int prev = __hip_atomic_fetch_add(&F.arrive[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = prev == (int)gridDim.y - 1;
    if (s_last) __hip_atomic_store(&F.arrive[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // all arrivals of this launch are in
  }
  __syncthreads();
  if (!s_last) return;
  float v_loss, v_pos, v_lse;
  constexpr int ZMAXN = 2 * tile_rows(NP) * NP / (2 * FIN_ROWS);      // floats per staged row the tile buffers hold for z1 and z2 each
  static_assert(ZMAXN >= NP, "tile buffers too small for the z rows");
  finalize_rows<true>((int)blockIdx.x, part, (int)gridDim.y, n_own, own, ldo, F.z2, F.ld2, q, F.tau, F.alpha, F.compat, 0, 0, F.log_b3,
                F.loss_i, F.pos_i, F.lse_i, F.T, FinScratch{&sm[0][0], &ss[0][0], scratch, scratch + FIN_ROWS * ZMAXN, ZMAXN, nullptr},
                v_loss, v_pos, v_lse);
  reduce_means(v_loss, v_pos, v_lse, F.M, (int)blockIdx.x, /*coherent=*/F.means != nullptr);
  if (!F.means) return;
  // second level: the last tile finisher sums all block sums (they were agent-scope stores of threads 0..2; wait for their acknowledgement)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int prev = __hip_atomic_fetch_add(&F.arrive[gridDim.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = prev == (int)gridDim.x - 1;
    if (s_last) __hip_atomic_store(&F.arrive[gridDim.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_last || threadIdx.x >= 64) return;
  float v[3] = {0.f, 0.f, 0.f};
  for (int b = threadIdx.x; b < (int)gridDim.x; b += 64) {
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] += __hip_atomic_load(&F.M.blocksums[b * 3 + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[a] += __shfl_down(v[a], off, 64);
  }
  if (threadIdx.x == 0) { F.means[0] = v[0] * F.inv_count; F.means[1] = v[1] * F.inv_count; F.means[2] = v[2] * F.inv_count; }
}

// per-exponent-kind launchers of the fused form (lp_loss_pk.hip with -DCLICA_PK=k); false: no such form, nothing launched
#define CLICA_LP_DECLARE_FIN(PKV)                                                                          \
  bool launch_fwd_partial_fin_pk##PKV(const Plan& P, const float* own, int64_t ldo, int64_t n_own,        \
                                      const float* str, int64_t lds, int64_t n_str, const Params& q,      \
                                      float2* part, const FinArgs& F, hipStream_t st);
CLICA_LP_DECLARE_FIN(0) CLICA_LP_DECLARE_FIN(1) CLICA_LP_DECLARE_FIN(2) CLICA_LP_DECLARE_FIN(3) CLICA_LP_DECLARE_FIN(4)

}  // namespace lp
}  // namespace clica
