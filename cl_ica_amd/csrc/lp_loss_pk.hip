// One exponent kind (CLICA_PK = 0 generic, 1, 2, 3; 4 = dot product) of the pairwise-Lp kernels; compiled four
// times so the 8 padded dims x 3 kernels x 4 kinds instantiate in parallel.
#include "lp_kernels.h"
#include "lp_finalize.h"
#include <stdlib.h>
#ifndef CLICA_PK
#error "compile with -DCLICA_PK=0|1|2|3|4"
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

namespace clica {
namespace lp {

#define LP_FOR_NP(NPV, BODY)                        \
  switch (NPV) {                                    \
    case 4: { constexpr int NP = 4; constexpr int NQ = NP / 2; BODY } break;   \
    case 8: { constexpr int NP = 8; constexpr int NQ = NP / 2; BODY } break;   \
    case 12: if (q.n <= 10) { constexpr int NP = 12; constexpr int NQ = 5; BODY }   /* n = 9, 10: the sixth pair is all padding */ \
             else { constexpr int NP = 12; constexpr int NQ = 6; BODY } break; \
    case 16: { constexpr int NP = 16; constexpr int NQ = NP / 2; BODY } break; \
    case 24: { constexpr int NP = 24; constexpr int NQ = NP / 2; BODY } break; \
    case 32: { constexpr int NP = 32; constexpr int NQ = NP / 2; BODY } break; \
    case 40: { constexpr int NP = 40; constexpr int NQ = NP / 2; BODY } break; \
    case 64: { constexpr int NP = 64; constexpr int NQ = NP / 2; BODY } break; \
    default: break;                                 \
  }

// wide rows (n > 64): one kernel template, instantiated per coordinate-register count
template <int PK, int MODE>
static void launch_wide(const Plan& P, const float* own, int64_t ldo, int64_t n_own, const float* str, int64_t lds, int64_t n_str,
                        const Params& q, const float* ownL, const float* ownC, const float* strL, const float* strC,
                        float2* part, float* part_g, hipStream_t st) {
  dim3 grid((unsigned)P.tiles, (unsigned)P.nsplit), block(THREADS);
  switch (wide_kc(q.n)) {
    case 8: hipLaunchKernelGGL((wide_k<8, PK, MODE>), grid, block, 0, st, own, ldo, n_own, str, lds, n_str, q, ownL, ownC, strL, strC, part, part_g, P.np, P.chunk); break;
    case 16: hipLaunchKernelGGL((wide_k<16, PK, MODE>), grid, block, 0, st, own, ldo, n_own, str, lds, n_str, q, ownL, ownC, strL, strC, part, part_g, P.np, P.chunk); break;
    default: hipLaunchKernelGGL((wide_k<32, PK, MODE>), grid, block, 0, st, own, ldo, n_own, str, lds, n_str, q, ownL, ownC, strL, strC, part, part_g, P.np, P.chunk); break;
  }
}

// part_g != nullptr selects the ROWGRAD variant; its plan must have been made with the backward's
// owners-per-thread (make_plan(..., bwd = true)): it carries the same register load as bwd_pairs_k.
void CAT(launch_fwd_partial_pk, CLICA_PK)(const Plan& P, const float* own, int64_t ldo, int64_t n_own,
                                          const float* str, int64_t lds, int64_t n_str, const Params& q,
                                          float2* part, float* part_g, hipStream_t st) {
  constexpr int PK = CLICA_PK;
  if (P.np > 64) {
    if (part_g) launch_wide<PK, W_FWD_ROWGRAD>(P, own, ldo, n_own, str, lds, n_str, q, nullptr, nullptr, nullptr, nullptr, part, part_g, st);
    else launch_wide<PK, W_FWD>(P, own, ldo, n_own, str, lds, n_str, q, nullptr, nullptr, nullptr, nullptr, part, part_g, st);
    return;
  }
  dim3 grid((unsigned)P.tiles, (unsigned)P.nsplit), block(THREADS);
  LP_FOR_NP(P.np, {
    if (part_g && q.pow)
      hipLaunchKernelGGL((fwd_partial_k<NP, PK, owners_bwd(NP), false, true, NQ>), grid, block, 0, st, own, ldo, n_own, str, lds,
                         n_str, q, part, part_g, P.chunk);
    else if (part_g)
      hipLaunchKernelGGL((fwd_partial_k<NP, PK, owners_bwd(NP), true, true, NQ>), grid, block, 0, st, own, ldo, n_own, str, lds,
                         n_str, q, part, part_g, P.chunk);
    else if (q.pow && (q.train & 1) && PK >= 1 && PK <= 3)
      hipLaunchKernelGGL((fwd_partial_k<NP, PK, owners_fwd(NP), false, false, NQ, (PK >= 1 && PK <= 3)>), grid, block, 0, st, own, ldo, n_own, str, lds,
                         n_str, q, part, part_g, P.chunk);
    else if (q.pow)
      hipLaunchKernelGGL((fwd_partial_k<NP, PK, owners_fwd(NP), false, false, NQ>), grid, block, 0, st, own, ldo, n_own, str, lds,
                         n_str, q, part, part_g, P.chunk);
    else
      hipLaunchKernelGGL((fwd_partial_k<NP, PK, owners_fwd(NP), true, false, NQ>), grid, block, 0, st, own, ldo, n_own, str, lds,
                         n_str, q, part, part_g, P.chunk);
  })
}

// The training sweep with the finalize folded in (lp_finalize.h: fwd_partial_fin_k).  Returns false where that form does not exist (rows
// beyond 16 padded coordinates, the generic exponent, the root form): the caller then launches sweep + fwd_finalize_k.
bool CAT(launch_fwd_partial_fin_pk, CLICA_PK)(const Plan& P, const float* own, int64_t ldo, int64_t n_own,
                                              const float* str, int64_t lds, int64_t n_str, const Params& q,
                                              float2* part, const FinArgs& F, hipStream_t st) {
  constexpr int PK = CLICA_PK;
  if constexpr (PK >= 1 && PK <= 3) {
    if (P.np > 16) return false;
    dim3 grid((unsigned)P.tiles, (unsigned)P.nsplit), block(THREADS);
    bool done = false;
    LP_FOR_NP(P.np, {
      if constexpr (NP <= 16 && HALF * owners_fwd(NP) == FIN_ROWS) {
        if (q.pow && (q.train & 1))        // the training pair's sweep (running maximum known)
          hipLaunchKernelGGL((fwd_partial_fin_k<NP, PK, owners_fwd(NP), false, NQ, true>), grid, block, 0, st, own, ldo, n_own, str, lds,
                             n_str, q, part, P.chunk, F);
        else if (q.pow)                    // generic forward, p-th power (LpSimCLRLoss default)
          hipLaunchKernelGGL((fwd_partial_fin_k<NP, PK, owners_fwd(NP), false, NQ, false>), grid, block, 0, st, own, ldo, n_own, str, lds,
                             n_str, q, part, P.chunk, F);
        else                               // generic forward, the norm itself
          hipLaunchKernelGGL((fwd_partial_fin_k<NP, PK, owners_fwd(NP), true, NQ, false>), grid, block, 0, st, own, ldo, n_own, str, lds,
                             n_str, q, part, P.chunk, F);
        done = true;
      }
    })
    return done;
  }
  return false;
}

void CAT(launch_bwd_pairs_pk, CLICA_PK)(const Plan& P, bool owner_stats, const float* own, int64_t ldo,
                                        int64_t n_own, const float* str, int64_t lds, int64_t n_str,
                                        const Params& q, const float* statL, const float* statC, float* part,
                                        hipStream_t st) {
  constexpr int PK = CLICA_PK;
  if (P.np > 64) {
    if (owner_stats) launch_wide<PK, W_BWD_OWNER>(P, own, ldo, n_own, str, lds, n_str, q, statL, statC, nullptr, nullptr, nullptr, part, st);
    else launch_wide<PK, W_BWD_STREAM>(P, own, ldo, n_own, str, lds, n_str, q, nullptr, nullptr, statL, statC, nullptr, part, st);
    return;
  }
  dim3 grid((unsigned)P.tiles, (unsigned)P.nsplit), block(THREADS);
  LP_FOR_NP(P.np, {
    if (owner_stats && q.pow)
      hipLaunchKernelGGL((bwd_pairs_k<NP, PK, owners_bwd(NP), 1, false, NQ>), grid, block, 0, st, own, ldo, n_own, str,
                         lds, n_str, q, statL, statC, statL, statC, part, P.chunk);
    else if (owner_stats)
      hipLaunchKernelGGL((bwd_pairs_k<NP, PK, owners_bwd(NP), 1, true, NQ>), grid, block, 0, st, own, ldo, n_own, str,
                         lds, n_str, q, statL, statC, statL, statC, part, P.chunk);
    else if (q.pow)
      hipLaunchKernelGGL((bwd_pairs_k<NP, PK, owners_bwd(NP), 2, false, NQ>), grid, block, 0, st, own, ldo, n_own, str,
                         lds, n_str, q, statL, statC, statL, statC, part, P.chunk);
    else
      hipLaunchKernelGGL((bwd_pairs_k<NP, PK, owners_bwd(NP), 2, true, NQ>), grid, block, 0, st, own, ldo, n_own, str,
                         lds, n_str, q, statL, statC, statL, statC, part, P.chunk);
  })
}

// symmetric sweep (stream = the pool the owners belong to): owner AND stream statistics
void CAT(launch_bwd_sym_pk, CLICA_PK)(const Plan& P, const float* own, int64_t ldo, int64_t n_own,
                                      const float* str, int64_t lds, int64_t n_str, const Params& q,
                                      const float* ownL, const float* ownC, const float* strL, const float* strC,
                                      float* part, hipStream_t st) {
  constexpr int PK = CLICA_PK;
  if (P.np > 64) {
    launch_wide<PK, W_BWD_SYM>(P, own, ldo, n_own, str, lds, n_str, q, ownL, ownC, strL, strC, nullptr, part, st);
    return;
  }
  dim3 grid((unsigned)P.tiles, (unsigned)P.nsplit), block(THREADS);
  LP_FOR_NP(P.np, {
    if (q.pow && (q.train & 2) && PK >= 1 && PK <= 3)
      hipLaunchKernelGGL((bwd_pairs_k<NP, PK, owners_bwd(NP), 3, false, NQ, (PK >= 1 && PK <= 3)>), grid, block, 0, st, own, ldo, n_own, str,
                         lds, n_str, q, ownL, ownC, strL, strC, part, P.chunk);
    else if (q.pow)
      hipLaunchKernelGGL((bwd_pairs_k<NP, PK, owners_bwd(NP), 3, false, NQ>), grid, block, 0, st, own, ldo, n_own, str,
                         lds, n_str, q, ownL, ownC, strL, strC, part, P.chunk);
    else
      hipLaunchKernelGGL((bwd_pairs_k<NP, PK, owners_bwd(NP), 3, true, NQ>), grid, block, 0, st, own, ldo, n_own, str,
                         lds, n_str, q, ownL, ownC, strL, strC, part, P.chunk);
  })
}

}  // namespace lp
}  // namespace clica
