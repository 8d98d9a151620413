// Squared-Euclidean pair sweeps of LpSimCLRLoss (p = 2, pow) on the bf16 matrix cores -- interface of lp_mfma.hip.
#pragma once
#include "common.h"

namespace clica {
namespace lp2 {

constexpr int ROWS = 32;             // rows per operand tile (one 32 x 32 x 16 MFMA block side)
constexpr int KSLOTS = 16;           // feature slots of a row: n coordinates + 3 ones + 3 norm pieces (lp_mfma.hip: planes)
constexpr int MAX_N = KSLOTS - 6;
constexpr int STAGE_TILES = 4;       // pool tiles per LDS stage
constexpr int ROWVEC = 3 * 2 * ROWS; // 16-byte vectors of one tile's row planes   [piece][k half][row]      (3 KB)
constexpr int FEATVEC = 3 * 2 * 2 * 32;   // ... of its feature planes              [piece][mfma][k half][feature slot < 32]   (6 KB)

struct Plan {
  int T;                  // anchor tiles per wave
  int64_t groups;         // workgroups along the anchors (4 waves x T tiles x 32 rows each)
  int64_t own_tiles;      // anchor tiles in the plane buffer (padded to whole workgroups)
  int64_t pool_tiles;     // pool tiles in the plane buffers (padded to whole chunks)
  int nsplit, chunk_tiles;
};
Plan make_plan(int64_t n_own, int64_t n_pool);
bool applies(int n, float p, int pow);      // the shapes / settings this path covers (and CLICA_LP_MFMA != 0, or set_enabled)
void set_enabled(int on);                   // process-wide override of CLICA_LP_MFMA: 1 / 0, negative = back to the environment's setting

// `spread` = 64 device floats in front of the planes (zeroed with the workspace):
//   [W_RUN_M] largest M any forward call has seen (diagnostic)          [W_MAXABS] this call's max |x'| (grid step of the hi pieces)
//   [W_STEP_M] THIS call's M -- the device-side guard reads it          [W_FALLBACKS] number of forward calls that fell back (as a float)
//   [W_ORIGIN .. +16) the origin rows are shifted by: the mean of the pool's first <= 64 rows (any point inside the data is valid;
//   the centre keeps M = log2(e)/tau max |x - origin|^2 at about a quarter of what an arbitrary data row gives)
constexpr int W_RUN_M = 0, W_MAXABS = 1, W_STEP_M = 2, W_FALLBACKS = 3, W_ORIGIN = 4;
struct Ws { float* spread; void* own_rows; void* pool_rows; void* pool_feat; size_t bytes; };

// The guard.  The expansion's gradient product accumulates terms of size sqrt(M) in fp32, so its error grows ~ sqrt(M) (measured against
// the fp64 oracle: tests/test_gpu_loss.py ..._spread_limit).  Every forward call measures its own M on the device (prep_k); when it exceeds
// spread_limit() the matrix-core sweeps of THIS call return at once and the coordinate-difference sweeps (lp_kernels.h), which are launched
// behind them with the opposite condition, do the work -- same partial formats, same finalize / reduce.  Both sets of launches are always
// in the stream (and in a captured graph), the choice is made per call by the kernels themselves: no host round trip, valid under replay.
float spread_limit();                       // CLICA_LP_MFMA_LIMIT (default kDefaultSpreadLimit) or set_spread_limit
void set_spread_limit(float m);             // <= 0: back to the environment's / default value
constexpr float kDefaultSpreadLimit = 512.f;
Ws carve(void* base, const Plan& P);       // plane buffers inside a caller-provided workspace (256-byte aligned base)

// x' = sqrt(2 log2(e) / tau) (x - origin): row planes of the anchors and of the pool (both sweeps read them).  Also keeps the running maximum
// of M = |x'|^2 / 2 = log2(e)/tau |x - origin|^2 over the pool rows in *w.spread (a float the caller zeroed with the workspace): what the logit's absolute error scales with
void launch_prep(const Plan& P, const Ws& w, const float* own, int64_t ldo, int64_t n_own, const float* pool, int64_t ldp, int64_t n_pool,
                 int n, float kscale, hipStream_t st);
// part[split][row] = (0, sum_j 2^x_ij) -- the partial format of fwd_partial_k<ZMAX>
void launch_fwd(const Plan& P, const Ws& w, int64_t n_own, float2* part, float limit, hipStream_t st);
// part[split][row][np] = gradient partials of the symmetric sweep (format of bwd_pairs_k<.., 3, .., FOLD>): 2 sum_j 2^x_ij (u_i + u_j) (a_i - p_j),
// u = C 2^-L from (ownL, ownC) / (poolL, poolC)
// (writes the pool's feature planes first -- they carry the pool rows' u_j -- unless `feat_ready`: on one rank the forward's finalize
// has written them, lp_mfma_dev.h)
void launch_bwd(const Plan& P, const Ws& w, const float* own, int64_t ldo, int64_t n_own, const float* pool, int64_t ldp, int64_t n_pool, int n,
                int np, float kscale, const float* ownL, const float* ownC, const float* poolL, const float* poolC, float* part, bool feat_ready,
                float limit, hipStream_t st);

}  // namespace lp2
}  // namespace clica
