// Squared-Euclidean pair sweeps of LpSimCLRLoss (p = 2, pow) on the bf16 matrix cores -- interface of lp_mfma.hip.
#pragma once
#include "common.h"
#include "lp_kernels.h"
#include "lp_guard.h"

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
void set_enabled(int on);                   // process-wide override of CLICA_LP_MFMA: 0 never / 1 pools >= 4 x the local rows (default) / 2 every pool; negative = the environment's setting
bool applies_to_pool(int64_t n_own, int64_t n_pool);      // ... and the pool policy of the setting in force

// `spread` = 64 device words (256 B, zeroed with the workspace) in front of the planes.  The planes of a call are built on a grid step
// D and an origin that were MEASURED BY THE PREVIOUS CALL (round 5: the separate launch that measured them first -- 7 us, latency-bound --
// is gone).  Any origin inside the data is valid and any D with max |x'| / D < 512 keeps the hi x hi product exact (D aims at
// [128, 256); a row in [256, 512) lands on the 2 D grid and only carries a larger remainder); every prep workgroup
// checks the latter for its rows and a violation (the cloud more than doubled since the last call; the very first call) sends
// THIS call to the difference sweeps through the guard below.  No locks, no resets between kernels of one call:
//   [W_RUN_M]      float   largest M any call with a valid grid has seen (diagnostic; kept by the forward sweep's workgroup 0)
//   [W_M64, +1]    u64     (call id << 32) | bits of this call's M          -- tagged atomicMax: an older call's value can never win
//   [W_V64, +1]    u64     (call id << 32) | 1 -- written only by a prep workgroup whose rows violated the grid; valid when its tag equals W_M64's
//   [W_CALL]       u32     call id, advanced by workgroup (0, 0) of the forward sweep (prep has finished; nobody else reads it afterwards)
//   [W_FALLBACKS]  float   number of calls that fell back
//   [W_MAXABS_CUR] float   max |x'| the current grid step derives from;  [W_MAXABS_NEXT]: accumulated by this call's prep for the next one
//   [W_ORIGIN_CUR .. +16)  origin of this call's planes;  [W_ORIGIN_NEXT .. +16): mean of this call's first <= 64 pool rows;
//   [W_ORIGIN_USED .. +16) copy of the origin this call's planes were built on: what the kernels BEHIND the forward sweep read (finalize's
//                          feature planes, the backward sweep) -- the forward sweep's workgroup (0, 0) moves NEXT -> CUR meanwhile
struct Ws { float* spread; void* own_rows; void* pool_rows; void* pool_feat; size_t bytes; };

// The guard.  The expansion's gradient product accumulates terms of size sqrt(M) in fp32, so its error grows ~ sqrt(M) (measured against
// the fp64 oracle: tests/test_gpu_loss.py ..._spread_limit).  Every forward call measures its own M on the device (prep_k); when it exceeds
// spread_limit() the matrix-core sweeps of THIS call return at once and the coordinate-difference sweeps (lp_kernels.h), which are launched
// behind them with the opposite condition, do the work -- same partial formats, same finalize / reduce.  Both sets of launches are always
// in the stream (and in a captured graph), the choice is made per call by the kernels themselves: no host round trip, valid under replay.
float spread_limit();                       // CLICA_LP_MFMA_LIMIT (default kDefaultSpreadLimit) or set_spread_limit
void set_spread_limit(float m);             // <= 0: back to the environment's / default value
// (round 6: 768 -> 700.  Asserted points right at the limit -- tests/test_gpu_loss.py ..._spread_limit -- measured 8.1e-6 at M = 755:
//  inside the limit the raw gradient error must stay <= 8e-6, profiles/r6_loss_spread_curve.json)
constexpr float kDefaultSpreadLimit = 700.f;
Ws carve(void* base, const Plan& P);       // plane buffers inside a caller-provided workspace (256-byte aligned base)

// x' = sqrt(2 log2(e) / tau) (x - origin): row planes of the anchors and of the pool (both sweeps read them).  Also keeps the running maximum
// of M = |x'|^2 / 2 = log2(e)/tau |x - origin|^2 over the pool rows in *w.spread (a float the caller zeroed with the workspace): what the logit's absolute error scales with
void launch_prep(const Plan& P, const Ws& w, const float* own, int64_t ldo, int64_t n_own, const float* pool, int64_t ldp, int64_t n_pool,
                 int n, float kscale, hipStream_t st);
// part[split][row] = (0, sum_j 2^x_ij) -- the partial format of fwd_partial_k<ZMAX>
// (PV, own .. q: the difference sweep of the same call -- lp_kernels.h's plan and parameters -- which the launch runs instead when the guard says so)
void launch_fwd(const Plan& P, const Ws& w, int64_t n_own, float2* part, float limit, const lp::Plan& PV, const float* own, int64_t ldo,
                const float* pool, int64_t ldp, int64_t n_pool, const lp::Params& q, hipStream_t st);
// part[split][row][np] = gradient partials of the symmetric sweep (format of bwd_pairs_k<.., 3, .., FOLD>): 2 sum_j 2^x_ij (u_i + u_j) (a_i - p_j),
// u = C 2^-L from (ownL, ownC) / (poolL, poolC)
// (writes the pool's feature planes first -- they carry the pool rows' u_j -- unless `feat_ready`: on one rank the forward's finalize
// has written them, lp_mfma_dev.h)
void launch_bwd(const Plan& P, const Ws& w, const float* own, int64_t ldo, int64_t n_own, const float* pool, int64_t ldp, int64_t n_pool, int n,
                int np, float kscale, const float* ownL, const float* ownC, const float* poolL, const float* poolC, float* part, bool feat_ready,
                float limit, const lp::Plan& PV, const lp::Params& q, hipStream_t st);

}  // namespace lp2
}  // namespace clica
