/*
 * clica.h -- C ABI of libclica_hip.so: the MI355X (gfx950) implementation of cl-ica's
 * contrastive-training hot path.
 *
 * The reference (brendel-group/cl-ica) is pure Python/PyTorch and has NO FFI of its own
 * (SURVEY.md section 8(b)); this header is the boundary a maintainer would bind with
 * ctypes.  Each entry point cites the reference code it replaces.
 *
 * Conventions
 *   - All matrices are row-major fp32 with an explicit leading dimension `ld*` counted in
 *     ELEMENTS (so strided views such as mu[::2] or z[:, :k] need no copy).
 *   - All pointers are DEVICE pointers unless the name ends in `_host`.
 *   - Buffers are allocated and owned by the caller (PyTorch); the library never frees or
 *     retains them.  `workspace` is caller-provided scratch of at least the size returned
 *     by the matching *_workspace_bytes query; it needs no initialisation unless stated.
 *   - Every call is asynchronous: it enqueues kernels on `stream` (a hipStream_t passed as
 *     void*; NULL = the default stream) and returns.  No call synchronises the device, so
 *     all of them may be captured in a HIP graph.
 *   - Return value: 0 on success, negative on error (CLICA_E_*).  clica_last_error()
 *     returns a thread-local human-readable message.  No C++ exception crosses the ABI.
 */
#ifndef CLICA_H
#define CLICA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLICA_OK 0
#define CLICA_E_INVALID (-1)     /* bad argument (shape, stride, unsupported mode) */
#define CLICA_E_WORKSPACE (-2)   /* workspace too small */
#define CLICA_E_HIP (-3)         /* HIP runtime error at launch */

typedef void* clica_stream_t;    /* hipStream_t */

const char* clica_last_error(void);
/* Test / tuning hook, process-wide (not part of the stable surface): the per-layer Linear entry points and SimCLRLoss choose their body
 * by SHAPE; a test that wants the other product path on a given shape sets it here.  Keys: "skinny" (0: every Linear shape through the MFMA
 * template instead of the vector-ALU kernels for tiny K / N), "gemm_cfg_fwd" / "gemm_cfg_dgrad" / "gemm_cfg_wgrad" (tile configuration id,
 * -1: by shape), "dot_mfma" (0: SimCLRLoss pair sweep at every width), "gemm16_epilogue" (0: clica_linear_split_fwd16 / _dgrad16 keep the
 * generic fused epilogue instead of the one specialised per output combination), "lp_fused_finalize" (0: clica_lp_loss_fwd_train as sweep +
 * finalize launch), "reset" (all defaults).  Unknown key: CLICA_E_INVALID.  The library
 * reads NO tuning switch from the environment (the environment variables it does read are listed in INTEGRATION.md). */
int clica_set_tuning(const char* key, int32_t value);
/* After a FAILED stream capture on `stream` (e.g. a collective that cannot be captured): end the capture if the stream is
 * still capturing, wait for the device and clear the runtime's pending error, so that later launches report their own
 * status.  Harmless when nothing failed. */
int clica_abort_capture(clica_stream_t stream);
int clica_version(void);

/* ------------------------------------------------------------------------------------
 * Lp InfoNCE loss  --  LpSimCLRLoss.loss, /root/reference/losses.py:430-477
 *   neg[i,j] = sum_k |z1[i,k]-z3[j,k]|^p   (then ^(1/p) unless `pow`)     losses.py:447-454
 *   pos[i]   = same for (z1[i], z2[i])                                    losses.py:450
 *   compat:  lse[i] = logsumexp_j(-[neg[i,:], pos[i]]/tau)                losses.py:458-462
 *   default: lse[i] = logsumexp_j(-neg[i,:]/tau) - log(B3)                losses.py:463-465
 *   loss[i]  = 2 (alpha pos[i]/tau + (1-alpha) lse[i])                    losses.py:467
 *   p < 1 uses the reference's eps/transposed branch (losses.py:433-442); it needs B3 == B.
 * The B x B3 matrix is never materialised (tiled online log-sum-exp).
 * ---------------------------------------------------------------------------------- */
typedef struct clica_lp_loss_desc {
  int64_t B;       /* rows of z1 and z2                                  */
  int64_t B3;      /* rows of z3                                         */
  int32_t n;       /* embedding dimension (1..512; register-resident kernels to 64, wide-row kernels beyond) */
  float p;         /* exponent of the norm (1, 2, 3 fast paths; any p>0) */
  float tau;
  float alpha;
  int32_t compat;  /* simclr_compatibility_mode                          */
  int32_t pow;     /* use p-th power of the norm                         */
  int32_t no_eps;  /* 0: LpSimCLRLoss semantics -- p < 1 takes the reference's eps branch (|z1 - z3 + 1e-12|, transposed pair
                      orientation, losses.py:433-442).  1: plain sum_k |d_k|^p for every p > 0, no eps, no transposition: the
                      pair term of UniformityLoss / AlignmentLoss (losses.py:211-221, 231-237)                              */
} clica_lp_loss_desc;

/* scratch needed by clica_lp_loss_fwd / _bwd for this problem size */
int clica_lp_loss_workspace_bytes(const clica_lp_loss_desc* d, size_t* fwd_bytes, size_t* bwd_bytes);

/* Forward.  Outputs (all fp32, device):
 *   loss_i  [B]  per-item loss                         (2nd return of LpSimCLRLoss.loss)
 *   pos_i   [B]  pos[i]/tau                            (its mean is the 3rd return's [0])
 *   lse_i   [B]  row statistic saved for the backward: the RAW log-sum-exp (without -log B3) of the row's logits in LOG2
 *                units, log2(sum_j 2^(log2(e) * logit_ij)) = natural-log lse * log2(e).  Opaque to callers (the natural-log
 *                mean is means[2]); kept in the kernels' own exponent domain so that saturated rows (|lse| ~ 10^3) are not
 *                rounded twice on the way to the backward
 *   means   [3]  mean(loss_i), mean(pos_i), mean(lse as the reference defines it)
 *   rowgrad [B,n] (ld `ldrg`) optional, NULL to skip: sum_j softmax_ij * d neg_ij / d z1_i, accumulated
 *           flash-style inside the same pair sweep.  Handing it to clica_lp_loss_bwd removes the
 *           backward's row pass (one of its two all-pairs sweeps) for ANY upstream gradient.
 * WORKSPACE: ZERO-FILLED before its first use (round 6).  Its first 4 KB are a header: arrival counters of the one-launch form of this
 * call (p in {1, 2, 3}, n <= 14, rowgrad == NULL, B <= 64 448: the last workgroup to deliver a partial of a 64-row tile finishes that
 * tile's rows, the last finisher of the launch writes `means` -- no finalize / means launches, same bits as with them,
 * clica_set_tuning("lp_fused_finalize", 0)).  Every launch leaves the counters zero and no entry point writes the header otherwise, so
 * forward, backward and training-pair calls may share one workspace.
 */
int clica_lp_loss_fwd(const clica_lp_loss_desc* d,
                      const float* z1, int64_t ld1, const float* z2, int64_t ld2,
                      const float* z3, int64_t ld3,
                      float* loss_i, float* pos_i, float* lse_i, float* means,
                      float* rowgrad, int64_t ldrg,
                      void* workspace, size_t workspace_bytes, clica_stream_t stream);

/* Backward (autograd of the forward; recomputes distances, no B x B3 storage).
 * `rowgrad` (optional): the forward's row-gradient output; NULL recomputes it with a row pass.
 * Upstream gradients (device pointers, any may be NULL):
 *   g_mean [1] for means[0] (NULL = 1.0), g_item [B] for loss_i (NULL = 0),
 *   g_pos [1] / g_neg [1] for means[1] / means[2] (NULL = 0).
 * Outputs: dz1 [B,n], dz2 [B,n], dz3 [B3,n] (any may be NULL to skip).  If
 * `accumulate_dz3` != 0 the z3 gradient is ADDED into dz3 (used when z3 aliases z1, i.e.
 * the reference's z3_rec = roll(z1_rec), main_mlp.py:272, up to a row permutation).
 */
int clica_lp_loss_bwd(const clica_lp_loss_desc* d,
                      const float* z1, int64_t ld1, const float* z2, int64_t ld2,
                      const float* z3, int64_t ld3, const float* lse_i,
                      const float* rowgrad, int64_t ldrg,
                      const float* g_mean, const float* g_item, const float* g_pos, const float* g_neg,
                      float* dz1, int64_t ldd1, float* dz2, int64_t ldd2,
                      float* dz3, int64_t ldd3, int32_t accumulate_dz3,
                      void* workspace, size_t workspace_bytes, clica_stream_t stream);

/* Symmetric backward for the reference's training usage z3_rec = roll(z1_rec) (main_mlp.py:272), i.e.
 * "the negatives are all z1 of the (global) batch".  `pool` [B3,n] holds the z1 rows of every rank in any
 * order, this rank's B rows included; `pool_lse` [B3] the raw logsumexp of every pool row (all-gather of
 * lse_i).  Because d(i,j) = d(j,i), one sweep with coefficient C_i w_ij + C_j w_ji yields the complete
 * gradient of the SUM of all ranks' mean losses w.r.t. this rank's z1 rows -- what the generic backward
 * delivers as dz1 + (reduce-scatter of dz3) -- with one pair pass and no gradient exchange.
 * Requirements: p >= 1; every pool row carries the same upstream weight as the local rows (g_mean / g_neg
 * scalars, no per-item upstream).  dz1 [B,n], dz2 [B,n] (dz2 may be NULL).
 */
int clica_lp_loss_bwd_sym(const clica_lp_loss_desc* d,
                          const float* z1, int64_t ld1, const float* z2, int64_t ld2,
                          const float* pool, int64_t ldp, const float* lse_i, const float* pool_lse,
                          const float* g_mean, const float* g_pos, const float* g_neg,
                          float* dz1, int64_t ldd1, float* dz2, int64_t ldd2,
                          void* workspace, size_t workspace_bytes, clica_stream_t stream);
/* Training-step pair (what the fused engine calls): the same mathematics as clica_lp_loss_fwd + clica_lp_loss_bwd_sym
 * with the default upstream gradient d(mean loss) = 1, three launches fewer.  fwd_train also writes the positive-pair
 * part of dz1 / dz2 and keeps the row statistics in the workspace; bwd_sym_train ADDS the pair-sweep gradient to dz1
 * and delivers means[3] = (loss, pos, neg) of the forward.  Both calls share ONE workspace of
 * clica_lp_loss_train_workspace_bytes, untouched in between (only the all-gather of lse_i belongs there).
 * REQUIREMENT (round 4): `pool` must CONTAIN the B rows of z1 bit for bit (it does in the training step: the pool is z1 itself or the
 * all-gather of every rank's z1).  The sweeps use it: every logit is <= 0 and the row's own pool entry gives exactly 0, so the forward
 * sums 2^x without a running maximum and the backward folds the row statistics into one factor per row (one exponential per pair).
 * For negatives that do not include the anchors use clica_lp_loss_fwd / clica_lp_loss_bwd.  CLICA_LP_TRAIN_FAST=0 restores the
 * general-purpose sweeps behind these entry points (A/B switch).
 * p = 2, pow, n <= 10 (BASELINE config 2: main_mlp.py defaults): the two pair sweeps run on the bf16 matrix cores (csrc/lp_mfma.hip:
 * logit = one augmented inner product of bf16 pieces, gradient = a second product against the pool) and z1 / pool must additionally be
 * UNCHANGED between the two calls (fwd_train leaves their operand planes in the workspace).  The expansion |a|^2 + |b|^2 - 2ab behind
 * it has terms of size M = log2(e)/tau max_i |z_i - origin|^2 (origin = the mean of the pool's first 64 rows).  The large part of every
 * term of the logit is computed EXACTLY (hi pieces on a grid common to the launch, accumulated apart) and the loss holds 1e-5 at every
 * spread measured (2e-6 at M = 15 000); the gradient's second product accumulates terms of size sqrt(M) in fp32 and its error grows
 * ~ sqrt(M).  THE GUARD (round 5): every fwd_train call measures its own M on the device; when it exceeds the spread limit (default 700:
 * gradient error < 1e-5 against the fp64 oracle with margin, tests/test_gpu_loss.py ..._spread_limit) the matrix-core sweeps of THAT call
 * return at once and the coordinate-difference sweeps (no such dependence), launched behind them with the opposite condition, do the
 * work.  The decision is made by the kernels per call -- no host round trip, valid inside a replayed graph; the fallback is not a launch
 * of its own: the forward / backward launch carries both sweeps and its workgroups branch on the guard words.  The planes of a call are
 * built on the grid (step D, origin) the PREVIOUS call's prep launch measured; a call whose rows do not fit that grid (the first call on
 * a zeroed workspace, a cloud that grew across a power of two) is sent to the difference sweep by the same guard -- so keep ONE
 * workspace per training loop and do not zero it between steps.
 * WHERE THEY RUN (round 6): by default only against a pool of at least four times the local rows (B3 >= 4 B: the gathered negatives of a
 * data-parallel job, where the pair sweeps are ~40 % of the step).  A single-rank step (B3 = B) runs the coordinate-difference sweeps: at
 * the spread the reference's own training reaches (M ~ 511) the matrix-core gradient measures 6.4e-6 of the 1e-5 budget where the
 * difference sweeps hold 2.1e-6, and the trade buys 13 us of a 350 us step.  CLICA_LP_MFMA / clica_lp_loss_set_matrix_cores: 0 never,
 * 1 the default policy, 2 every pool.
 * ONE LAUNCH (round 6): on the difference sweeps, rows of <= 16 padded coordinates, p in {1, 2, 3}, fwd_train is a single launch -- the last
 * workgroup to deliver a partial of a 64-row owner tile finishes that tile's rows (per-tile arrival counters in the workspace, handed over
 * by agent-scope stores / loads, no fences).  The counters must be ZERO before the first call on a workspace (allocate it zero-filled) and
 * every launch leaves them zero; results are those of the two-launch form bit for bit (clica_set_tuning("lp_fused_finalize", 0)).
 * clica_lp_loss_train_path reports which launches a call makes:
 * *path = 1 matrix cores with the guarded fallback behind them, 0 VALU sweeps only. */
int clica_lp_loss_train_workspace_bytes(const clica_lp_loss_desc* d, size_t* bytes);
int clica_lp_loss_train_path(const clica_lp_loss_desc* d, int32_t* path);
/* Diagnostic: *spread (HOST float) = the largest M any clica_lp_loss_fwd_train call has seen in this workspace since it was zeroed
 * (0 on the VALU path).  Synchronises `stream`; not for the training loop itself -- call it where the loop reads losses anyway. */
int clica_lp_loss_train_spread(const clica_lp_loss_desc* d, const void* workspace, size_t workspace_bytes, float* spread,
                               clica_stream_t stream);
/* The guard's state, HOST floats out4[4] = { largest M so far, M of the last forward call, the limit in force, number of forward calls
 * that fell back to the difference sweeps since the workspace was zeroed }.  Synchronises `stream`. */
int clica_lp_loss_train_guard(const clica_lp_loss_desc* d, const void* workspace, size_t workspace_bytes, float* out4,
                              clica_stream_t stream);
/* Process-wide spread limit of the guard (M above which a call falls back); <= 0: back to CLICA_LP_MFMA_LIMIT / the default.  The limit is
 * a kernel ARGUMENT: a captured graph keeps the one it was captured with. */
int clica_lp_loss_set_spread_limit(float limit);
/* Process-wide policy of the matrix-core sweeps behind the training pair: 0 never (VALU sweeps only), 1 pools of >= 4 x the local rows
 * (the default), 2 every pool; negative: back to CLICA_LP_MFMA's setting.  Takes effect at the next clica_lp_loss_fwd_train call; a workspace sized for the matrix-core path is large
 * enough for the other one; a captured graph keeps the launches it was captured with (re-capture). */
int clica_lp_loss_set_matrix_cores(int32_t on);
int clica_lp_loss_fwd_train(const clica_lp_loss_desc* d,
                            const float* z1, int64_t ld1, const float* z2, int64_t ld2, const float* pool, int64_t ldp,
                            float* loss_i, float* pos_i, float* lse_i,
                            float* dz1, int64_t ldd1, float* dz2, int64_t ldd2,
                            void* workspace, size_t workspace_bytes, clica_stream_t stream);
int clica_lp_loss_bwd_sym_train(const clica_lp_loss_desc* d,
                                const float* z1, int64_t ld1, const float* pool, int64_t ldp,
                                const float* lse_i, const float* pool_lse,
                                float* dz1, int64_t ldd1, float* means, int32_t* tick_counter /* NULL, or a device counter to advance by 1 */,
                                void* workspace, size_t workspace_bytes, clica_stream_t stream);
/* The same call WITHOUT its closing reduction launch: the symmetric sweep's per-split partials stay in the workspace and `parts` (host
 * struct, filled here) tells the consumer what that launch would have done -- dz1[i][k] += sum over the splits of part[split][i][k], the
 * forward's three means from the row blocks' sums, the counter tick.  clica_mlp_dgrad_split_tail takes it and does all three in its
 * prologue (the training step then has no launch between the pair sweep and the backward chain; same sums in the same order, bit for
 * bit).  Which sweep wrote the partials -- matrix cores or, beyond the guard's limit, coordinate differences -- is read from the guard
 * words on the device, as the reduction launch does.  Reference: the backward of losses.py:430-477 through main_mlp.py:274-285. */
typedef struct clica_lp_dy_parts {
  const float* part;                 /* [nsplit][rows][np] */
  int32_t nsplit, nsplit_alt;        /* splits of the sweep that ran by default / of the other one (guard) */
  int32_t np, n;                     /* padded / real row width */
  int64_t rows;
  const float* guard_words;          /* NULL: no guard, nsplit is final */
  float guard_limit;
  const float* blocksums; int32_t nblocks; float inv_count; float* means;      /* the forward's loss / pos / neg means */
  int32_t* tick;                     /* device counter to advance by 1, or NULL */
} clica_lp_dy_parts;
int clica_lp_loss_bwd_sym_train_parts(const clica_lp_loss_desc* d,
                                      const float* z1, int64_t ld1, const float* pool, int64_t ldp,
                                      const float* lse_i, const float* pool_lse, float* means, int32_t* tick_counter,
                                      void* workspace, size_t workspace_bytes, clica_lp_dy_parts* parts, clica_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Dot-product InfoNCE  --  SimCLRLoss.loss, /root/reference/losses.py:177-202
 *   (optional row L2-normalisation is done by the caller-visible wrapper kernels below)
 *   neg = z1 z3^T, pos = <z1,z2>, lse = logsumexp([neg,pos]/tau),
 *   loss = 2(alpha(-pos/tau) + (1-alpha) lse)
 * n < 96: the tiled pair sweep (vector ALU, no B x B3 matrix).  n >= 96: the three contractions
 * (z1 z3^T, W z3, W^T z1) run on the fp32 MFMA GEMMs with the logit matrix in the workspace
 * (4 B B3 bytes more; clica_dot_loss_workspace_bytes accounts for it; CLICA_DOT_MFMA=0 disables).
 * ---------------------------------------------------------------------------------- */
typedef struct clica_dot_loss_desc {
  int64_t B, B3;
  int32_t n;
  float tau, alpha;
  int32_t normalize;   /* losses.py:180-185 */
} clica_dot_loss_desc;

int clica_dot_loss_workspace_bytes(const clica_dot_loss_desc* d, size_t* fwd_bytes, size_t* bwd_bytes);
int clica_dot_loss_fwd(const clica_dot_loss_desc* d,
                       const float* z1, int64_t ld1, const float* z2, int64_t ld2,
                       const float* z3, int64_t ld3,
                       float* loss_i, float* pos_i, float* lse_i, float* means,
                       float* rowgrad, int64_t ldrg,
                       void* workspace, size_t workspace_bytes, clica_stream_t stream);
int clica_dot_loss_bwd(const clica_dot_loss_desc* d,
                       const float* z1, int64_t ld1, const float* z2, int64_t ld2,
                       const float* z3, int64_t ld3, const float* lse_i,
                       const float* rowgrad, int64_t ldrg,
                       const float* g_mean, const float* g_item, const float* g_pos, const float* g_neg,
                       float* dz1, int64_t ldd1, float* dz2, int64_t ldd2,
                       float* dz3, int64_t ldd3, int32_t accumulate_dz3,
                       void* workspace, size_t workspace_bytes, clica_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Fused Linear (+bias) (+LeakyReLU)  --  the nn.Sequential get_mlp builds,
 * /root/reference/encoders.py:36-48 (nn.Linear + nn.LeakyReLU(0.01)); fp32 MFMA
 * (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate).
 *
 *   fwd :  Y[M,N]  = act(X[M,K] W[N,K]^T + bias[N])          act = LeakyReLU(slope) or identity
 *   dgrad: dX[M,K] = (dY[M,N] W[N,K]) * act'(Xact[M,K])      Xact = the saved OUTPUT of the
 *          previous layer's activation (sign(Xact) == sign(pre-activation) for slope > 0);
 *          pass Xact = NULL for no activation derivative
 *   wgrad: dW[N,K] (+)= dY[M,N]^T X[M,K];  db[N] (+)= sum_m dY[m,:]
 * ---------------------------------------------------------------------------------- */
int clica_linear_fwd(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias,
                     float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                     int32_t leaky, float slope, clica_stream_t stream);
int clica_linear_dgrad(const float* dY, int64_t lddy, const float* W, int64_t ldw,
                       const float* Xact, int64_t ldxa, float slope,
                       float* dX, int64_t lddx, int64_t M, int64_t N, int64_t K,
                       clica_stream_t stream);
int clica_linear_wgrad_workspace_bytes(int64_t M, int64_t N, int64_t K, size_t* bytes);
/* Which kernel instance a launch of this shape uses (host-only query for profiling / benches):
 * op 0 = fwd, 1 = dgrad, 2 = wgrad; returns the workgroup tile, waves per workgroup and, for wgrad,
 * the number of contraction splits.  The kernel symbol is gemm_k<tile_m, tile_n, ...>. */
int clica_linear_plan(int32_t op, int64_t M, int64_t N, int64_t K, int32_t* tile_m, int32_t* tile_n,
                      int32_t* waves, int32_t* splits);
int clica_linear_wgrad(const float* dY, int64_t lddy, const float* X, int64_t ldx,
                       float* dW, int64_t lddw, float* db, int64_t M, int64_t N, int64_t K,
                       int32_t accumulate, void* workspace, size_t workspace_bytes,
                       clica_stream_t stream);

/* Whole-stack forward in one launch (csrc/fused_mlp.hip): layer l computes
 *   out[l] = act(in_l W[l]^T + bias[l]),  in_0 = X, in_l = out[l-1];  act = LeakyReLU(slope) on all but the last
 * with the activation panel resident in LDS (48 rows per workgroup); every out[l] is also written to HBM
 * (the backward needs the saved activations).  All widths must be <= 512 and n_layers <= 8 (the n = 10
 * encoder of main_mlp.py:297-307); otherwise CLICA_E_INVALID -- call clica_linear_fwd per layer instead.
 * W / ldw / bias / out / ldo / N / K are HOST arrays of n_layers entries (the pointers in them are device
 * pointers).
 * `packed` (optional, NULL = read W directly): the same weights in MFMA fragment order, produced by
 * clica_mlp_pack into a buffer of clica_mlp_pack_bytes; a wave's weight fetch is then one contiguous 1 KB
 * request instead of sixteen 64-byte pieces (re-pack after every optimizer step: ~3.4 MB, a few us).  The
 * padding entries (widths that are not multiples of 16 / 32) are written as zeros by every pack. */
int clica_mlp_fwd(const float* X, int64_t ldx, int64_t M, int32_t n_layers,
                  const float* const* W, const int64_t* ldw, const float* const* bias,
                  float* const* out, const int64_t* ldo, const int32_t* N, const int32_t* K,
                  const float* packed, uint64_t* const* signmask, float slope, clica_stream_t stream);
/* Same stack with the mixing network g in the prologue: Z = [M, n] latents, x = g(Z) (clica_mixing_fwd's arithmetic,
 * n = K[0] <= 16) is computed per 48-row panel on chip, fed to layer 0 and also written to x_out (layer 0's weight
 * gradient reads it): one launch less per step. */
int clica_mlp_fwd_mixed(const float* Z, int64_t ldz, int64_t M, const float* mix_W, int32_t mix_layers, float mix_slope,
                        float* x_out, int64_t ldxo, int32_t n_layers,
                        const float* const* W, const int64_t* ldw, const float* const* bias,
                        float* const* out, const int64_t* ldo, const int32_t* N, const int32_t* K,
                        const float* packed, uint64_t* const* signmask, float slope, clica_stream_t stream);
/* signmask (array of n_layers pointers, or NULL; entries may be NULL): per layer an opaque buffer of
 * clica_mlp_signmask_bytes(M) bytes that receives one bit per output element, (out > 0), in the kernel's
 * accumulator order.  clica_mlp_dgrad takes it in place of re-reading the saved activation: 8 bytes per lane
 * instead of 48 strided 4-byte loads in the epilogue of every link. */
int clica_mlp_signmask_bytes(int64_t M, size_t* bytes);
int clica_mlp_pack_bytes(int32_t n_layers, const int32_t* N, const int32_t* K, int32_t transpose, size_t* bytes);
int clica_mlp_pack(int32_t n_layers, const float* const* W, const int64_t* ldw, const int32_t* N, const int32_t* K,
                   int32_t transpose, float* packed, clica_stream_t stream);
/* One launch that writes BOTH layouts a training step needs: packed_fwd = clica_mlp_pack(transpose = 0) of layers
 * 0..L-1 and packed_bwd = clica_mlp_pack(transpose = 1) of layers L-1..1 (the chain order of clica_mlp_dgrad). */
int clica_mlp_pack_both(int32_t n_layers, const float* const* W, const int64_t* ldw, const int32_t* N, const int32_t* K,
                        float* packed_fwd, float* packed_bwd, clica_stream_t stream);
/* Backward data chain of the same stack in one launch (dZ panel resident in LDS):
 *   out[j] = (in_j B_j) * LeakyReLU'(act[j]),  in_0 = dY, in_j = out[j-1],  j = 0..n_links-1
 * B_j only in fragment order: `packed` = clica_mlp_pack(..., transpose = 1, ...) of the encoder layers in CHAIN
 * order (layer L-1 first, down to layer 1); N[j] / K[j] = output / contraction width of link j (= K_l / N_l of
 * the layer it differentiates); act[j] = the saved activation that fed that layer (sign -> act'), NULL = none;
 * signmask[j] (array or NULL) = the sign bits clica_mlp_fwd stored for that same activation: used instead of act[j];
 * out[j] = dZ of the layer below, written to HBM for the weight-gradient GEMMs.  Widths <= 512. */
int clica_mlp_dgrad(const float* dY, int64_t lddy, int64_t M, int32_t n_links,
                    const int32_t* N, const int32_t* K, const float* packed,
                    const float* const* act, const int64_t* ldact, const uint64_t* const* signmask,
                    float* const* out, const int64_t* ldo, float slope, clica_stream_t stream);
/* ---- split-bf16 arithmetic for the whole-stack kernels and the weight gradients (same results to fp32 rounding level) ----
 * Both fp32 operands of every Linear are split exactly into three bf16 pieces and the six piece products of order
 * <= 2 are accumulated in fp32 on the bf16 matrix cores (csrc/fused_mlp.hip, mlp_split_k): measured max error vs fp64
 * 8.6e-7 of max|y| on a 500 x 500 layer (fp32-MFMA path: 1.0e-6), 0.375 of the matrix-core time.
 * packed_split buffers come from clica_mlp_pack_split[_both] (clica_mlp_pack_split_bytes bytes; 6 B per weight).  The
 * sign-bit buffers have the same size as the fp32 kernels' but a different bit order: use the pair fwd_split /
 * dgrad_split together.  Layer outputs (saved activations / dZ) are written as fp32 (`out[l]`, as by clica_mlp_fwd /
 * clica_mlp_dgrad) and / or as three bf16 PLANES in the operand format of clica_mlp_wgrad_split (`planes[l]`, a buffer of
 * clica_mlp_planes_bytes(M, N_l, ones) bytes; ones = 1 for the forward's activations -- they carry the constant-1 feature
 * that makes the weight-gradient GEMM return db -- and 0 for the backward chain's dZ).  `planes` may be NULL (no planes
 * at all); out[l] may be NULL when planes[l] is given (a hidden activation / dZ that only the weight gradients read).
 * mix_W may be NULL (then X is the stack's input and x_out is ignored). */
int clica_mlp_planes_bytes(int64_t M, int32_t width, int32_t ones_column, size_t* bytes);
int clica_mlp_pack_split_bytes(int32_t n_layers, const int32_t* N, const int32_t* K, int32_t transpose, size_t* bytes);
int clica_mlp_pack_split(int32_t n_layers, const float* const* W, const int64_t* ldw, const int32_t* N, const int32_t* K,
                         int32_t transpose, void* packed, clica_stream_t stream);
int clica_mlp_pack_split_both(int32_t n_layers, const float* const* W, const int64_t* ldw, const int32_t* N, const int32_t* K,
                              void* packed_fwd, void* packed_bwd, clica_stream_t stream);
int clica_mlp_fwd_split(const float* X, int64_t ldx, int64_t M, const float* mix_W, int32_t mix_layers, float mix_slope,
                        float* x_out, int64_t ldxo, int32_t n_layers, const float* const* bias,
                        float* const* out, const int64_t* ldo, const int32_t* N, const int32_t* K,
                        const void* packed_split, uint64_t* const* signmask, void* const* planes, float slope,
                        clica_stream_t stream);
int clica_mlp_dgrad_split(const float* dY, int64_t lddy, int64_t M, int32_t n_links, const int32_t* N, const int32_t* K,
                          const void* packed_split, const uint64_t* const* signmask,
                          float* const* out, const int64_t* ldo, void* const* planes, float slope, clica_stream_t stream);
/* clica_mlp_wgrad in the split-bf16 arithmetic (csrc/wgrad_split.hip): the autograd of nn.Linear.weight / .bias over the whole
 * encoder (/root/reference/encoders.py:36-48, main_mlp.py:282) with both operands of every MFMA-sized layer taken from the
 * bf16 plane copies the split kernels above wrote (dZ_planes[l], X_planes[l]; X_planes[l] with the constant-1 feature).
 * Layers with a tiny dimension (clica_mlp_wgrad_split_kind -> 1: min(N, K) <= 16 and max(N, K) <= 128, the n-wide first /
 * last encoder layer) run the fp32 VALU kernel of clica_mlp_wgrad on the fp32 operands dZ[l] / X[l]; the other layers'
 * fp32 pointers may be NULL.  Same slab workspace scheme and deterministic reduction as clica_mlp_wgrad. */
int clica_mlp_wgrad_split_kind(int32_t N, int32_t K, int32_t* kind);
/* fp32 [M, width] (ld `ldx`) -> the bf16 plane copy clica_mlp_wgrad_split reads (buffer of clica_mlp_planes_bytes(M, width, ones)
 * bytes): for operands that no split whole-stack kernel wrote -- the activations / gradients of the per-layer kernels of wide
 * encoders (BASELINE config 3's 2000-wide layers), the encoder input, the loss gradient.  HBM-bound (4 B in, 6 B out per element). */
int clica_mlp_planes_from_f32(const float* X, int64_t ldx, int64_t M, int32_t width, int32_t ones_column, void* planes_out,
                              clica_stream_t stream);
/* Wide layers (a width beyond the 512 the whole-stack kernel holds on chip: BASELINE config 3, main_mlp.py:297-307 with --n 40):
 * forward and data gradient of ONE nn.Linear (+ LeakyReLU, encoders.py:36-48) in the same split-bf16 arithmetic, on the bodies of
 * the weight-gradient GEMM.  Operands are bf16 planes (clica_mlp_planes_bytes) of the TRANSPOSED tensors -- "T-planes of X" =
 * planes of X^T: rows = feature index of X, features = batch row -- so that the contraction runs along plane rows:
 *   clica_linear_split_fwd    Y = leaky(X W^T + b):  xT = T-planes of X [M, K], wT = T-planes of W [N, K] (planes of W^T);
 *                             writes any of: yT (T-planes of Y: next layer's xT / the sign gate of its backward), yN (planes of Y,
 *                             rows = batch row: X operand of the next layer's clica_mlp_wgrad_split; yN_ones = that buffer was
 *                             allocated with the ones column, which this call leaves untouched), Y (fp32).
 *   clica_linear_split_dgrad  dX = (dZ W) * leaky'(act):  dzT = T-planes of dZ [M, N], wN = planes of W [N, K] (rows = N),
 *                             actT = T-planes of the layer's INPUT activation [M, K] (sign gate; NULL: no gate);
 *                             writes any of dxT (T-planes of dX), dxN (planes of dX: dZ operand of the previous layer's
 *                             weight gradient), dX (fp32).
 *   clica_mlp_planes_from_f32_t  fp32 X [M, width] -> T-planes of X (buffer of clica_mlp_planes_bytes(width, M, 0) bytes).
 * Plane buffers must be zero-initialised once (padding rows / features are never written and must read as zero). */
int clica_mlp_planes_from_f32_t(const float* X, int64_t ldx, int64_t M, int32_t width, void* planes_out, clica_stream_t stream);
int clica_linear_split_fwd(const void* xT_planes, const void* wT_planes, const float* bias, int64_t M, int32_t N, int32_t K,
                           int32_t leaky, float slope, void* yT_planes, void* yN_planes, int32_t yN_ones,
                           float* Y, int64_t ldy, clica_stream_t stream);
int clica_linear_split_dgrad(const void* dzT_planes, const void* wN_planes, const void* actT_planes, float slope,
                             int64_t M, int32_t N, int32_t K, void* dxT_planes, void* dxN_planes,
                             float* dX, int64_t lddx, clica_stream_t stream);
int clica_mlp_wgrad_split_workspace_bytes(int64_t M, int32_t n_layers, const int32_t* N, const int32_t* K, size_t* bytes);
int clica_mlp_wgrad_split(int64_t M, int32_t n_layers, const void* const* dZ_planes, const void* const* X_planes,
                          const float* const* dZ, const int64_t* lddz, const float* const* X, const int64_t* ldx,
                          float* const* dW, const int64_t* lddw, float* const* db, const int32_t* N, const int32_t* K,
                          int32_t accumulate, void* workspace, size_t workspace_bytes, clica_stream_t stream);

/* ---- f16x2 arithmetic (round 5): the same whole-encoder kernels with TWO fp16 pieces per operand --------------------------------
 * Replaces nothing in the reference: it is a second fp32 EMULATION next to the bf16x3 one above, for encoders.py:36-48's Linear
 * stack.  v s = hi + lo with hi = RN_f16(v s), lo = RN_f16(v s - hi) (22 significand bits) and the three products hi.hi, hi.lo,
 * lo.hi accumulated in fp32 (v_mfma_f32_*_f16): half the matrix work and 4 instead of 6 bytes per operand element.  fp16's 5-bit
 * exponent needs a power-of-two scale s PER TENSOR: a caller-owned device `state` (clica_split16_state_bytes bytes, 16-byte aligned,
 * initialised once by clica_split16_state_init) holds, per tensor of one encoder, the scale in force and the running maximum the
 * producers record; clica_split16_update (one tiny launch per training step, after the step's last producer) turns the maxima into
 * the next step's scales (largest scaled magnitude in [256, 512)).  So a launch uses the scales of the PREVIOUS step: run L + 1
 * un-applied passes (L = number of layers: each pass settles at least one more link of the gradient chain) after initialisation /
 * after parameters were set from outside so that the scales match the data (the engine does: ContrastiveTrainer.calibrate_scales).  A scaled magnitude beyond 32768 (a tensor grew > 64 x between consecutive steps) raises bit 0 of the state's sticky
 * flags -- results of that launch are not to be trusted; clica_split16_read (synchronises) reports it.  Measured error: at the
 * native fp32-MFMA kernels' level (tests: every engine family in this arithmetic at 1e-5).
 * Order of one training step on a state: clica_mlp_pack_split16_both -> clica_mlp_fwd_split16 -> clica_mlp_dgrad_split16 ->
 * clica_mlp_wgrad_split16 -> clica_split16_update.  Plane buffers hold 2 KB per unit (clica_mlp_planes16_bytes); everything else
 * (fragment order, sign bits, the constant-1 feature -- stored as 1.0; the weight-gradient kernel takes X's scale out of dW only --, argument meaning) is as for the
 * bf16x3 entry points of the same name without "16".  Needs a LeakyReLU slope in (0, 1). */
int clica_split16_state_bytes(size_t* bytes);
int clica_split16_state_init(void* state, clica_stream_t stream);
int clica_split16_update(void* state, int32_t n_layers, clica_stream_t stream);
/* HOST outputs (each may be NULL): sticky flags, number of updates, the scales in force for the activations (forward order, [0] =
 * encoder input), for the chain gradients (chain order, [0] = d loss / d last pre-activation) and for the weights, and the activation /
 * gradient scales the LAST step ran with (what its plane copies are scaled by); 9 floats each.  Synchronises `stream`. */
int clica_split16_read(const void* state, int32_t* flags, int32_t* updates, float* scales_a, float* scales_d, float* scales_w,
                       float* last_scales_a, float* last_scales_d, clica_stream_t stream);
int clica_split16_clear_flags(void* state, clica_stream_t stream);
/* THE GUARD of the f16x2 arithmetic (round 6; the reference's fp32 encoder, encoders.py:36-48, has no input it silently mis-computes, so
 * neither may this one).  Every producer checks the fp32 magnitudes it cuts to fp16 against the scale in force; a scaled magnitude beyond
 * 2^15 (or a non-finite value) POISONS the step on the device.  The launch that applies the optimizer for that step -- clica_adam_step_s16,
 * clica_mlp_wgrad_split_adam -- then leaves parameters and moments untouched; the scale update riding in it counts the step as skipped,
 * raises flag bit 1 and takes the step / RNG counter back by one, so a replayed step graph redoes the SAME batch on the scales the
 * withheld step measured (one more link of the chain settles per redo at worst).  No host round trip: valid inside graph replay.
 * flags: bit 0 an overflow was seen (sticky), bit 1 a step was withheld (sticky), bit 2 the update saw an overflow no producer announced.
 * clica_split16_guard: HOST outputs (each may be NULL) flags, number of withheld steps, number of updates, and whether the step whose
 * producers ran last is poisoned; synchronises `stream`.
 * Data parallel: all ranks must take the same decision.  clica_split16_set_dp_poison(state, slot) makes the optimizer launches read the
 * verdict from the device float `slot` (> 0: poisoned) instead of the rank's own words; clica_split16_poison_export(state, slot) writes
 * this rank's verdict (0 / 1) there -- the caller puts `slot` behind its gradient arena and lets it ride through the gradient
 * all-reduce (sum).  slot == NULL restores the single-rank behaviour. */
int clica_split16_guard(const void* state, int32_t* flags, int32_t* skipped, int32_t* updates, int32_t* poisoned, clica_stream_t stream);
int clica_split16_set_dp_poison(void* state, const float* slot, clica_stream_t stream);
int clica_split16_poison_export(const void* state, float* slot, clica_stream_t stream);
int clica_mlp_planes16_bytes(int64_t M, int32_t width, int32_t ones_column, size_t* bytes);
int clica_mlp_pack_split16_bytes(int32_t n_layers, const int32_t* N, const int32_t* K, int32_t transpose, size_t* bytes);
int clica_mlp_pack_split16_both(int32_t n_layers, const float* const* W, const int64_t* ldw, const int32_t* N, const int32_t* K,
                                void* packed_fwd, void* packed_bwd, void* state, clica_stream_t stream);
int clica_mlp_fwd_split16(const float* X, int64_t ldx, int64_t M, const float* mix_W, int32_t mix_layers, float mix_slope,
                          float* x_out, int64_t ldxo, int32_t n_layers, const float* const* bias,
                          float* const* out, const int64_t* ldo, const int32_t* N, const int32_t* K,
                          const void* packed_split16, uint64_t* const* signmask, void* const* planes, float slope,
                          void* state, clica_stream_t stream);
int clica_mlp_dgrad_split16(const float* dY, int64_t lddy, int64_t M, int32_t n_links, const int32_t* N, const int32_t* K,
                            const void* packed_split16, const uint64_t* const* signmask,
                            float* const* out, const int64_t* ldo, void* const* planes, float slope, void* state, clica_stream_t stream);
/* a_index[l] / d_index[l]: positions of layer l's input activation / of dZ_l in the state's activation / chain-gradient scale
 * arrays (an L-layer encoder passed whole: a_index[l] = l, d_index[l] = L - 1 - l). */
int clica_mlp_wgrad_split16(int64_t M, int32_t n_layers, const void* const* dZ_planes, const void* const* X_planes,
                            const float* const* dZ, const int64_t* lddz, const float* const* X, const int64_t* ldx,
                            float* const* dW, const int64_t* lddw, float* const* db, const int32_t* N, const int32_t* K,
                            int32_t accumulate, const void* state, const int32_t* a_index, const int32_t* d_index,
                            void* workspace, size_t workspace_bytes, clica_stream_t stream);

/* Round 5, the training step's small launches: the backward chain can leave the weight-gradient slabs of the encoder's n-wide FIRST
 * and LAST layer itself (every chain workgroup adds the products of its 48 rows behind the last link, fp32 on the vector ALU -- the
 * arithmetic of the tiny-dimension kernel that clica_mlp_wgrad_split* would otherwise launch).  `tail` names what that needs beyond
 * the chain's own arguments: the fp32 input activation of the last layer, the encoder input, the encoder's shapes in forward order
 * (n_layers = n_links + 1) and the workspace the FOLLOWING clica_mlp_wgrad_split_adam(..., tail_slabs = 1, ...) call is given (the
 * slabs go where that call's plan expects them).  The last link must write its fp32 output.  clica_mlp_chain_tail_supported: first
 * and last layer of the tiny-dimension kind with K[0] <= 15, N[0] <= 128, N[L-1] <= 16, K[L-1] <= 127 (get_mlp's n -> 10 n ... 10 n -> n
 * up to n = 12; encoders.py:36-48). */
typedef struct clica_chain_tail {
  const float* a_last; int64_t lda;      /* [M][K[L-1]] input of the last layer */
  const float* x; int64_t ldx;           /* [M][K[0]]   encoder input */
  int32_t n_layers; const int32_t* N; const int32_t* K;
  void* wgrad_workspace; size_t wgrad_workspace_bytes;
  const clica_lp_dy_parts* dy_parts;     /* NULL, or: dY (which must be writable) still lacks the pair sweep's partials -- see clica_lp_dy_parts */
} clica_chain_tail;
int clica_mlp_chain_tail_supported(int32_t n_layers, const int32_t* N, const int32_t* K, int32_t* supported);
/* state == NULL: bf16x3 (clica_mlp_dgrad_split), else f16x2 (clica_mlp_dgrad_split16) */
int clica_mlp_dgrad_split_tail(const float* dY, int64_t lddy, int64_t M, int32_t n_links, const int32_t* N, const int32_t* K,
                               const void* packed_split, const uint64_t* const* signmask,
                               float* const* out, const int64_t* ldo, void* const* planes, float slope, void* state,
                               const clica_chain_tail* tail, clica_stream_t stream);

/* clica_mlp_wgrad_split16 with `tail_slabs` as in clica_mlp_wgrad_split_adam (the chain call in front was clica_mlp_dgrad_split_tail on the
 * SAME workspace): the n-wide first / last layer's slabs are already there, no tiny-dimension launch.  No optimizer: what the drop-in
 * encoder's autograd backward calls. */
int clica_mlp_wgrad_split16_tail(int64_t M, int32_t n_layers, const void* const* dZ_planes, const void* const* X_planes,
                                 const float* const* dZ, const int64_t* lddz, const float* const* X, const int64_t* ldx,
                                 float* const* dW, const int64_t* lddw, float* const* db, const int32_t* N, const int32_t* K,
                                 int32_t accumulate, const void* state, const int32_t* a_index, const int32_t* d_index,
                                 int32_t tail_slabs, void* workspace, size_t workspace_bytes, clica_stream_t stream);

/* Weight gradients + optimizer in one call (round 5: the N = 1 training step has no optimizer launch of its own).  The reduction that
 * ends clica_mlp_wgrad_split / _split16 applies torch.optim.Adam's update (main_mlp.py:312, as clica_adam_step_at) to every element
 * it has just reduced; dW[l] (contiguous: lddw[l] = K[l]) and db[l] must be views of ONE gradient arena `grad` that this call covers
 * completely (alignment padding aside), param / exp_avg / exp_avg_sq are arenas of the same layout.  The gradients are still written.
 * split16_state != NULL: the f16x2 scale update (clica_split16_update(state, n_layers)) rides in front of the same launch.
 * `state` (NULL: bf16x3 plane copies, else f16x2), a_index, d_index as for clica_mlp_wgrad_split16.  tail_slabs != 0: the slabs of the
 * first and last layer are already in the workspace (clica_mlp_dgrad_split_tail of the same step): no tiny-dimension launch. */
typedef struct clica_adam_desc {
  float* param; float* grad; float* exp_avg; float* exp_avg_sq; int64_t count;
  float lr, beta1, beta2, eps, grad_scale;
  const int32_t* step_dev; int32_t t_offset;      /* update number = *step_dev + t_offset (0 or 1), as clica_adam_step_at */
  void* split16_state; int32_t n_layers;
} clica_adam_desc;
int clica_mlp_wgrad_split_adam(int64_t M, int32_t n_layers, const void* const* dZ_planes, const void* const* X_planes,
                               const float* const* dZ, const int64_t* lddz, const float* const* X, const int64_t* ldx,
                               float* const* dW, const int64_t* lddw, float* const* db, const int32_t* N, const int32_t* K,
                               const void* state, const int32_t* a_index, const int32_t* d_index, const clica_adam_desc* adam,
                               int32_t tail_slabs, void* workspace, size_t workspace_bytes, clica_stream_t stream);

/* f16x2 variants of the per-layer entry points (BASELINE config 3's wide chain).  A tensor is named by (family, index) in the state:
 * family 0 = activations (index l = the INPUT of layer l), 1 = gradients (index l = dZ_l), 2 = weights (index l); here the caller passes
 * a_index[l] = l and d_index[l] = l to clica_mlp_wgrad_split16.  The constant-1 feature of an N-plane buffer is 1.0 (not the scale). */
int clica_mlp_planes16_from_f32(const float* X, int64_t ldx, int64_t M, int32_t width, int32_t ones_column, void* planes_out,
                                void* state, int32_t family, int32_t index, clica_stream_t stream);
int clica_mlp_planes16_from_f32_t(const float* X, int64_t ldx, int64_t M, int32_t width, void* planes_out,
                                  void* state, int32_t family, int32_t index, clica_stream_t stream);
int clica_linear_split_fwd16(const void* xT_planes, const void* wT_planes, const float* bias, int64_t M, int32_t N, int32_t K,
                             int32_t leaky, float slope, void* yT_planes, void* yN_planes, int32_t yN_ones,
                             float* Y, int64_t ldy, void* state, int32_t layer, clica_stream_t stream);
int clica_linear_split_dgrad16(const void* dzT_planes, const void* wN_planes, const void* actT_planes, float slope,
                               int64_t M, int32_t N, int32_t K, void* dxT_planes, void* dxN_planes,
                               float* dX, int64_t lddx, void* state, int32_t layer, clica_stream_t stream);

/* Weight/bias gradients of ALL layers in two launches (one grouped split-K GEMM over equal-length work items
 * + one grouped deterministic slab reduction):  dW[l] = dZ[l]^T X[l]  ([N_l, K_l]),  db[l] = column sums of dZ[l]
 * (db[l] may be NULL).  dZ[l] = [M, N_l] gradient at layer l's pre-activation, X[l] = [M, K_l] its input. */
int clica_mlp_wgrad_workspace_bytes(int64_t M, int32_t n_layers, const int32_t* N, const int32_t* K, size_t* bytes);
int clica_mlp_wgrad(int64_t M, int32_t n_layers, const float* const* dZ, const int64_t* lddz,
                    const float* const* X, const int64_t* ldx, float* const* dW, const int64_t* lddw,
                    float* const* db, const int32_t* N, const int32_t* K, int32_t accumulate,
                    void* workspace, size_t workspace_bytes, clica_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Output heads  --  RescaleLayer (mode "eq") layers.py:63-66, SoftclipLayer layers.py:87-91
 * ---------------------------------------------------------------------------------- */
int clica_rescale_fwd(const float* X, int64_t ldx, const float* r /*[1]*/, float* Y, int64_t ldy,
                      float* inv_norm /*[M] saved*/, int64_t M, int32_t n, clica_stream_t stream);
int clica_rescale_bwd(const float* X, int64_t ldx, const float* r, const float* inv_norm,
                      const float* dY, int64_t lddy, float* dX, int64_t lddx,
                      float* dr_partial /*[ceil(M/256)] or NULL*/, int64_t M, int32_t n,
                      clica_stream_t stream);
int clica_softclip_fwd(const float* X, int64_t ldx, const float* bound /*[n]*/, float* Y, int64_t ldy,
                       int64_t M, int32_t n, clica_stream_t stream);
int clica_softclip_bwd(const float* X, int64_t ldx, const float* bound, const float* dY, int64_t lddy,
                       float* dX, int64_t lddx, float* dbound_partial /*[ceil(M/256), n] or NULL*/,
                       int64_t M, int32_t n, clica_stream_t stream);

/* Stand-alone LeakyReLU between a backbone's output and the encoder head's Linear
 * (/root/reference/main_3dident.py:365-370: Sequential(backbone, nn.LeakyReLU(), nn.Linear(10 n_lat, n_lat), rescaling)).
 * The backward takes the saved OUTPUT Yact (slope >= 0; slope = 0 is ReLU, whose output is positive exactly where its derivative is 1). */
int clica_leaky_relu_fwd(const float* X, int64_t ldx, float* Y, int64_t ldy, int64_t M, int32_t n, float slope,
                         clica_stream_t stream);
int clica_leaky_relu_bwd(const float* Yact, int64_t ldy, const float* dY, int64_t lddy, float* dX, int64_t lddx,
                         int64_t M, int32_t n, float slope, clica_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Convolution stack of the KITTI-masks encoder  --  BetaVAE_H, /root/reference/kitti_masks/model.py:41-56:
 * Conv2d(k = 4, stride 2, pad 1) + ReLU stages as implicit GEMMs (fp32 MFMA, csrc/linear.hip, conv section), channels-last.
 * (The reference runs them through nn.Conv2d = MIOpen; these entry points replace the forward, the data gradient and the
 * weight / bias gradients of kitti_masks/model.py:42-49; the k = 4 stage on the 4 x 4 map (:50) is clica_linear_* over the flattened map.)
 *
 *   S    padded space-to-depth input of a stage: [images][hs][ws][4 C], hs = H/2 + 1, ws = W/2 + 1,
 *        S[i][sy][sx][(py * 2 + px) * C + c] = in[i][2 sy + py - 1][2 sx + px - 1][c], zero where that pixel is outside the
 *        H x W map; the buffer must be readable (zeros) for (ws + 2) * 4 C floats behind its end.
 *   Wg   GEMM weights [Cout][16 C]: Wg[co][((dy * 2 + dx) * 4 + py * 2 + px) * C + c] = weight[co][c][2 dy + py][2 dx + px]
 *   rows GEMM rows cover the whole hs x ws grid, r = (i * hs + y) * ws + x; outputs are the rows with y < hs - 1, x < ws - 1.
 *
 *   fwd:   scatter = 1: out = the NEXT stage's S tensor ([images][(hs-1)/2 + 1][(ws-1)/2 + 1][4 Cout], border pre-zeroed by
 *          the caller and never written);  scatter = 0: out[r][Cout] for every row r (non-output rows hold finite garbage).
 *          scatter = 2: out[r][Cout] on the row grid like scatter = 0, but only the OUTPUT rows are computed and written (the caller
 *          zeroes the buffer once; the GEMM then runs over images * (hs-1) * (ws-1) rows -- scatter = 1 does the same).
 *          relu != 0 applies ReLU after the bias.
 *   fwd_patches: the same from an explicit patch matrix [images * ho * wo][K] (first stage; clica_conv_im2col_k4s2 builds it
 *          from the NCHW input with K = 16 C, column (ky * 4 + kx) * C + c), rows = output pixels of an ho x wo grid.
 *   dgrad: dO [rows][Cout] = gradient of the stage's pre-activation output on the hs x ws row grid, ZERO on non-output rows, with
 *          (ws + 1) * Cout readable zeros IN FRONT of it;  Wd [4 Cout][4 C]: Wd[((1-dy) * 2 + (1-dx)) * Cout + co][j] = Wg[co][(dy * 2 + dx) * 4 C + j];
 *          the result, gated by ReLU' of the previous stage (S > 0), is scattered to dPrev[images][dhs][dws][C] = the previous
 *          stage's dO grid (its valid 2(hs-1) x 2(ws-1) pixels are all written, the others never).
 *   gate_bits (optional, Cout resp. C a multiple of 32): the forward also writes one bit per output element, (out > 0), as words
 *          [row][Cout / 32] on ITS row grid (scatter = 1 only); the NEXT stage's dgrad, whose destination grid that is, reads the bit
 *          instead of the 4-byte element of S (the gate then costs 1/32 of the traffic: 485 -> 320 us on the widest stage).  NULL:
 *          no bits are written / the gate is read from S.
 *   wgrad: dWg[Cout][16 C] (+)= sum_r dO[r]^T A[r],  db[Cout] (+)= sum_r dO[r]   (A = the stage's implicit patch rows of S)
 * ---------------------------------------------------------------------------------- */
int clica_conv_im2col_k4s2(const float* x /*[images][C][H][W]*/, int64_t images, int32_t C, int32_t H, int32_t W,
                           float* patches /*[images * H/2 * W/2][16 C]*/, clica_stream_t stream);
int clica_conv_k4s2_fwd_patches(const float* patches, const float* Wg /*[Cout][K]*/, const float* bias, int64_t images, int32_t K,
                                int32_t Cout, int32_t ho, int32_t wo, int32_t relu, int32_t scatter, float* out,
                                uint32_t* gate_bits, clica_stream_t stream);
int clica_conv_k4s2_fwd(const float* S, const float* Wg, const float* bias, int64_t images, int32_t C, int32_t Cout,
                        int32_t hs, int32_t ws, int32_t relu, int32_t scatter, float* out, uint32_t* gate_bits, clica_stream_t stream);
int clica_conv_k4s2_dgrad(const float* dO, const float* Wd, const float* S, int64_t images, int32_t C, int32_t Cout,
                          int32_t hs, int32_t ws, float* dPrev, int32_t dhs, int32_t dws, const uint32_t* gate_bits,
                          clica_stream_t stream);
int clica_conv_k4s2_wgrad_workspace_bytes(int64_t rows, int32_t Cout, int32_t K, size_t* bytes);
int clica_conv_k4s2_wgrad(const float* dO, const float* S, int64_t images, int32_t C, int32_t Cout, int32_t hs, int32_t ws,
                          float* dWg, float* db, int32_t accumulate, void* workspace, size_t workspace_bytes,
                          clica_stream_t stream);

/* First stage: dWg[Cout][K] (+)= dO^T patches, db (+)= column sums of dO, from the explicit patch matrix (HBM-bound VALU kernel for a
 * small Cout x K, e.g. 32 x 16; Cout, K multiples of 4 with (Cout/4)(K/4) dividing 256 -- otherwise CLICA_E_INVALID: use clica_mlp_wgrad). */
int clica_conv_k4s2_wgrad_patches_workspace_bytes(int64_t rows, int32_t Cout, int32_t K, size_t* bytes);
int clica_conv_k4s2_wgrad_patches(const float* dO, const float* patches, int64_t rows, int32_t Cout, int32_t K, float* dWg, float* db,
                                  int32_t accumulate, void* workspace, size_t workspace_bytes, clica_stream_t stream);

/* Weight re-ordering for the conv stack: n <= 16 index-mapped copies in one launch, dst[i][e] (+)= map[i][e] >= 0 ? src[i][map[i][e]] : 0 for
 * e < count[i] (nn.Conv2d.weight [co][c][ky][kx] -> Wg / Wd / the zero-padded rows of the 4 x 4 stage before a step, GEMM-layout gradients
 * -> Conv2d.weight layout after it; the maps are permutations built once per shape by the caller; map[i] = NULL is the identity, e.g. a
 * bias gradient).  accumulate != 0 adds into dst: the step's gradients go straight into the optimizer's .grad views. */
/* First stage for ONE input channel straight from the images x [images][1][H][W] (no patch matrix): forward = clica_conv_k4s2_fwd_patches_amax
 * with K = 16, Cout = 32, scatter = 1 (out = the next stage's S, gate bits / maximum slots optional); weight gradient =
 * clica_conv_k4s2_wgrad_patches (same workspace size: clica_conv_k4s2_wgrad_patches_workspace_bytes(rows, Cout, 16)).  Wg / dWg in the
 * patch-matrix order [Cout][ky * 4 + kx].  H, W multiples of 4. */
int clica_conv_k4s2_fwd_image(const float* x, const float* Wg, const float* bias, int64_t images, int32_t H, int32_t W, int32_t Cout,
                              int32_t relu, float* out, uint32_t* gate_bits, uint32_t* amax_slots, clica_stream_t stream);
int clica_conv_k4s2_wgrad_image(const float* dO, const float* x, int64_t images, int32_t H, int32_t W, int32_t Cout, float* dWg, float* db,
                                int32_t accumulate, void* workspace, size_t workspace_bytes, clica_stream_t stream);
/* d loss / d image of the first stage (the reference's nn.Conv2d is differentiable w.r.t. its input, kitti_masks/model.py:41-56):
 * dO = gradient at the first stage's pre-activation on its (H/2) x (W/2) output grid [images][ho][wo][Cout], W = Conv2d.weight
 * [Cout][C][4][4] as the module stores it, dX = [images][C][H][W].  C <= 4, Cout * C <= 128. */
int clica_conv_k4s2_dgrad_input(const float* dO, const float* W, int64_t images, int32_t C, int32_t Cout, int32_t H, int32_t Wd,
                                float* dX, clica_stream_t stream);
int clica_conv_gather(int32_t n, const float* const* src, const int32_t* const* map, float* const* dst, const int32_t* count,
                      int32_t accumulate, clica_stream_t stream);

/* ------------------------------------------------------------------------------------
 * The same convolution stages in the f16x2 split arithmetic (csrc/conv16.hip)  --  kitti_masks/model.py:42-49.
 * Every product runs as three fp16 matrix instructions on two-piece operands (hi = RN_f16(v s), lo = RN_f16(v s - hi), s a power
 * of two per tensor; fp32 accumulation; the encoder's arithmetic, clica_mlp_*_split16).  Tensors stay fp32 in HBM with the layouts
 * documented above and are split by the consuming kernel; the scale of a tensor comes from its MAXIMUM SLOTS, 256 uint32 (float
 * bits of max |value|) that the producing kernel fills in the same step (amax_out) and the consumer reduces (amax_in; NULL = scale 1).
 * Slots are zeroed by the caller before the producer runs (clica_conv16_zero_slots: `tensors` consecutive slot arrays).
 *   pack:   n <= 8 weight matrices, dst[i] = [hi plane][lo plane] of count[i] f16 each, element e = piece(src[i][map[i][e]] * scale_i)
 *           (map[i] = NULL: identity; a negative index gives 0), scale_i from the maximum over src[i][0 .. src_count[i]), written to scales[i].
 *   fwd:    Wg16 = packed [Cout][16 C] (clica_conv_k4s2_fwd's Wg); scatter 1 or 2 as there; only output rows are computed.
 *   dgrad:  WdT16 = packed TRANSPOSE of clica_conv_k4s2_dgrad's Wd, i.e. [4 C][4 Cout]; gate bits required.
 *   wgrad:  as clica_conv_k4s2_wgrad (C a multiple of 32, Cout 32 or 64).
 *   clica_conv_k4s2_fwd_patches_amax: the fp32 first-stage kernel, also recording its output's maximum (K = 16, Cout = 32 only).
 *   clica_conv16_amax: max |x| of a tensor some other kernel produced, into its slots.
 * ---------------------------------------------------------------------------------- */
int clica_conv16_pack(int32_t n, const float* const* src, const int32_t* src_count, const int32_t* const* map, uint16_t* const* dst,
                      const int32_t* count, float* scales, clica_stream_t stream);
int clica_conv16_amax(const float* x, int64_t n, uint32_t* slots, clica_stream_t stream);
int clica_conv16_zero_slots(uint32_t* slots, int32_t tensors, clica_stream_t stream);
int clica_conv_k4s2_fwd_patches_amax(const float* patches, const float* Wg, const float* bias, int64_t images, int32_t K,
                                     int32_t Cout, int32_t ho, int32_t wo, int32_t relu, int32_t scatter, float* out,
                                     uint32_t* gate_bits, uint32_t* amax_slots, clica_stream_t stream);
/* First stage for ONE input channel in the same arithmetic: x [images][1][H][W] -> the next stage's S (scatter = 1), W16 = packed
 * [Cout = 32][16] in the patch-matrix order (ky * 4 + kx), amax_in = maximum slots of x (clica_conv16_amax), amax_out = those of S. */
int clica_conv16_first_fwd(const float* x, const uint16_t* W16, const float* wscale, const float* bias, int64_t images, int32_t H, int32_t W,
                           int32_t Cout, int32_t relu, float* out, uint32_t* gate_bits, const uint32_t* amax_in, uint32_t* amax_out,
                           clica_stream_t stream);
int clica_conv16_k4s2_fwd(const float* S, const uint16_t* Wg16, const float* wscale, const float* bias, int64_t images, int32_t C,
                          int32_t Cout, int32_t hs, int32_t ws, int32_t relu, int32_t scatter, float* out, uint32_t* gate_bits,
                          const uint32_t* amax_in, uint32_t* amax_out, clica_stream_t stream);
int clica_conv16_k4s2_dgrad(const float* dO, const uint16_t* WdT16, const float* wscale, int64_t images, int32_t C, int32_t Cout,
                            int32_t hs, int32_t ws, float* dPrev, int32_t dhs, int32_t dws, const uint32_t* gate_bits,
                            const uint32_t* amax_in, uint32_t* amax_out, clica_stream_t stream);
int clica_conv16_k4s2_wgrad_workspace_bytes(int64_t rows, int32_t Cout, int32_t K, size_t* bytes);
int clica_conv16_k4s2_wgrad(const float* dO, const float* S, int64_t images, int32_t C, int32_t Cout, int32_t hs, int32_t ws,
                            float* dWg, float* db, int32_t accumulate, const uint32_t* amax_dO, const uint32_t* amax_S,
                            void* workspace, size_t workspace_bytes, clica_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Mixing network g  --  construct_invertible_mlp's nn.Sequential forward,
 * /root/reference/invertible_network_utils.py:87-115: bias-free n x n Linear layers with
 * LeakyReLU(slope) between them; frozen (no backward).
 *   W : n_layers contiguous [n,n] row-major matrices (nn.Linear.weight layout)
 * ---------------------------------------------------------------------------------- */
int clica_mixing_fwd(const float* Z, int64_t ldz, const float* W, int32_t n_layers, float slope,
                     float* X, int64_t ldx, int64_t M, int32_t n, clica_stream_t stream);
/* The other hidden activations construct_invertible_mlp offers (invertible_network_utils.py:44-66, --act-fct):
 *   act_kind 0: LeakyReLU(act_param) [0.2; act_param = 0 is ReLU]   1: ELU(alpha = act_param)
 *            2: SmoothLeakyReLU, a x + (1 - a) log(1 + e^x)         3: Softplus(beta = act_param, threshold 20) */
int clica_mixing_fwd_act(const float* Z, int64_t ldz, const float* W, int32_t n_layers, int32_t act_kind, float act_param,
                         float* X, int64_t ldx, int64_t M, int32_t n, clica_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Exact k-nearest-neighbour search in squared L2 over a latent table in HBM  --  the lookup
 * datasets/threedident_dataset.py:71, 83, 104-105 does on the host with faiss.IndexFlatL2 (`index.add(latents)`,
 * `index.search(z, 1)`, `index.search(z_tilde, 2)`): dist[i, c] = |query_i - table_idx[i, c]|^2 ascending in c, ties to the
 * lower row; idx is int64 like faiss' labels (-1 where the table has fewer than k rows); dist may be NULL.
 * 1 <= n <= 64, 1 <= k <= 4.
 * ---------------------------------------------------------------------------------- */
int clica_nn_search_workspace_bytes(int64_t n_query, int64_t n_table, int32_t n, int32_t k, size_t* bytes);
int clica_nn_search(const float* table, int64_t ldt, int64_t n_table, const float* query, int64_t ldq, int64_t n_query,
                    int32_t n, int32_t k, int64_t* idx, float* dist, void* workspace, size_t workspace_bytes,
                    clica_stream_t stream);

/* ------------------------------------------------------------------------------------
 * KITTI-masks temporal pairs  --  the batch assembly of kitti_masks/dataset.py:90-131 (__getitem__: two frames of one pedestrian
 * sequence, uint8 * 255 -> float32 / 255) and :134-142 (custom_collate: first / second interleaved) from a frame table in HBM:
 *   images[2 i]     = float(frames[first_frame[i]]),  images[2 i + 1] = float(frames[second_frame[i]])     ([2 n_pairs][frame_elems])
 *   labels[2 i + c] = latents[first / second frame]                                                          ([2 n_pairs][n_latents], optional)
 * frames: uint8 [n_frames][frame_elems] (bool masks as 0 / 1), latents: float32 [n_frames][n_latents].
 * ---------------------------------------------------------------------------------- */
int clica_kitti_gather_pairs(const uint8_t* frames, int64_t frame_elems, int64_t n_frames, const int64_t* first_frame,
                             const int64_t* second_frame, int64_t n_pairs, float* images, const float* latents, int32_t n_latents,
                             float* labels, clica_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Second moments for the disentanglement scores  --  replaces the host-side sklearn / numpy / scipy passes of
 * disentanglement_utils.py:23 (r2_score), :40 (np.corrcoef), :38 (spearmanr, on ranks), :97-100 (LinearRegression.fit/predict):
 *   out[d][d] (fp64, row-major) = [A | B | 1]^T [A | B | 1],  d = da + db + 1 <= 129,
 * A [M, da], B [M, db] fp32 row-major with leading dimensions lda / ldb (B may be NULL with db = 0).  Every score of that
 * file is a function of this matrix; only d x d doubles leave the device.  Deterministic (no atomics).
 * ---------------------------------------------------------------------------------- */
int clica_moments_workspace_bytes(int64_t M, int32_t d, size_t* bytes);
int clica_moments(const float* A, int64_t lda, int32_t da, const float* B, int64_t ldb, int32_t db, int64_t M, double* out,
                  void* workspace, size_t workspace_bytes, clica_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Adam  --  torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8) as used at main_mlp.py:312,
 * over one flat parameter arena.  `step_dev` is a device int32 holding the number of
 * updates already applied; the call uses t = *step_dev + 1 for the bias corrections and
 * leaves *step_dev unchanged (advance it with clica_tick so a graph replay stays valid).
 * ---------------------------------------------------------------------------------- */
int clica_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                    float lr, float beta1, float beta2, float eps, float grad_scale,
                    const int32_t* step_dev, clica_stream_t stream);
/* Same update with t = *step_dev + t_offset (t_offset = 0 when the counter was already advanced earlier in the step,
 * e.g. by clica_lp_loss_bwd_sym_train's tick_counter). */
int clica_adam_step_at(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                       float lr, float beta1, float beta2, float eps, float grad_scale,
                       const int32_t* step_dev, int32_t t_offset, clica_stream_t stream);
/* clica_adam_step_at with the f16x2 encoder arithmetic's scale update (clica_split16_update(state, n_layers)) riding in the same launch
 * (its first 27 workgroups, one tensor each): the training step's last launch then also prepares the next step's scales. */
int clica_adam_step_s16(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                        float lr, float beta1, float beta2, float eps, float grad_scale,
                        const int32_t* step_dev, int32_t t_offset, void* split16_state, int32_t n_layers, clica_stream_t stream);
/* Same update, and the LAST workgroup to finish advances *step_dev by one (replaces the separate clica_tick launch).
 * `ticket` is a device int32 owned by the caller, zero before the first call; the kernel leaves it at zero. */
int clica_adam_step_tick(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                         float lr, float beta1, float beta2, float eps, float grad_scale,
                         int32_t* step_dev, int32_t* ticket, clica_stream_t stream);
/* clica_adam_step_tick with the f16x2 scale update and guard of clica_adam_step_s16 in the same launch (update number *step_dev + 1; the
 * launch's last workgroup advances the counter, a step the guard withholds leaves it where it was). */
int clica_adam_step_s16_tick(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                             float lr, float beta1, float beta2, float eps, float grad_scale,
                             int32_t* step_dev, int32_t* ticket, void* split16_state, int32_t n_layers, clica_stream_t stream);
/* *counter += 1 (single-thread kernel; keeps step/RNG counters on device for graph replay) */
int clica_tick(int32_t* counter, clica_stream_t stream);
/* Host reads from the MIDDLE of a captured step (cl_ica_amd/graphed.py: the reference's train_step reads three loss scalars per step,
 * main_mlp.py:283-285, which become final right after the loss forward).  A one-wave kernel copies src[0..n) (device, n <= 64) to
 * host_dst (host memory the device can write: hipHostMalloc / a pinned torch tensor), then advances *seq_dev and stores the new value
 * to *host_seq with a system-scope release.  The host spins on *host_seq reaching the number of replays it has issued and reads the
 * values while the rest of the graph (backward, optimizer) is still running. */
int clica_publish_host(const float* src, int32_t n, float* host_dst, uint32_t* seq_dev, uint32_t* host_seq, clica_stream_t stream);
/* Measurement aid (bench.py): device-side interval stamps that work INSIDE a captured graph, where event records cannot be timed.
 * `slot` = 1 + 2 * capacity device uint64, zero-initialised: a one-thread kernel writes the 100 MHz wall clock (s_memrealtime) to
 * begin (which = 0) / end (which = 1) entry (count % capacity) and the begin stamp advances the count in slot[0].  Launched on the
 * stream right before / after the kernel of interest, the pair brackets it exactly as the in-order stream executes it. */
int clica_stamp(unsigned long long* slot, int32_t which, int32_t capacity, clica_stream_t stream);
/* Shader-clock probe (bench.py): a one-wave kernel that writes n samples (wall clock in 100 MHz ticks, core-clock counter) `period_ticks`
 * apart into samples[2 n] and sleeps in between.  Launched on a SIDE stream it shares a CU with the kernels of the main stream; cycles /
 * wall time between two samples = the clock that XCD held over that stretch, which bench.py reports next to every in-step kernel time
 * (the chip throttles under bf16 MFMA load: every roofline fraction against the nominal peak contains it). */
int clica_clock_probe(unsigned long long* samples, int32_t n, int32_t period_ticks, clica_stream_t stream);

/* ------------------------------------------------------------------------------------
 * On-device latent samplers (Philox4x32-10, counter = (element, draw, *step_dev, stream_id))
 * replacing torch/NumPy host RNG + rejection loops with host syncs:
 *   spaces.py:273-302 (box uniform / truncated normal via spaces_utils.py:106-142),
 *   spaces.py:134-170 (sphere uniform / projected normal), spaces.py:44-119 (R^n),
 *   laplace spaces.py:74-96, generalized normal spaces_utils.py:82-103,
 *   von Mises-Fisher vmf.py:48-134 (Wood's rejection sampler).
 * `mean` may be NULL for the marginal kinds; for conditional kinds it is [M,n] (ldm) or, with
 * ldm == 0, a single row broadcast to all M samples.
 * ---------------------------------------------------------------------------------- */
enum clica_space { CLICA_SPACE_REAL = 0, CLICA_SPACE_BOX = 1, CLICA_SPACE_SPHERE = 2 };
enum clica_dist {
  CLICA_DIST_UNIFORM = 0,   /* box: U[min,max]^n ; sphere: uniform on S^{n-1}           */
  CLICA_DIST_NORMAL = 1,    /* N(mean, scale^2); box: per-element truncation; sphere: projected */
  CLICA_DIST_LAPLACE = 2,   /* Laplace(mean, scale)                                        */
  CLICA_DIST_GENNORM = 3,   /* generalized normal, exponent shape_p                        */
  CLICA_DIST_VMF = 4        /* von Mises-Fisher(mean, kappa = scale), sphere only          */
};
typedef struct clica_sampler_desc {
  int32_t space;      /* enum clica_space */
  int32_t dist;       /* enum clica_dist  */
  int32_t n;
  float box_min, box_max;
  float scale;        /* std / lambda / kappa */
  float shape_p;      /* exponent for GENNORM */
  uint64_t seed;
  uint32_t stream_id; /* distinguishes independent draws inside one step (z, z~, rank...) */
} clica_sampler_desc;
int clica_sample(const clica_sampler_desc* d, const float* mean, int64_t ldm,
                 float* out, int64_t ldo, int64_t M, const int32_t* step_dev,
                 clica_stream_t stream);
/* Same draw with a per-coordinate scale tensor: coordinate k of row i uses d->scale * scale_vec[i * lds + k] (lds = 0: one
 * row for every sample).  spaces.py:60-72, 157-166, 297: `std` of normal() may be a tensor of shape (n,), (1, n) or (size, n).
 * scale_vec = NULL is clica_sample. */
int clica_sample_scaled(const clica_sampler_desc* d, const float* mean, int64_t ldm,
                        const float* scale_vec, int64_t lds,
                        float* out, int64_t ldo, int64_t M, const int32_t* step_dev,
                        clica_stream_t stream);
/* z ~ marginal and z~ ~ conditional(. | z) (main_mlp.py:196-200) in ONE launch when both kinds are coordinate-wise
 * (box / R^n); the draws are bit-identical to clica_sample(marginal) followed by clica_sample(conditional, mean = z).
 * Row-wise kinds (sphere, vMF) fall through to those two launches. */
int clica_sample_pair(const clica_sampler_desc* marginal, const clica_sampler_desc* conditional,
                      const float* marginal_mean, int64_t ldmm, float* z, int64_t ldz, float* zt, int64_t ldzt,
                      int64_t M, const int32_t* step_dev, clica_stream_t stream);
/* The two independent front launches of a training step in ONE: clica_mlp_pack_split16_both (the f16x2 weight pieces of the
 * parameters Adam has just written; reference: the nn.Linear weights of encoders.py:36-48 as the kernels' operand) and clica_sample_pair
 * (z, z~ of main_mlp.py:196-200) -- same arguments, same results bit for bit (the pair draw keeps its Philox counters).  As two launches
 * in line they cost 12 + 9 us in front of the encoder; as two branches of a HIP graph the fork / join cost more than it hid.  Row-wise
 * sampler kinds (sphere, vMF) run the two calls one after the other. */
int clica_mlp_pack_split16_both_sample(int32_t n_layers, const float* const* W, const int64_t* ldw, const int32_t* N, const int32_t* K,
                                       void* packed_fwd, void* packed_bwd, void* state,
                                       const clica_sampler_desc* marginal, const clica_sampler_desc* conditional,
                                       const float* marginal_mean, int64_t ldmm, float* z, int64_t ldz, float* zt, int64_t ldzt,
                                       int64_t M, const int32_t* step_dev, clica_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CLICA_H */
