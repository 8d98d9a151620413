// Pieces of the grouped weight-gradient path shared by linear.hip (fp32 MFMA bodies, tiny-dimension VALU kernel, deterministic
// slab reduction) and wgrad_split.hip (the split-bf16 MFMA body): argument blocks and the host-side launchers of the two
// helper kernels, which both arithmetic modes use unchanged.
#pragma once
#include "common.h"

namespace clica {
namespace gemm {

// One GEMM problem C[M,N] = A_op[M,Kc] * B_op[Kc,N] (see linear.hip for the three operand layouts).
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

constexpr int MAXG = 8;             // problems (layers) per grouped launch
constexpr int TINY_MAX_S = 16;
constexpr int TINY_ROWS = 64;       // batch rows per LDS tile
constexpr int TINY_MAX_LG = 128;    // widest large dimension: LDS tile [TINY_ROWS][Lg] = 32 KB
constexpr int TINY_THREADS = 256;
struct TinyArgs { int n; Args p[MAXG]; };

// Optional epilogue of the grouped slab reduction: the Adam update of the elements it has just reduced (adam_math.h), so that the
// training step needs no optimizer launch of its own.  dW / db of every problem must then be views of ONE gradient arena starting at
// `gbase`; p / m / v are the parameter / exp_avg / exp_avg_sq arenas of the same layout (element offset = address - gbase).
// s16_state != nullptr: the f16x2 arithmetic's scale update (split16.h) rides in front of the launch, as it does in adam_k.
struct ReduceAdam {
  float* p; float* m; float* v; const float* gbase;      // p == nullptr: plain reduction
  float lr, b1, b2, eps, gscale;
  const int* step_dev; int t_offset;
  void* s16_state; int s16_layers;
};
// grouped slab reduction: dW[i][j] (+)= sum_s slab[s][i][j];  db[i] (+)= sum_s dbslab[s][i]  for up to MAXG problems
struct ReduceGroupArgs {
  ReduceAdam adam;
  int n, accumulate;
  int splits[MAXG];
  int first[MAXG + 1];      // first block of each problem
  int dw_blocks[MAXG], vec4[MAXG];
  const float* slab[MAXG]; const float* dbslab[MAXG];
  float* dW[MAXG]; float* db[MAXG];
  int64_t M[MAXG], N[MAXG], lddw[MAXG];
};

// host side (defined in linear.hip)
bool wgrad_tiny_shape(int32_t N, int32_t K);                                    // the layer takes the tiny-dimension VALU kernel
void wgrad_tiny_plan(int64_t Mrows, int n_tiny, int* splits, int64_t* k_per_split);
int launch_wgrad_tiny(const TinyArgs& T, int splits, hipStream_t st);
int launch_slab_reduce_group(const ReduceGroupArgs& R, int blocks, hipStream_t st);
// fills the reduction entry of problem l (dW [N][K], `sp` slabs) and returns the number of blocks it adds
int slab_reduce_entry(ReduceGroupArgs& R, int l, int first_block, int sp, const float* slab, const float* dbslab,
                      float* dW, int64_t lddw, float* db, int32_t N, int32_t K);

}  // namespace gemm

namespace wsplit {
// The backward chain of a training step can leave the weight-gradient slabs of the encoder's n-wide first and last layer itself
// (fused_mlp.hip: SplitArgs::Tail; one slab per chain workgroup of kChainRows rows).  Defined in wgrad_split.hip, which owns the
// workspace layout: the encoder's shapes in forward order, as clica_mlp_wgrad_split* gets them.
constexpr int kChainRows = 48;
bool chain_tail_supported(int n_layers, const int32_t* N, const int32_t* K);
int chain_tail_slabs(int64_t M, int n_layers, const int32_t* N, const int32_t* K, void* workspace, size_t workspace_bytes,
                     float** slab_first, float** db_first, float** slab_last, float** db_last);
}  // namespace wsplit
}  // namespace clica
