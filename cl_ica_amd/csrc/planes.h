// Operand format of the split-bf16 weight-gradient GEMM (wgrad_split.hip), written by the split-bf16 whole-encoder
// kernels (fused_mlp.hip: mlp_split_k) instead of -- or next to -- the fp32 copy of a layer output.
//
// A tensor T[M rows][F features] is stored as three bf16 PLANES (hi / mid / lo: T = hi + mid + lo exactly, 8 + 8 + 8
// mantissa bits by truncation) in units of 16 batch rows x 32 features:
//     buffer[row group g = row / 16][unit u = feature / 32][plane p][512 bf16]            (1 KB per piece, 3 KB per unit)
// and inside a piece, for key k = row % 16 and feature f = feature % 32:
//     index = (k / 4) * 128 + (f / 16) * 64 + (k % 4) * 16 + (f % 16)
// i.e. eight [4 keys][16 features] blocks of 128 bytes ordered (k / 4, f / 16).  Why this order: the consumer moves whole
// pieces HBM/L2 -> LDS with one global_load_lds_dwordx4 per wave (lane-linear image, no VGPRs) and reads an MFMA operand of
// v_mfma_f32_32x32x16_bf16 -- lane = (feature l & 31, key half l >> 5), eight consecutive keys per lane -- with two
// ds_read_b64_tr_b16: in each, lanes 0..31 cover 256 contiguous bytes (the blocks (2h, 0), (2h, 1)) and lanes 32..63 the
// next-but-one 256 (h = l >> 5), which is the conflict-free pattern of the transposing read.
// Feature F (the first padding column) may hold the constant 1 in the hi plane (`ones`): the product with it is
// db = column sums of the other operand.  Rows are padded to whole producer workgroups (48 rows = 3 groups).
#pragma once
#include <stdint.h>

namespace clica {
namespace planes {
constexpr int kGroupRows = 16, kUnitFeat = 32, kPieceBytes = 1024, kUnitBytes = 3 * kPieceBytes, kProducerRows = 48;
static inline int units(int width, int ones) { return (width + (ones ? 1 : 0) + kUnitFeat - 1) / kUnitFeat; }
static inline int64_t groups_alloc(int64_t M) { return (M + kProducerRows - 1) / kProducerRows * (kProducerRows / kGroupRows); }
static inline int64_t groups_used(int64_t M) { return (M + kGroupRows - 1) / kGroupRows; }
static inline size_t bytes(int64_t M, int width, int ones) { return (size_t)groups_alloc(M) * units(width, ones) * kUnitBytes; }
}  // namespace planes
}  // namespace clica
