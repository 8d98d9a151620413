// Weight gradients of the MLP encoder (get_mlp, /root/reference/encoders.py:36-48: the dW / db that autograd's
// nn.Linear backward produces) on the bf16 matrix cores with EXACT 3-way operand splits -- the weight-gradient half of the
// split-bf16 mode (fused_mlp.hip: mlp_split_k is the forward / backward-chain half).
//
//   dW_l[N, K] = dZ_l^T X_l,   db_l = dZ_l^T 1          (contraction over the 2B batch rows)
//
// Arithmetic: every fp32 operand value v is hi + mid + lo (three bf16 pieces, 8 + 8 + 8 mantissa bits by truncation, exact),
// and a product a.b is taken as the six piece products of order <= 2 (lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi), each
// exact in fp32, accumulated in fp32 by v_mfma_f32_32x32x16_bf16; the dropped products are below 2^-24 |a||b|.  fp32
// emulation, not reduced precision (same scheme and error as mlp_split_k: 8.6e-7 of max|y| against fp64, the fp32-MFMA
// kernels 1.0e-6).  Six 8-pass bf16 instructions cover K = 16 where fp32 MFMA needs eight 16-pass ones: 0.375 of the matrix time.
//
// Operands: both have the contraction along the batch rows, i.e. a bf16 MFMA operand (eight consecutive k per lane) is a
// TRANSPOSED read of the row-major tensors.  The producers (mlp_split_k epilogues, which hold the three pieces of every
// output value in registers anyway) therefore write the operands as bf16 planes in 16-row x 32-feature units (planes.h);
// a 1 KB piece goes HBM/L2 -> LDS with one global_load_lds_dwordx4 per wave (no VGPRs, no ds_write) and an operand
// fragment is two ds_read_b64_tr_b16 (the LDS transposing read of gfx950).  The fp32 copies of the hidden activations /
// gradients are not written at all in this mode (6 B instead of 4 B per element, but read by nobody else).
//
// Work decomposition = the fp32 grouped kernel's (linear.hip: wgrad_group_k): items (layer, contraction split, 256 x 128 or
// 128 x 256 output tile) of equal length, one 8-wave workgroup per CU, fp32 slabs + the shared deterministic slab reduction;
// db comes out of the matrix cores through the constant-1 feature the producer appends to X (planes.h).  Layers with a
// tiny dimension (the n-wide first / last layer) keep the fp32 VALU kernel (wgrad_tiny_k) on fp32 operands.
#include "common.h"
#include "planes.h"
#include "split16.h"
#include "wgrad_shared.h"
#include <algorithm>
#include <stdlib.h>

namespace clica {
namespace wsplit {

using gemm::MAXG;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef CLICA_WSPLIT_ABLATE      // timing ablations (WRONG results): 1 no DMA in the loop, 2 every step re-reads tile 0 (L2-hot), 4 no fragment reads (small body), 8 no per-step barrier (256 x 256 body)
#define CLICA_WSPLIT_ABLATE 0
#endif
constexpr int STG = 4;                      // LDS stages of one 16-row step each
constexpr int UW = 8, UN = 4;               // 32-feature units of the wide / narrow side of a tile (256 / 128 features)
constexpr int THREADS = 512;
// The two split arithmetics (fused_mlp.hip: Arith): AR = 0 bf16x3 -- three pieces per unit, six piece products; AR = 1 f16x2 -- two
// pieces per unit (planes.h with 2 KB units), three products (hi.hi, hi.lo, lo.hi), operands scaled per tensor (the scales in force
// arrive as device pointers in Prob and are divided out of the slab).
template <int AR> struct WArith;
template <> struct WArith<0> { static constexpr int NP = 3, NPROD = 6; static constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0}; };
template <> struct WArith<1> { static constexpr int NP = 2, NPROD = 3; static constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0}; };
template <int AR> constexpr int pieces_of() { return WArith<AR>::NP * (UW + UN); }        // one-KB pieces per step (36 / 24 KB)
template <int AR> constexpr int stage_bytes_of() { return pieces_of<AR>() * 1024; }
constexpr size_t kLdsBytes = (size_t)STG * stage_bytes_of<0>();      // 144 KB (the f16x2 launches use 96 KB)
template <int AR> constexpr size_t lds_bytes_of() { return (size_t)STG * stage_bytes_of<AR>(); }
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct Prob {
  const char* A; const char* B;   // plane buffers of dZ_l (rows of dW) and X_l (columns of dW)
  int fuA, fuB;                   // 32-feature units per 16-row group
  float* C; int64_t ldc;          // slabs [splits][M][ldc]
  float* dbslab;                  // [splits][M] or nullptr
  int M, N;                       // dW rows (N_l) and columns (K_l)
  int groups, gps;                // 16-row groups of the batch; groups per contraction split
  int gx, gy, a_wide;             // tiles along the columns / rows; tile shape 1: 256 x 128, 0: 128 x 256, 2: 256 x 256 (body_big)
  int n_big, rows_big;            // gemm_split_k, a_wide == 3 (mixed): the first n_big items are 256 x 256 tiles over rows [0, rows_big), the rest 128 x 256
  // fused epilogues of gemm_split_k (forward / data gradient of a wide layer, see below); unused (zero) in the weight-gradient launch
  int epi;                        // 1: + bias, LeakyReLU   2: x LeakyReLU'(sign of the layer's input activation)
  int leaky; float slope;
  const float* bias;              // [N] or nullptr
  const char* signT; int fuS;     // epi 2: T-planes of the activation whose sign gates the gradient (rows = C column, features = C row)
  char* outT; int fuT;            // output as planes of C^T (rows = C column, features = C row) or nullptr
  char* outN; int fuN;            // output as planes of C   (rows = C row, features = C column) or nullptr
  float* outF; int64_t ldf;       // output as fp32 C[row][column] or nullptr
  const float* scaleA; const float* scaleB;   // f16x2 only: device pointers to the scales of the A and B tensors (the slab gets acc / (sA sB))
  s16::S16Tensor outS;                        // f16x2 fused epilogues: scale in force for the OUTPUT tensor's planes, and where to record its maximum
};
struct GroupArgs { int n, total; int first[MAXG + 1]; Prob p[MAXG]; };

__device__ __attribute__((aligned(16))) unsigned g_zero16[4] = {0u, 0u, 0u, 0u};

typedef __attribute__((address_space(3))) char* lds_ptr;
// one wave instruction: 64 lanes x 16 B from per-lane global addresses to the lane-linear 1 KB at LDS offset `lds_off`.
// Issued from inline asm: the compiler does not see an LDS write through VMEM (it would serialise every later ds_read
// behind vmcnt(0)); the waits in the loop below are the exact ones.
__device__ __forceinline__ void dma_1k(const char* lane_src, unsigned lds_off) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(lane_src), "s"(lds_off) : "memory", "m0");
#pragma clang diagnostic pop
}
template <int N> __device__ __forceinline__ void wait_vm() {   // s_waitcnt vmcnt(N) only (gfx9 encoding)
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}
// A fragment is kept as four 32-bit registers (not as eight bf16 values: across the loop's back edge the optimiser would
// split a bf16 vector into 16-bit scalars and re-pack it with v_perm_b32 in front of every MFMA).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef u32x4 frag_t;
__device__ __forceinline__ frag_t read_frag(const char* p) {   // keys 8h .. 8h+3 and 8h+4 .. 8h+7 of this lane's feature (planes.h)
  const u32x2 a = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds_ptr)p));
  const u32x2 b = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds_ptr)(p + 256)));
  return (frag_t){a.x, a.y, b.x, b.y};
}

template <int AR>
__device__ __forceinline__ f32x16 mfma32(const frag_t a, const frag_t b, const f32x16 c) {
  if constexpr (AR == 0) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// Debug build only (-DCLICA_WSPLIT_TRACE, tools/wsplit_trace.py): s_memtime stamps per (workgroup, wave, phase), kept in
// scalar registers and written once at the end of the item (a store or a pointer load inside the loop would enter the
// wave's in-order vmcnt queue and change the very waits that are being measured)
#ifdef CLICA_WSPLIT_TRACE
__device__ unsigned long long* g_wstrace = nullptr;
#define WS_DECL unsigned long long ws_ts[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define WS_STAMP(ph) do { ws_ts[ph] = __builtin_readcyclecounter(); } while (0)
#define WS_STEP_STAMP(t, ph) do { if ((t) == 30) WS_STAMP(ph); } while (0)
#define WS_NOTE(q, v) do { ws_ts[q] = (unsigned long long)(v); } while (0)
#define WS_FLUSH do { if (g_wstrace && (threadIdx.x & 63) == 0) { for (int q_ = 0; q_ < 16; ++q_) g_wstrace[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + q_] = ws_ts[q_]; } } while (0)
#else
#define WS_DECL do { } while (0)
#define WS_STAMP(ph) do { } while (0)
#define WS_STEP_STAMP(t, ph) do { } while (0)
#define WS_NOTE(q, v) do { } while (0)
#define WS_FLUSH do { } while (0)
#endif

// ---- fused epilogue of the forward / data-gradient GEMMs of a wide layer (gemm_split_k) ------------------------------------------
// The accumulator block (i, j) of a wave holds C[row0 + 32 i + 8 q + 4 h + e][col0 + 32 j + (lane & 31)], q, e = 0..3: per lane FOUR
// CONSECUTIVE ROWS of one column.  With C = Y (rows = batch row m, columns = output feature n) that is
//   * 8 contiguous bytes per plane of the T-planes of Y (planes of Y^T: rows = n, features = m) -- the operand format of the NEXT
//     layer's forward and of this layer's sign gate in the backward chain;
//   * after a 4 x 4 transpose inside each lane quad (quad_transpose16: a lane then holds ONE row of the quad's four columns), 8
//     contiguous bytes per plane of the N-planes of Y (rows = m, features = n) -- the X operand of the next layer's weight gradient;
//   * four 4-byte stores of the fp32 copy (lanes = consecutive columns: 128 contiguous bytes), only where an fp32 consumer follows.
// A 2000-deep contraction amortises all of it: ~500 stores per wave behind ~6000 MFMAs.
__device__ __forceinline__ void split3_hi(float v, unsigned& hb, unsigned& mb, unsigned& lb);
// 4 x 4 transpose of 16-bit values inside a lane quad (lanes 4a .. 4a + 3): in, lane j holds rows 0..3 of its column as
// w0 = [row 0 | row 1 << 16], w1 = [row 2 | row 3 << 16]; out, lane j holds ROW j of the quad's four columns, two per word.
// Two exchange steps on the data-parallel-primitive path (no LDS): partner j ^ 1 with a byte permute, partner j ^ 2 with a select.
__device__ __forceinline__ u32x2 quad_transpose16(unsigned w0, unsigned w1, const unsigned sel, const bool upper) {
  const unsigned p0 = (unsigned)__builtin_amdgcn_mov_dpp((int)w0, 0xB1, 0xF, 0xF, true);      // quad_perm [1, 0, 3, 2]
  const unsigned p1 = (unsigned)__builtin_amdgcn_mov_dpp((int)w1, 0xB1, 0xF, 0xF, true);
  const unsigned a0 = __builtin_amdgcn_perm(p0, w0, sel);       // even lane: rows 0 of (j, j + 1); odd lane: rows 1 of (j - 1, j)
  const unsigned a1 = __builtin_amdgcn_perm(p1, w1, sel);       // rows 2 / rows 3
  const unsigned q = upper ? a0 : a1;
  const unsigned r = (unsigned)__builtin_amdgcn_mov_dpp((int)q, 0x4E, 0xF, 0xF, true);        // quad_perm [2, 3, 0, 1]
  return upper ? (u32x2){r, a1} : (u32x2){a0, r};
}
template <int NI>
__device__ __forceinline__ void fused_epilogue(const Prob& g, const f32x16 (&acc)[NI][2], const int row0, const int col0, int lane) {
  // the lane-derived addresses are recomputed from an opaque copy of the lane id: hoisted above the k-loop by the compiler they
  // would be live across it and spill (the 256 x 256 body has 14 registers to spare)
  asm volatile("" : "+v"(lane));
  const int h = lane >> 5, l31 = lane & 31;
  const bool fwd = g.epi == 1;
  const bool slope01 = g.slope > 0.f && g.slope < 1.f;
  const int j4 = lane & 3;                                     // position inside the lane quad = column inside a group of four
  const unsigned sel = (j4 & 1) ? 0x03020706u : 0x05040100u;
  const bool upper = (j4 & 2) != 0;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = col0 + j * 32 + l31;
    const bool col_ok = col < g.N;                             // (no early exit: every lane of a quad takes part in the transposes)
    const int colc = col_ok ? col : 0;
    const float b = (fwd && g.bias && col_ok) ? g.bias[colc] : 0.f;
    // byte offset of (row = col, feature 0) inside T-planes with fu units per group; of (row 0, feature = col) inside N-planes
    const size_t t_grp = (size_t)(colc >> 4) * 3072, t_in = (size_t)(((colc & 15) >> 2) * 256 + (colc & 3) * 32);
    const size_t n_col = (size_t)(colc >> 5) * 3072 + (size_t)(((colc >> 4) & 1) * 128 + (colc & 15) * 2);
    const int c0 = col & ~3;
    const bool quad_full = c0 + 3 < g.N;                       // the four columns of this quad are all real features
    const size_t n_c0 = (size_t)(c0 >> 5) * 3072 + (size_t)(((c0 >> 4) & 1) * 128 + (c0 & 15) * 2);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = row0 + i * 32 + 8 * q + 4 * h;          // rows m .. m + 3 (the same for the four lanes of a quad)
        const bool m_ok = m < g.M;
        const int mc = m_ok ? m : 0;
        const size_t t_feat = (size_t)(mc >> 5) * 3072 + (size_t)(((mc >> 4) & 1) * 128 + (mc & 15) * 2);
        unsigned sgn[2] = {0x3F803F80u, 0x3F803F80u};           // "positive" when there is no gate
        if (!fwd && g.signT && col_ok && m_ok) {
          const u32x2 sv = *reinterpret_cast<const u32x2*>(g.signT + t_grp * g.fuS + t_in + t_feat);
          sgn[0] = sv.x; sgn[1] = sv.y;
        }
        unsigned hb[4], mb[4], lb[4];
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = acc[i][j][4 * q + e];
          if (fwd) {
            t += b;
            if (g.leaky) t = slope01 ? fmaxf(t, t * g.slope) : (t > 0.f ? t : t * g.slope);
          } else {
            const unsigned hbits = (e & 1) ? (sgn[e >> 1] >> 16) : (sgn[e >> 1] & 0xFFFFu);      // hi piece of the activation (bf16)
            const bool pos = (hbits & 0x8000u) == 0u && (hbits & 0x7FFFu) != 0u;
            t = pos ? t : t * g.slope;
          }
          if (m + e >= g.M || !col_ok) t = 0.f;
          v[e] = t;
          split3_hi(t, hb[e], mb[e], lb[e]);
        }
        const u32x2 ph = (u32x2){(hb[0] >> 16) | hb[1], (hb[2] >> 16) | hb[3]};
        const u32x2 pm = (u32x2){(mb[0] >> 16) | mb[1], (mb[2] >> 16) | mb[3]};
        const u32x2 pl = (u32x2){(lb[0] >> 16) | lb[1], (lb[2] >> 16) | lb[3]};
        if (g.outT && col_ok && m_ok) {
          char* dst = g.outT + t_grp * g.fuT + t_in + t_feat;
          *reinterpret_cast<u32x2*>(dst) = ph;
          *reinterpret_cast<u32x2*>(dst + 1024) = pm;
          *reinterpret_cast<u32x2*>(dst + 2048) = pl;
        }
        if (g.outN) {
          // this lane's four rows of ONE column -> one row (m + j4) of the quad's FOUR columns: 8 contiguous bytes per plane
          const u32x2 th = quad_transpose16(ph.x, ph.y, sel, upper);
          const u32x2 tm = quad_transpose16(pm.x, pm.y, sel, upper);
          const u32x2 tl = quad_transpose16(pl.x, pl.y, sel, upper);
          const int row = m + j4;
          if (quad_full) {
            if (row < g.M) {
              char* dst = g.outN + (size_t)(row >> 4) * 3072 * g.fuN + (size_t)(((row & 15) >> 2) * 256 + (row & 3) * 32) + n_c0;
              *reinterpret_cast<u32x2*>(dst) = th;
              *reinterpret_cast<u32x2*>(dst + 1024) = tm;
              *reinterpret_cast<u32x2*>(dst + 2048) = tl;
            }
          } else if (col_ok && m_ok) {      // ragged last quad of the feature range: element stores (the ones column next to it stays)
            char* dst = g.outN + (size_t)(m >> 4) * 3072 * g.fuN + (size_t)(((m & 15) >> 2) * 256) + n_col;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (m + e >= g.M) continue;
              unsigned short* d2 = reinterpret_cast<unsigned short*>(dst + e * 32);
              d2[0] = (unsigned short)(hb[e] >> 16); d2[512] = (unsigned short)(mb[e] >> 16); d2[1024] = (unsigned short)(lb[e] >> 16);
            }
          }
        }
        if (g.outF && col_ok) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (m + e < g.M) g.outF[(int64_t)(m + e) * g.ldf + col] = v[e];
        }
      }
    }
  }
}

// The same epilogue in the f16x2 arithmetic: the accumulator holds (sA sB) x the true product; t = acc / (sA sB) (+ bias, LeakyReLU, or
// the sign gate -- read from the hi piece of the gate tensor's T-planes: an fp16's sign bit and "non-zero" test are the bf16 ones), the
// workgroup's max |t| goes to the output tensor's slot (split16.h), the planes get the two fp16 pieces of t x s_out (units of 2 KB: two
// pieces), the fp32 copy gets t.  The constant-1 column of an N-plane buffer is 1.0 whatever the scale (written once by the conversion
// kernel that prepared the buffer) and is left alone here as in the bf16x3 epilogue.
__device__ __forceinline__ void split16_hi_lo(float t, unsigned& hb, unsigned& lb) {     // piece bits in the LOW half
  const _Float16 h = (_Float16)t;
  hb = (unsigned)__builtin_bit_cast(unsigned short, h);
  lb = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)(t - (float)h));
}
template <int NI>
__device__ __forceinline__ void fused_epilogue16(const Prob& g, const f32x16 (&acc)[NI][2], const int row0, const int col0, int lane) {
  asm volatile("" : "+v"(lane));
  __shared__ float red[8];
  const int h = lane >> 5, l31 = lane & 31;
  const bool fwd = g.epi == 1;
  const bool slope01 = g.slope > 0.f && g.slope < 1.f;
  const int j4 = lane & 3;
  const unsigned sel = (j4 & 1) ? 0x03020706u : 0x05040100u;
  const bool upper = (j4 & 2) != 0;
  const float unscale = 1.f / (*g.scaleA * *g.scaleB);
  const float s_out = g.outS.scale ? *g.outS.scale : 1.f;
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = col0 + j * 32 + l31;
    const bool col_ok = col < g.N;
    const int colc = col_ok ? col : 0;
    const float b = (fwd && g.bias && col_ok) ? g.bias[colc] : 0.f;
    const size_t t_grp = (size_t)(colc >> 4) * 2048, t_in = (size_t)(((colc & 15) >> 2) * 256 + (colc & 3) * 32);
    const size_t n_col = (size_t)(colc >> 5) * 2048 + (size_t)(((colc >> 4) & 1) * 128 + (colc & 15) * 2);
    const int c0 = col & ~3;
    const bool quad_full = c0 + 3 < g.N;
    const size_t n_c0 = (size_t)(c0 >> 5) * 2048 + (size_t)(((c0 >> 4) & 1) * 128 + (c0 & 15) * 2);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = row0 + i * 32 + 8 * q + 4 * h;
        const bool m_ok = m < g.M;
        const int mc = m_ok ? m : 0;
        const size_t t_feat = (size_t)(mc >> 5) * 2048 + (size_t)(((mc >> 4) & 1) * 128 + (mc & 15) * 2);
        unsigned sgn[2] = {0x3C003C00u, 0x3C003C00u};           // "positive" when there is no gate
        if (!fwd && g.signT && col_ok && m_ok) {
          const u32x2 sv = *reinterpret_cast<const u32x2*>(g.signT + t_grp * g.fuS + t_in + t_feat);
          sgn[0] = sv.x; sgn[1] = sv.y;
        }
        unsigned hb[4], lb[4];
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = acc[i][j][4 * q + e] * unscale;
          if (fwd) {
            t += b;
            if (g.leaky) t = slope01 ? fmaxf(t, t * g.slope) : (t > 0.f ? t : t * g.slope);
          } else {
            const unsigned hbits = (e & 1) ? (sgn[e >> 1] >> 16) : (sgn[e >> 1] & 0xFFFFu);
            const bool pos = (hbits & 0x8000u) == 0u && (hbits & 0x7FFFu) != 0u;
            t = pos ? t : t * g.slope;
          }
          if (m + e >= g.M || !col_ok) t = 0.f;
          v[e] = t;
          amax = fmaxf(amax, fabsf(t));
          split16_hi_lo(t * s_out, hb[e], lb[e]);
        }
        const u32x2 ph = (u32x2){hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16)};
        const u32x2 pl = (u32x2){lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16)};
        if (g.outT && col_ok && m_ok) {
          char* dst = g.outT + t_grp * g.fuT + t_in + t_feat;
          *reinterpret_cast<u32x2*>(dst) = ph;
          *reinterpret_cast<u32x2*>(dst + 1024) = pl;
        }
        if (g.outN) {
          const u32x2 th = quad_transpose16(ph.x, ph.y, sel, upper);
          const u32x2 tl = quad_transpose16(pl.x, pl.y, sel, upper);
          const int row = m + j4;
          if (quad_full) {
            if (row < g.M) {
              char* dst = g.outN + (size_t)(row >> 4) * 2048 * g.fuN + (size_t)(((row & 15) >> 2) * 256 + (row & 3) * 32) + n_c0;
              *reinterpret_cast<u32x2*>(dst) = th;
              *reinterpret_cast<u32x2*>(dst + 1024) = tl;
            }
          } else if (col_ok && m_ok) {
            char* dst = g.outN + (size_t)(m >> 4) * 2048 * g.fuN + (size_t)(((m & 15) >> 2) * 256) + n_col;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (m + e >= g.M) continue;
              unsigned short* d2 = reinterpret_cast<unsigned short*>(dst + e * 32);
              d2[0] = (unsigned short)hb[e]; d2[512] = (unsigned short)lb[e];
            }
          }
        }
        if (g.outF && col_ok) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (m + e < g.M) g.outF[(int64_t)(m + e) * g.ldf + col] = v[e];
        }
      }
    }
  }
  if (g.outS.slots) s16::s16_commit_block_max(g.outS, amax, red, blockIdx.x, gridDim.x);
}


// The f16x2 epilogue SPECIALISED per output combination (round 6).  The generic version above decides direction and the
// three output formats at run time per accumulator block and clamps every address: ~12 000 instructions and 204 spilled registers behind
// the 256 x 256 body -- run by all eight waves of a CU at once, with nothing to hide behind.  Here direction and outputs are template
// parameters, the host guarantees M % 4 == 0 and N % 4 == 0 (row quads and lane quads are whole: one predicate per store, no element
// fallback), every address is ONE lane base per column block plus a compile-time offset per accumulator block (the plane layout is linear
// in (i, q) because a wave's row0 is a multiple of 32), and the gate words of a column block are loaded up front, ahead of the stores
// (the data-gradient epilogue used to wait for the previous block's stores in front of every gate word: vmcnt counts both).
constexpr int epi_code(bool bwd, bool T, bool N, bool F) { return 16 + (bwd ? 8 : 0) + (T ? 4 : 0) + (N ? 2 : 0) + (F ? 1 : 0); }
template <int NI, int CODE>
__device__ __forceinline__ void fused_epilogue16_fast(const Prob& g, const f32x16 (&acc)[NI][2], const int row0, const int col0, int lane) {
  constexpr bool BWD = (CODE & 8) != 0, HAS_T = (CODE & 4) != 0, HAS_N = (CODE & 2) != 0, HAS_F = (CODE & 1) != 0;
  asm volatile("" : "+v"(lane));
  __shared__ float red[8];
  const int h = lane >> 5, l31 = lane & 31, j4 = lane & 3;
  const unsigned sel = (j4 & 1) ? 0x03020706u : 0x05040100u;
  const bool upper = (j4 & 2) != 0;
  const float unscale = 1.f / (*g.scaleA * *g.scaleB);
  const float s_out = g.outS.scale ? *g.outS.scale : 1.f;
  const float slope = g.slope;
  const bool leaky = g.leaky != 0, slope01 = slope > 0.f && slope < 1.f;
  const int M = g.M;
  const int mrow = row0 + 4 * h;                              // first row of this lane's accumulator rows (block (0, 0))
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = col0 + j * 32 + l31;
    const bool col_ok = col < g.N;
    const float b = (!BWD && g.bias && col_ok) ? g.bias[col] : 0.f;
    // lane bases (bytes): T-planes: (row = col, feature = mrow); N-planes: (row = mrow + j4, feature = first column of the lane quad)
    const size_t t_lane = (size_t)(((col & 15) >> 2) * 256 + (col & 3) * 32) + (size_t)(row0 >> 5) * 2048 + (size_t)(8 * h);
    const int c0 = col & ~3;
    char* pT = nullptr; const char* pS = nullptr; char* pN = nullptr; float* pF = nullptr;
    if constexpr (HAS_T) pT = g.outT + (size_t)(col >> 4) * 2048 * g.fuT + t_lane;
    if constexpr (BWD) pS = g.signT ? g.signT + (size_t)(col >> 4) * 2048 * g.fuS + t_lane : nullptr;
    const size_t n_step = (size_t)2048 * g.fuN;              // one 16-row group of the N-planes
    if constexpr (HAS_N) pN = g.outN + (size_t)(row0 >> 4) * n_step + (size_t)(h * 256 + j4 * 32)
                              + (size_t)(c0 >> 5) * 2048 + (size_t)(((c0 >> 4) & 1) * 128 + (c0 & 15) * 2);
    if constexpr (HAS_F) pF = g.outF + (int64_t)mrow * g.ldf + col;
    u32x2 gate[NI][4];
    if constexpr (BWD) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          gate[i][q] = (u32x2){0x3C003C00u, 0x3C003C00u};        // "positive" when there is no gate
          if (pS && col_ok && mrow + i * 32 + 8 * q < M)
            gate[i][q] = *reinterpret_cast<const u32x2*>(pS + i * 2048 + (q >> 1) * 128 + (q & 1) * 16);
        }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool ok = col_ok && (mrow + i * 32 + 8 * q < M);    // rows m .. m + 3 exist together (M % 4 == 0)
        unsigned hb[4], lb[4];
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = acc[i][j][4 * q + e] * unscale;
          if constexpr (!BWD) {
            t += b;
            if (leaky) t = slope01 ? fmaxf(t, t * slope) : (t > 0.f ? t : t * slope);
          } else {
            const unsigned w = (e >> 1) ? gate[i][q].y : gate[i][q].x;
            const unsigned hbits = (e & 1) ? (w >> 16) : (w & 0xFFFFu);
            const bool pos = (hbits & 0x8000u) == 0u && (hbits & 0x7FFFu) != 0u;
            t = pos ? t : t * slope;
          }
          t = ok ? t : 0.f;
          v[e] = t;
          amax = fmaxf(amax, fabsf(t));
          if constexpr (HAS_T || HAS_N) split16_hi_lo(t * s_out, hb[e], lb[e]);
        }
        if constexpr (HAS_T || HAS_N) {
          const u32x2 ph = (u32x2){hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16)};
          const u32x2 pl = (u32x2){lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16)};
          if constexpr (HAS_T) {
            if (ok) {
              char* dst = pT + i * 2048 + (q >> 1) * 128 + (q & 1) * 16;
              *reinterpret_cast<u32x2*>(dst) = ph;
              *reinterpret_cast<u32x2*>(dst + 1024) = pl;
            }
          }
          if constexpr (HAS_N) {
            const u32x2 th = quad_transpose16(ph.x, ph.y, sel, upper);      // (all lanes take part)
            const u32x2 tl = quad_transpose16(pl.x, pl.y, sel, upper);
            if (ok) {                                                       // (N % 4 == 0: the quad's four columns exist together)
              char* dst = pN + (size_t)(2 * i + (q >> 1)) * n_step + (q & 1) * 512;
              *reinterpret_cast<u32x2*>(dst) = th;
              *reinterpret_cast<u32x2*>(dst + 1024) = tl;
            }
          }
        }
        if constexpr (HAS_F) {
          if (ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) pF[(int64_t)(i * 32 + 8 * q + e) * g.ldf] = v[e];
          }
        }
      }
    }
  }
  if (g.outS.slots) s16::s16_commit_block_max(g.outS, amax, red, blockIdx.x, gridDim.x);
}

// One work item: output tile (bx, by) of problem g over the row groups of contraction split bz.
// Eight waves, wave tile 64 x 64 = 2 x 2 accumulator blocks of 32 x 32 (64 registers); per 16-row step a wave reads
// 2 + 2 unit fragments x 3 planes (24 transposing reads of 512 B) and issues 24 MFMAs (768 matrix cycles).
// Pipeline: four 36 KB stages; the pieces of step t + 3 are requested during the first half of step t, the fragments of
// step t + 1 are read during its second half (two register sets, ping-pong), one barrier per step (in the middle).
template <bool A_WIDE, int FUSED, int AR>
__device__ __forceinline__ void body(const Prob& g, const int bx, const int by, const int bz) {
  constexpr int NP = WArith<AR>::NP, NPROD = WArith<AR>::NPROD, PIECES = pieces_of<AR>(), STAGE_BYTES = stage_bytes_of<AR>();
  constexpr int NJ = (PIECES + 7) / 8;                       // DMA pieces per wave and step (the last one only for the first PIECES - 8 (NJ - 1) waves)
  constexpr int NM = NPROD * 4;                              // MFMAs per wave and step

  constexpr int UA = A_WIDE ? UW : UN, UB = A_WIDE ? UN : UW;
  constexpr int BM = UA * 32, BN = UB * 32;
  constexpr int WN = BN / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave % WN, h = lane >> 5, l31 = lane & 31;
  const int g0 = bz * g.gps;
  const int nt = min(g.groups, g0 + g.gps) - g0;
  WS_DECL;
  WS_STAMP(0);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // this wave's DMA pieces: piece pi = wave + 8 j of the stage image [A: UA units x NP planes][B: UB units x NP planes];
  // units beyond the tensor (a tile that sticks out) are read from a 16-byte page of zeros with stride 0
  const bool five = wave < PIECES - 8 * (NJ - 1);            // bf16x3: waves 0..3 move five pieces per step, waves 4..7 four; f16x2: three each
  const char* src[NJ]; int stride[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int pi = wave + 8 * j;
    const bool is_a = pi < NP * UA;
    const int q = is_a ? pi : pi - NP * UA;
    const int unit = q / NP, plane = q - NP * unit;
    const int u = (is_a ? by * UA : bx * UB) + unit, fu = is_a ? g.fuA : g.fuB;
    const bool ok = pi < PIECES && u < fu;
    const char* base = is_a ? g.A : g.B;
    src[j] = ok ? base + (((int64_t)g0 * fu + u) * NP + plane) * 1024 + lane * 16 : reinterpret_cast<const char*>(g_zero16);
    stride[j] = (ok && !(CLICA_WSPLIT_ABLATE & 2)) ? fu * NP * planes::kPieceBytes : 0;
  }
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
  auto issue = [&](int t) {
    const unsigned st = lds0 + (unsigned)((t % STG) * STAGE_BYTES + wave * 1024);
#pragma unroll
    for (int j = 0; j < NJ - 1; ++j) { dma_1k(src[j], st + 8192u * j); src[j] += stride[j]; }
    if (five) { dma_1k(src[NJ - 1], st + 8192u * (NJ - 1)); src[NJ - 1] += stride[NJ - 1]; }
  };
  // at most `k` of the most recently requested steps may still be in flight
  auto wait_steps = [&](int k) {
    if (k <= 0) wait_vm<0>();
    else if (five) { if (k == 1) wait_vm<NJ>(); else wait_vm<2 * NJ>(); }
    else { if (k == 1) wait_vm<NJ - 1>(); else wait_vm<2 * (NJ - 1)>(); }
  };

  const int lane_off = h * 512 + ((lane >> 4) & 1) * 128 + (lane & 15) * 8;
  const char* fa_base = smem + lane_off + (wm * 2) * NP * 1024;
  const char* fb_base = smem + lane_off + (UA + wn * 2) * NP * 1024;
  frag_t fa0[NP][2], fb0[NP][2], fa1[NP][2], fb1[NP][2];
  auto load_frags_of = [&](frag_t (&f)[NP][2], const char* base, int t) {
    const int so = (t % STG) * STAGE_BYTES;
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) f[p][i] = read_frag(base + so + (NP * i + p) * 1024);
  };
  auto load_frags = [&](frag_t (&fa)[NP][2], frag_t (&fb)[NP][2], int t) { load_frags_of(fa, fa_base, t); load_frags_of(fb, fb_base, t); };
  // (dZ piece, X piece) of the products, small terms first: WArith<AR>::PA / PB
  auto mfma1 = [&](const frag_t (&fa)[NP][2], const frag_t (&fb)[NP][2], int m) {      // MFMA m = 0..NM-1 of a step: product m / 4, block ((m % 4) / 2, m % 2)
    const int t = m >> 2, i = (m >> 1) & 1, j = m & 1;
    acc[i][j] = mfma32<AR>(fa[WArith<AR>::PA[t]][i], fb[WArith<AR>::PB[t]][j], acc[i][j]);
  };
  auto issue_one = [&](int t, int j) { dma_1k(src[j], lds0 + (unsigned)((t % STG) * STAGE_BYTES + (wave + 8 * j) * 1024)); src[j] += stride[j]; };
  // Step t (all eight waves run the same schedule; a wave issues in order):
  //   phase 1: MFMAs 0..11 of step t; the wave's 4..5 DMA requests for tile t + 3 go out ONE behind every second MFMA -- a request
  //            costs the wave ~15..60 issue cycles, which the 32-cycle MFMA in front of it covers; issued as one block behind the
  //            barrier they kept every wave off the matrix pipe for ~300 cycles per step (tools/wsplit_trace.py);
  //   middle : tile t + 1 has landed (own pieces: vmcnt; everybody's: barrier);
  //   phase 2: MFMAs 12..23 of step t with the 24 transposing reads of step t + 1's fragments pinned two behind each MFMA (one
  //            burst of 24 fills the LDS queue of all eight waves at once and stalls every wave's MFMA issue behind its own reads).
  // Stage (t + 3) % STG was last read in phase 2 of step t - 2, which every wave has left before anyone passes the barrier of t - 1.
  auto step = [&](const frag_t (&fa)[NP][2], const frag_t (&fb)[NP][2], frag_t (&na)[NP][2], frag_t (&nb)[NP][2], int t) {
    const bool more = t + 3 < nt && !(CLICA_WSPLIT_ABLATE & 1);
    WS_STEP_STAMP(t, 4); WS_STEP_STAMP(t - 1, 10);
    static_assert(NM / 4 >= NJ, "one DMA request behind every second MFMA of the first half");
#pragma unroll
    for (int q = 0; q < NM / 4; ++q) {
      mfma1(fa, fb, 2 * q); mfma1(fa, fb, 2 * q + 1);
      __builtin_amdgcn_sched_barrier(0);
      if (q < NJ - 1) { if (more) issue_one(t + 3, q); }
      else if (q == NJ - 1) { if (more && five) issue_one(t + 3, NJ - 1); }
      __builtin_amdgcn_sched_barrier(0);
    }
    WS_STEP_STAMP(t, 5);
    if (t + 1 < nt) {
      wait_steps((CLICA_WSPLIT_ABLATE & 1) ? 0 : min(t + 3, nt - 1) - (t + 1));
      WS_STEP_STAMP(t, 6);
      __syncthreads();
      WS_STEP_STAMP(t, 7);
#if !(CLICA_WSPLIT_ABLATE & 4)
      load_frags(na, nb, t + 1);
#endif
#pragma unroll
      for (int m = NM / 2; m < NM; ++m) mfma1(fa, fb, m);
      constexpr int RPM = (8 * NP + NM / 2 - 1) / (NM / 2);      // transposing reads per MFMA: 8 NP reads over the second half's NM / 2 MFMAs (2 / 3)
#pragma unroll
      for (int i = 0; i < NM / 2; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);    // RPM DS reads
      }
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int m = NM / 2; m < NM; ++m) mfma1(fa, fb, m);
    }
    WS_STEP_STAMP(t, 8);
  };

  if (nt > 0) issue(0);
  if (nt > 1) issue(1);
  if (nt > 2) issue(2);
  wait_steps(nt > 2 ? 2 : nt - 1);
  __syncthreads();
  WS_STAMP(1);
  if (nt > 0) load_frags(fa0, fb0, 0);
  int t = 0;
  for (; t + 1 < nt; t += 2) {
    step(fa0, fb0, fa1, fb1, t);
    step(fa1, fb1, fa0, fb0, t + 1);
  }
  if (t < nt) step(fa0, fb0, fa1, fb1, t);
  WS_STAMP(2);

  // slab epilogue (same layout as the fp32 kernel: accumulator row = (r & 3) + 8 (r >> 2) + 4 h, column = lane & 31)
  const int m0 = by * BM, n0 = bx * BN;
  if constexpr (FUSED) {
    if constexpr (AR == 0) fused_epilogue<2>(g, acc, m0 + wm * 64, n0 + wn * 64, lane);
    else if constexpr (FUSED >= 16) fused_epilogue16_fast<2, FUSED>(g, acc, m0 + wm * 64, n0 + wn * 64, lane);
    else fused_epilogue16<2>(g, acc, m0 + wm * 64, n0 + wn * 64, lane);
    return;
  }
  float* Cbase = g.C + (int64_t)bz * g.M * g.ldc;
  float unscale = 1.f, unscale_db = 1.f;
  if constexpr (AR == 1) { unscale = 1.f / (*g.scaleA * *g.scaleB); unscale_db = 1.f / *g.scaleA; }      // powers of two: exact; the constant-1 feature of X is stored as 1.0, not as X's scale
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
      const bool is_db = g.dbslab && col == g.N;          // the constant-1 feature of X: db = dZ^T 1
      if (col >= g.N && !is_db) continue;
      float* dst = is_db ? g.dbslab + (int64_t)bz * g.M : Cbase + col;
      const int64_t ld = is_db ? 1 : g.ldc;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= g.M) continue;
        dst[row * ld] = AR == 1 ? acc[i][j][r] * (is_db ? unscale_db : unscale) : acc[i][j][r];
      }
    }
  }
  WS_STAMP(3);
  WS_FLUSH;
}

// ---- 256 x 256 output tile (problems with both dimensions beyond 128) -------------------------------------------------------------
// Why: both bodies move their operands L2 -> LDS at the rate the memory side delivers to a CU (measured ~17 B/clk/CU with all 256 CUs
// pulling: 36 KB per 16-row step of the 256 x 128 tile = ~2100 cycles for 1536 cycles of MFMA work, tools/wsplit_trace.py); a
// 256 x 256 tile does twice the MFMAs on 48 KB, i.e. 2/3 of the bytes per flop.
// Eight waves as 2 (rows) x 4 (columns), wave tile 128 x 64 = 4 x 2 accumulator blocks (128 registers), so the fragments of a
// whole step no longer fit twice: a step is two halves over the wave's A units {0, 1} and {2, 3} with the same B fragments:
//   H0(t): 24 MFMAs on A-half 0 / B(t), reading A-half 1 of tile t;   [tile t + 1 landed: vmcnt + barrier]
//   H1(t): 24 MFMAs on A-half 1 / B(t), requesting tile t + 3 (6 pieces per wave) and reading A-half 0 and B of tile t + 1.
// Registers: 128 accumulators + 2 x 24 (A halves) + 2 x 24 (B, ping-pong across steps).  Three 48 KB stages: stage t % 3 is last
// read in H0(t), free behind barrier(t), refilled by the requests of H1(t) with tile t + 3, first needed at barrier(t + 2).
constexpr int STG2 = 3;
static_assert((size_t)STG2 * 3 * (UW + UW) * 1024 <= kLdsBytes && (size_t)STG2 * 2 * (UW + UW) * 1024 <= lds_bytes_of<1>(), "the big-tile stages must fit the launch's LDS");
template <int FUSED, int AR>
__device__ __forceinline__ void body_big(const Prob& g, const int bx, const int by, const int bz) {
  constexpr int NP = WArith<AR>::NP, NPROD = WArith<AR>::NPROD, PIECES2 = NP * (UW + UW), STAGE2_BYTES = PIECES2 * 1024;
  constexpr int NJ2 = PIECES2 / 8;                            // DMA pieces per wave and step (6 / 4)
  constexpr int NMH = NPROD * 4;                              // MFMAs per half step

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3, h = lane >> 5, l31 = lane & 31;
  const int g0 = bz * g.gps;
  const int nt = min(g.groups, g0 + g.gps) - g0;
  WS_DECL;
  WS_STAMP(0);
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // six DMA pieces per wave and step: piece pi = wave + 8 j of the stage image [A: 8 units x 3 planes][B: 8 units x 3 planes]
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
  const char* src[NJ2]; int stride[NJ2];
#pragma unroll
  for (int j = 0; j < NJ2; ++j) {
    const int pi = wave + 8 * j;
    const bool is_a = pi < NP * UW;
    const int q = is_a ? pi : pi - NP * UW;
    const int unit = q / NP, plane = q - NP * unit;
    const int u = (is_a ? by : bx) * UW + unit, fu = is_a ? g.fuA : g.fuB;
    const bool ok = u < fu;
    const char* base = is_a ? g.A : g.B;
    src[j] = ok ? base + (((int64_t)g0 * fu + u) * NP + plane) * 1024 + lane * 16 : reinterpret_cast<const char*>(g_zero16);
    stride[j] = ok ? fu * NP * planes::kPieceBytes : 0;
  }
  auto issue_one = [&](int t, int j) { dma_1k(src[j], lds0 + (unsigned)((t % STG2) * STAGE2_BYTES + (wave + 8 * j) * 1024)); src[j] += stride[j]; };
  const int lane_off = h * 512 + ((lane >> 4) & 1) * 128 + (lane & 15) * 8;
  const char* fa_base = smem + lane_off + (wm * 4) * NP * 1024;
  const char* fb_base = smem + lane_off + (UW + wn * 2) * NP * 1024;
  frag_t a0[NP][2], a1[NP][2], b0[NP][2], b1[NP][2];        // [plane][unit of the half / of the wave's two B units]
  auto load_a = [&](frag_t (&f)[NP][2], int t, int half) {
    const int so = (t % STG2) * STAGE2_BYTES;
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) f[p][i] = read_frag(fa_base + so + (NP * (2 * half + i) + p) * 1024);
  };
  auto load_b = [&](frag_t (&f)[NP][2], int t) {
    const int so = (t % STG2) * STAGE2_BYTES;
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int j = 0; j < 2; ++j) f[p][j] = read_frag(fb_base + so + (NP * j + p) * 1024);
  };
  auto mfma1 = [&](const frag_t (&fa)[NP][2], const frag_t (&fb)[NP][2], int half, int m) {     // m = 0..NMH-1 of a half
    const int t = m >> 2, i = (m >> 1) & 1, j = m & 1;
    acc[2 * half + i][j] = mfma32<AR>(fa[WArith<AR>::PA[t]][i], fb[WArith<AR>::PB[t]][j], acc[2 * half + i][j]);
  };
  auto wait_tiles = [&](int k) {            // at most k of the most recently requested tiles (NJ2 pieces each) may be in flight
    if (k <= 0) wait_vm<0>(); else if (k == 1) wait_vm<NJ2>(); else wait_vm<2 * NJ2>();
  };
  auto step = [&](const frag_t (&fb)[NP][2], frag_t (&nb)[NP][2], int t) {
    WS_STEP_STAMP(t, 4);
    load_a(a1, t, 1);
#pragma unroll
    for (int m = 0; m < NMH; ++m) mfma1(a0, fb, 0, m);
#pragma unroll
    for (int i = 0; i < NMH / 2; ++i) {                        // the half's 4 NP transposing reads spread over its MFMAs
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // 2 MFMAs
      __builtin_amdgcn_sched_group_barrier(0x100, AR == 0 ? 1 : 2, 0);      // DS reads
    }
    __builtin_amdgcn_sched_barrier(0);
    const bool next = t + 1 < nt;
    WS_STEP_STAMP(t, 5);
    if (next) {
      wait_tiles((CLICA_WSPLIT_ABLATE & 1) ? 0 : min(t + 2, nt - 1) - (t + 1));
      WS_STEP_STAMP(t, 6);
#if !(CLICA_WSPLIT_ABLATE & 8)
      __syncthreads();
#endif
    }
    WS_STEP_STAMP(t, 7);
    const bool more = t + 3 < nt && !(CLICA_WSPLIT_ABLATE & 1);
    constexpr int NG = 2 * NP, MPG = NMH / NG;                 // groups of the half: one piece's two fragments (four reads) and MPG MFMAs each (4 / 3)
    static_assert(NG >= NJ2 && NMH % NG == 0, "one DMA request per group");
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      if (next) {
        if (q < NP) {                                       // A-half 0 of tile t + 1: two fragments (four reads) per group
#pragma unroll
          for (int i = 0; i < 2; ++i) a0[q][i] = read_frag(fa_base + ((t + 1) % STG2) * STAGE2_BYTES + (NP * i + q) * 1024);
        } else {
#pragma unroll
          for (int j = 0; j < 2; ++j) nb[q - NP][j] = read_frag(fb_base + ((t + 1) % STG2) * STAGE2_BYTES + (NP * j + (q - NP)) * 1024);
        }
      }
#pragma unroll
      for (int m = MPG * q; m < MPG * q + MPG; ++m) mfma1(a1, fb, 1, m);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more && q < NJ2) issue_one(t + 3, q);
      __builtin_amdgcn_sched_barrier(0);
    }
    WS_STEP_STAMP(t, 8);
    WS_STEP_STAMP(t - 1, 10);
  };
  // note: a0 of step t + 1 is loaded in H1(t) while H1(t) computes on a1 -- a0 (tile t) is dead behind H0(t)
#pragma unroll
  for (int tt = 0; tt < 3; ++tt)
    if (tt < nt) {
#pragma unroll
      for (int j = 0; j < NJ2; ++j) issue_one(tt, j);
    }
  wait_tiles(nt > 2 ? 2 : nt - 1);
  __syncthreads();
  WS_STAMP(1);
  if (nt > 0) { load_a(a0, 0, 0); load_b(b0, 0); }
  int t = 0;
  for (; t + 1 < nt; t += 2) { step(b0, b1, t); step(b1, b0, t + 1); }
  if (t < nt) step(b0, b1, t);
  WS_STAMP(2);

  const int m0 = by * 256, n0 = bx * 256;
  if constexpr (FUSED) {
    if constexpr (AR == 0) fused_epilogue<4>(g, acc, m0 + wm * 128, n0 + wn * 64, lane);
    else if constexpr (FUSED >= 16) fused_epilogue16_fast<4, FUSED>(g, acc, m0 + wm * 128, n0 + wn * 64, lane);
    else fused_epilogue16<4>(g, acc, m0 + wm * 128, n0 + wn * 64, lane);
    return;
  }
  float* Cbase = g.C + (int64_t)bz * g.M * g.ldc;
  float unscale = 1.f, unscale_db = 1.f;
  if constexpr (AR == 1) { unscale = 1.f / (*g.scaleA * *g.scaleB); unscale_db = 1.f / *g.scaleA; }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
      const bool is_db = g.dbslab && col == g.N;
      if (col >= g.N && !is_db) continue;
      float* dst = is_db ? g.dbslab + (int64_t)bz * g.M : Cbase + col;
      const int64_t ld = is_db ? 1 : g.ldc;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= g.M) continue;
        dst[row * ld] = AR == 1 ? acc[i][j][r] * (is_db ? unscale_db : unscale) : acc[i][j][r];
      }
    }
  }
  WS_STAMP(3);
  WS_NOTE(11, 1000 + nt);
  WS_FLUSH;
}

__device__ __forceinline__ int xcd_contiguous(int b, int nwg) {     // each XCD walks a contiguous range of the work items
  const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

template <int AR>
__global__ __launch_bounds__(THREADS) void wgrad_split_k(GroupArgs G) {
  const int id = xcd_contiguous(blockIdx.x, G.total);
  int q = 0;
#pragma unroll
  for (int i = 1; i < MAXG; ++i) q += (i < G.n && id >= G.first[i]) ? 1 : 0;
  const Prob& g = G.p[q];
  const int local = id - G.first[q];
  const int tiles = g.gx * g.gy;
  const int bz = local / tiles, t = local - bz * tiles;
  const int by = t / g.gx, bx = t - by * g.gx;
  if (g.a_wide == 2) body_big<false, AR>(g, bx, by, bz); else if (g.a_wide) body<true, false, AR>(g, bx, by, bz); else body<false, false, AR>(g, bx, by, bz);
}

// ---- forward / data gradient of ONE wide layer on the same bodies -------------------------------------------------------------------
// BASELINE config 3 (n = 40: 2000-wide layers, main_mlp.py:297-307) does not fit the whole-stack kernel (mlp_split_k keeps a 48-row
// activation panel of <= 512 features in LDS); its per-layer GEMMs ran on the fp32 matrix cores (linear.hip: gemm_k, 0.70-0.76 of
// the 157 TFLOP/s fp32 peak).  Both are contractions this file's bodies already do when the operands are given TRANSPOSED:
//   forward   Y[m][n]  = sum_k X[m][k] W[n][k]  : A = planes of X^T (rows = k, features = m),  B = planes of W^T (rows = k, features = n)
//   backward  dX[m][k] = sum_n dZ[m][n] W[n][k] : A = planes of dZ^T (rows = n, features = m), B = planes of W   (rows = n, features = k)
// no contraction split (K = 400..2000 is 25..125 steps of a 256 x 256 tile), fused_epilogue instead of the slab.
// Mixed tiling (a_wide == 3): 12288 x 2000 is 384 tiles of 256 x 256 -- 1.5 rounds of the 256 CUs, and the half-empty second round
// costs more than the big tile's 2/3 bytes per flop save (572 us against 511 us for 768 tiles of 128 x 256).  So ONE full round of big
// tiles over the first rows_big rows (the first 256 workgroups, one per CU) and the remaining rows as 128 x 256 tiles, picked up as
// the CUs come free: 286 + 170 us of work per CU instead of 3 x 170.
template <int AR, int EPI = 1>
__global__ __launch_bounds__(THREADS) void gemm_split_k(GroupArgs G) {
  const Prob& g = G.p[0];
  if (g.a_wide == 3) {
    const int b = (int)blockIdx.x;
    if (b < g.n_big) {
      const int id = xcd_contiguous(b, g.n_big);
      const int by = id / g.gx, bx = id - by * g.gx;
      body_big<EPI, AR>(g, bx, by, 0);
    } else {
      const int id = xcd_contiguous(b - g.n_big, G.total - g.n_big);
      const int by = id / g.gx, bx = id - by * g.gx;
      body<false, EPI, AR>(g, bx, g.rows_big / 128 + by, 0);
    }
    return;
  }
  const int id = xcd_contiguous(blockIdx.x, G.total);
  const int by = id / g.gx, bx = id - by * g.gx;
  if (g.a_wide == 2) body_big<EPI, AR>(g, bx, by, 0); else if (g.a_wide) body<true, EPI, AR>(g, bx, by, 0); else body<false, EPI, AR>(g, bx, by, 0);
}

#ifdef CLICA_WSPLIT_TRACE
extern "C" int clica_debug_wsplit_trace(unsigned long long* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(clica::wsplit::g_wstrace), &buf, sizeof(buf));
}
#endif

// ---- fp32 [M][F] -> bf16 planes (planes.h) -------------------------------------------------------------------------------
// For operands no split whole-stack kernel produced: the per-layer fp32 kernels of wide encoders (a width beyond 512: BASELINE
// config 3's 2000-wide layers), the encoder input x, the loss gradient dY.  One thread = four consecutive features of one row
// (one float4 in, three 8-byte pieces out -- the store pattern of the mlp_split_k epilogue); HBM-bound, 4 B read + 6 B
// written per element.  Rows beyond M and features beyond F are written as zeros (feature F as 1 when `ones`).
__device__ __forceinline__ void split3_hi(float v, unsigned& hb, unsigned& mb, unsigned& lb) {     // piece bits in the HIGH half
  hb = __float_as_uint(v) & 0xFFFF0000u;
  const float r1 = v - __uint_as_float(hb);
  mb = __float_as_uint(r1) & 0xFFFF0000u;
  const float r2 = r1 - __uint_as_float(mb);
  lb = __float_as_uint(r2) & 0xFFFF0000u;
}
__global__ __launch_bounds__(256) void planes_from_f32_k(const float* __restrict__ X, int64_t ldx, int64_t M, int F, int ones, int units,
                                                         char* __restrict__ out, int64_t groups) {
  // thread -> (group g, unit u, key k, feature quad q): 16 keys x 8 quads = 128 threads per unit, two units per workgroup
  const int64_t wu = (int64_t)blockIdx.x * 2 + (threadIdx.x >> 7);
  if (wu >= groups * units) return;
  const int64_t g = wu / units; const int u = (int)(wu - g * units);
  const int t = threadIdx.x & 127, k = t >> 3, q = t & 7;
  const int64_t row = g * 16 + k;
  const int f0 = u * 32 + q * 4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (row < M) {
    const float* src = X + row * ldx + f0;
    if (f0 + 3 < F && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
      const float4 x4 = *reinterpret_cast<const float4*>(src);
      v[0] = x4.x; v[1] = x4.y; v[2] = x4.z; v[3] = x4.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (f0 + e < F) v[e] = src[e];
    }
  }
  if (ones) {
#pragma unroll
    for (int e = 0; e < 4; ++e) if (f0 + e == F) v[e] = 1.f;       // every row of the group, padded rows included (they meet zero rows)
  }
  unsigned hb[4], mb[4], lb[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split3_hi(v[e], hb[e], mb[e], lb[e]);
  const int s = q >> 2, c = (q & 3) * 4;                           // 16-feature half of the unit, first feature inside it
  char* dst = out + ((g * units + u) * 3) * 1024 + (k >> 2) * 256 + s * 128 + (k & 3) * 32 + c * 2;
  *reinterpret_cast<u32x2*>(dst) = (u32x2){(hb[0] >> 16) | hb[1], (hb[2] >> 16) | hb[3]};
  *reinterpret_cast<u32x2*>(dst + 1024) = (u32x2){(mb[0] >> 16) | mb[1], (mb[2] >> 16) | mb[3]};
  *reinterpret_cast<u32x2*>(dst + 2048) = (u32x2){(lb[0] >> 16) | lb[1], (lb[2] >> 16) | lb[3]};
}

// fp32 X[M][F] -> planes of X^T (rows = feature index of X, features = row index of X): the A operand of the first split
// forward / backward GEMM of a chain whose producer was an fp32 kernel.  One 16 x 32 unit per 128 threads through a padded
// LDS tile: reads are 64-byte row segments of X, writes the usual 8 bytes per plane and thread.
__global__ __launch_bounds__(256) void planes_from_f32_t_k(const float* __restrict__ X, int64_t ldx, int64_t M, int F, int units,
                                                           char* __restrict__ out, int64_t groups) {
  __shared__ float tile[2][16][33];
  const int half = threadIdx.x >> 7, t = threadIdx.x & 127;
  const int64_t wu = (int64_t)blockIdx.x * 2 + half;
  const bool live = wu < groups * units;
  const int64_t g = live ? wu / units : 0; const int u = live ? (int)(wu - g * units) : 0;
  {
    const int phi = t >> 2, c = (t & 3) * 4;                      // row of X inside the unit, first of four features
    const int64_t row = (int64_t)u * 32 + phi; const int64_t f0 = g * 16 + c;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (live && row < M) {
      const float* src = X + row * ldx + f0;
      if (f0 + 3 < F && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
        const float4 x4 = *reinterpret_cast<const float4*>(src);
        v[0] = x4.x; v[1] = x4.y; v[2] = x4.z; v[3] = x4.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (f0 + e < F) v[e] = src[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[half][c + e][phi] = v[e];
  }
  __syncthreads();
  if (!live) return;
  const int k = t >> 3, q = t & 7;
  unsigned hb[4], mb[4], lb[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split3_hi(tile[half][k][q * 4 + e], hb[e], mb[e], lb[e]);
  const int sidx = q >> 2, c = (q & 3) * 4;
  char* dst = out + ((g * units + u) * 3) * 1024 + (k >> 2) * 256 + sidx * 128 + (k & 3) * 32 + c * 2;
  *reinterpret_cast<u32x2*>(dst) = (u32x2){(hb[0] >> 16) | hb[1], (hb[2] >> 16) | hb[3]};
  *reinterpret_cast<u32x2*>(dst + 1024) = (u32x2){(mb[0] >> 16) | mb[1], (mb[2] >> 16) | mb[3]};
  *reinterpret_cast<u32x2*>(dst + 2048) = (u32x2){(lb[0] >> 16) | lb[1], (lb[2] >> 16) | lb[3]};
}

// ---- the same two conversions in the f16x2 arithmetic: v x scale -> hi / lo fp16 pieces (2 KB units), the constant-1 feature as 1.0;
//      the workgroup's max |v| goes to the tensor's slots (split16.h) for the next step's scale ----
__global__ __launch_bounds__(256) void planes16_from_f32_k(const float* __restrict__ X, int64_t ldx, int64_t M, int F, int ones, int units,
                                                           char* __restrict__ out, int64_t groups, const s16::S16Tensor T) {
  __shared__ float red[8];
  const int64_t wu = (int64_t)blockIdx.x * 2 + (threadIdx.x >> 7);
  const bool live = wu < groups * units;
  const int64_t g = live ? wu / units : 0; const int u = live ? (int)(wu - g * units) : 0;
  const int t = threadIdx.x & 127, k = t >> 3, q = t & 7;
  const int64_t row = g * 16 + k;
  const int f0 = u * 32 + q * 4;
  const float sc = *T.scale;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (live && row < M) {
    const float* src = X + row * ldx + f0;
    if (f0 + 3 < F && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
      const float4 x4 = *reinterpret_cast<const float4*>(src);
      v[0] = x4.x; v[1] = x4.y; v[2] = x4.z; v[3] = x4.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (f0 + e < F) v[e] = src[e];
    }
  }
  float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
  unsigned hb[4], lb[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    split16_hi_lo(v[e] * sc, hb[e], lb[e]);
    if (ones && f0 + e == F) { hb[e] = 0x3C00u; lb[e] = 0u; }       // every row of the group, padded rows included (they meet zero rows)
  }
  if (live) {
    const int s2 = q >> 2, c = (q & 3) * 4;
    char* dst = out + ((g * units + u) * 2) * 1024 + (k >> 2) * 256 + s2 * 128 + (k & 3) * 32 + c * 2;
    *reinterpret_cast<u32x2*>(dst) = (u32x2){hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16)};
    *reinterpret_cast<u32x2*>(dst + 1024) = (u32x2){lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16)};
  }
  s16::s16_commit_block_max(T, amax, red, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256) void planes16_from_f32_t_k(const float* __restrict__ X, int64_t ldx, int64_t M, int F, int units,
                                                             char* __restrict__ out, int64_t groups, const s16::S16Tensor T) {
  __shared__ float tile[2][16][33];
  __shared__ float red[8];
  const int half = threadIdx.x >> 7, t = threadIdx.x & 127;
  const int64_t wu = (int64_t)blockIdx.x * 2 + half;
  const bool live = wu < groups * units;
  const int64_t g = live ? wu / units : 0; const int u = live ? (int)(wu - g * units) : 0;
  const float sc = *T.scale;
  float amax = 0.f;
  {
    const int phi = t >> 2, c = (t & 3) * 4;
    const int64_t row = (int64_t)u * 32 + phi; const int64_t f0 = g * 16 + c;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (live && row < M) {
      const float* src = X + row * ldx + f0;
      if (f0 + 3 < F && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
        const float4 x4 = *reinterpret_cast<const float4*>(src);
        v[0] = x4.x; v[1] = x4.y; v[2] = x4.z; v[3] = x4.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (f0 + e < F) v[e] = src[e];
      }
    }
    amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[half][c + e][phi] = v[e];
  }
  __syncthreads();
  if (live) {
    const int k = t >> 3, q = t & 7;
    unsigned hb[4], lb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split16_hi_lo(tile[half][k][q * 4 + e] * sc, hb[e], lb[e]);
    const int sidx = q >> 2, c = (q & 3) * 4;
    char* dst = out + ((g * units + u) * 2) * 1024 + (k >> 2) * 256 + sidx * 128 + (k & 3) * 32 + c * 2;
    *reinterpret_cast<u32x2*>(dst) = (u32x2){hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16)};
    *reinterpret_cast<u32x2*>(dst + 1024) = (u32x2){lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16)};
  }
  s16::s16_commit_block_max(T, amax, red, blockIdx.x, gridDim.x);
}

// ---- plan: which body per layer, how many contraction splits -----------------------------------------------------------------
struct Plan { int splits, gps, groups, tiles, n_tiny, tiny_splits; int64_t tiny_kps; int big_tiles; };
// tile shape of a problem: 2 = 256 x 256 when both dimensions need more than one 128-wide tile, else 256 x 128 / 128 x 256 along
// the longer side
static void tile_shape(int32_t N, int32_t K, int* kind, int* gx, int* gy, bool with_db) {
  const int cols = K + (with_db ? 1 : 0);
  if (N > 128 && cols > 128) { *kind = 2; *gy = (int)ceil_div(N, 256); *gx = (int)ceil_div(cols, 256); return; }
  *kind = N >= K ? 1 : 0;
  const int bm = *kind ? 256 : 128, bn = *kind ? 128 : 256;
  *gy = (int)ceil_div(N, bm);
  *gx = (int)ceil_div(cols, bn);
}
// A 256 x 256 item does twice the MFMAs of a 256 x 128 item per 16-row step: big-tile problems get TWICE the contraction splits
// (half the steps), so that every item of the launch carries the same matrix work.  `splits` / `gps` are the small-tile figures.
static int problem_splits(const Plan& p, int kind) { return kind == 2 ? (int)ceil_div(p.groups, (int64_t)std::max(1, p.gps / 2)) : p.splits; }
static int problem_gps(const Plan& p, int kind) { return kind == 2 ? std::max(1, p.gps / 2) : p.gps; }
static Plan make_plan(int64_t Mrows, int n, const int32_t* N, const int32_t* K) {
  Plan p{};
  p.groups = (int)planes::groups_used(Mrows);
  for (int l = 0; l < n; ++l) {
    if (gemm::wgrad_tiny_shape(N[l], K[l])) { ++p.n_tiny; continue; }
    int kind, gx, gy; tile_shape(N[l], K[l], &kind, &gx, &gy, true);
    if (kind == 2) p.big_tiles += gx * gy; else p.tiles += gx * gy;
  }
  // rounds of equal items (gps x 16 rows of a small tile = gps / 2 x 16 rows of a big tile) + a fixed prologue / slab epilogue
  // per round + slab traffic per split (same cost model as the fp32 plan, linear.hip: plan_wgrad_group)
  const int max_s = std::max(1, std::min(64, p.groups / (p.big_tiles ? 16 : 8)));
  double best = 1e300;
  p.splits = 1; p.gps = p.groups;
  for (int s = 1; s <= max_s && (p.tiles + p.big_tiles) > 0; ++s) {
    int gps = (int)ceil_div(p.groups, s);
    if (p.big_tiles) gps = (gps + 1) & ~1;                    // even: the big tiles take exactly half
    const int sp = (int)ceil_div(p.groups, gps);
    const int sp_big = p.big_tiles ? (int)ceil_div(p.groups, gps / 2) : 0;
    const int64_t items = (int64_t)p.tiles * sp + (int64_t)p.big_tiles * sp_big;
    const int64_t rounds = ceil_div(items, kNumCU);
    const double cost = (double)rounds * (16.0 * gps + 48.0) + 8.0 * sp + (p.big_tiles ? 4.0 * sp_big : 0.0);
    if (cost < best) { best = cost; p.splits = sp; p.gps = gps; }
  }
  if (p.n_tiny > 0) gemm::wgrad_tiny_plan(Mrows, p.n_tiny, &p.tiny_splits, &p.tiny_kps);
  return p;
}
static int layer_splits(const Plan& p, int32_t N, int32_t K) {
  if (gemm::wgrad_tiny_shape(N, K)) return p.tiny_splits;
  int kind, gx, gy; tile_shape(N, K, &kind, &gx, &gy, true);
  return problem_splits(p, kind);
}
bool chain_tail_supported(int n_layers, const int32_t* N, const int32_t* K) {
  if (n_layers < 2) return false;
  const int l = n_layers - 1;
  return gemm::wgrad_tiny_shape(N[0], K[0]) && gemm::wgrad_tiny_shape(N[l], K[l]) && K[0] <= 15 && N[0] <= 128 && N[l] <= 16 && K[l] <= 127;
}
static int tail_splits(int64_t Mrows) { return (int)ceil_div(Mrows, (int64_t)kChainRows); }
// slabs a layer's region of the workspace holds: the plan's own splits, or -- first / last layer of an encoder whose chain can leave
// their slabs (one per chain workgroup) -- whichever of the two is larger, so that the layout does not depend on who fills it
static size_t slabs_alloc(const Plan& p, int64_t Mrows, int n, const int32_t* N, const int32_t* K, int l) {
  size_t sp = (size_t)layer_splits(p, N[l], K[l]);
  if ((l == 0 || l == n - 1) && chain_tail_supported(n, N, K)) sp = std::max(sp, (size_t)tail_splits(Mrows));
  return sp;
}
static size_t ws_layout(const Plan& p, int64_t Mrows, int n, const int32_t* N, const int32_t* K, size_t* slab_off, size_t* db_off) {
  size_t off = 0;
  for (int l = 0; l < n; ++l) {
    const size_t sp = slabs_alloc(p, Mrows, n, N, K, l);
    if (slab_off) slab_off[l] = off;
    off += align_up(sp * N[l] * K[l] * sizeof(float), 256);
    if (db_off) db_off[l] = off;
    off += align_up(sp * N[l] * sizeof(float), 256);
  }
  return off;
}
int chain_tail_slabs(int64_t M, int n_layers, const int32_t* N, const int32_t* K, void* workspace, size_t workspace_bytes,
                     float** slab_first, float** db_first, float** slab_last, float** db_last) {
  if (!chain_tail_supported(n_layers, N, K) || n_layers > MAXG) { set_error("chain tail: these layer shapes are not supported"); return CLICA_E_INVALID; }
  size_t slab_off[MAXG], db_off[MAXG];
  const size_t need = ws_layout(make_plan(M, n_layers, N, K), M, n_layers, N, K, slab_off, db_off);
  if (!workspace || need > workspace_bytes) { set_error("chain tail: weight-gradient workspace %zu < %zu", workspace_bytes, need); return CLICA_E_WORKSPACE; }
  char* w = reinterpret_cast<char*>(workspace);
  *slab_first = reinterpret_cast<float*>(w + slab_off[0]); *db_first = reinterpret_cast<float*>(w + db_off[0]);
  *slab_last = reinterpret_cast<float*>(w + slab_off[n_layers - 1]); *db_last = reinterpret_cast<float*>(w + db_off[n_layers - 1]);
  return CLICA_OK;
}

}  // namespace wsplit
}  // namespace clica

using namespace clica;
using namespace clica::wsplit;

extern "C" int clica_mlp_planes_from_f32(const float* X, int64_t ldx, int64_t M, int32_t width, int32_t ones_column, void* planes_out,
                                         clica_stream_t stream) {
  CLICA_CHECK_ARG(X && planes_out && M > 0 && width >= 1 && ldx >= width, "clica_mlp_planes_from_f32: bad argument");
  CLICA_CHECK_ARG((reinterpret_cast<uintptr_t>(planes_out) & 15) == 0, "clica_mlp_planes_from_f32: plane buffer must be 16-byte aligned");
  const int units = planes::units(width, ones_column);
  const int64_t groups = planes::groups_alloc(M);      // every group of the buffer is written (groups beyond the batch: zeros)
  hipLaunchKernelGGL(planes_from_f32_k, dim3((unsigned)ceil_div(groups * units, 2)), dim3(256), 0, as_stream(stream),
                     X, ldx, M, (int)width, ones_column ? 1 : 0, units, reinterpret_cast<char*>(planes_out), groups);
  return launch_status("clica_mlp_planes_from_f32");
}

extern "C" int clica_mlp_wgrad_split_kind(int32_t N, int32_t K, int32_t* kind) {
  CLICA_CHECK_ARG(kind && N >= 1 && K >= 1, "clica_mlp_wgrad_split_kind: bad argument");
  *kind = gemm::wgrad_tiny_shape(N, K) ? 1 : 0;
  return CLICA_OK;
}

extern "C" int clica_mlp_wgrad_split_workspace_bytes(int64_t M, int32_t n_layers, const int32_t* N, const int32_t* K, size_t* bytes) {
  CLICA_CHECK_ARG(bytes && N && K && M > 0 && n_layers >= 1 && n_layers <= MAXG, "clica_mlp_wgrad_split_workspace_bytes: bad argument");
  for (int l = 0; l < n_layers; ++l) CLICA_CHECK_ARG(N[l] >= 1 && K[l] >= 1, "clica_mlp_wgrad_split_workspace_bytes: layer %d: bad size", l);
  *bytes = ws_layout(make_plan(M, n_layers, N, K), M, n_layers, N, K, nullptr, nullptr);
  return CLICA_OK;
}

// the layout of the f16x2 state's scale arrays (fused_mlp.hip: Split16State) as far as this file needs it
typedef s16::Split16State Split16Scales;

static int mlp_wgrad_split_impl(int64_t M, int32_t n_layers, const void* const* dZ_planes, const void* const* X_planes,
                                const float* const* dZ, const int64_t* lddz, const float* const* X, const int64_t* ldx,
                                float* const* dW, const int64_t* lddw, float* const* db, const int32_t* N, const int32_t* K,
                                int32_t accumulate, const void* state16, const int32_t* a_index, const int32_t* d_index,
                                void* workspace, size_t workspace_bytes, clica_stream_t stream, const clica_adam_desc* adam = nullptr,
                                int tail_slabs = 0) {
  CLICA_CHECK_ARG(dZ_planes && X_planes && dZ && lddz && X && ldx && dW && lddw && db && N && K && workspace && M > 0,
                  "clica_mlp_wgrad_split: NULL pointer / empty batch");
  CLICA_CHECK_ARG(n_layers >= 1 && n_layers <= MAXG, "clica_mlp_wgrad_split: %d layers (1..%d supported)", n_layers, MAXG);
  CLICA_CHECK_ARG(!tail_slabs || chain_tail_supported(n_layers, N, K), "clica_mlp_wgrad_split_adam: tail_slabs set for shapes the chain's tail does not cover");
  const Plan p = make_plan(M, n_layers, N, K);
  size_t slab_off[MAXG], db_off[MAXG];
  const size_t need = ws_layout(p, M, n_layers, N, K, slab_off, db_off);
  if (need > workspace_bytes) { set_error("clica_mlp_wgrad_split: workspace %zu < %zu", workspace_bytes, need); return CLICA_E_WORKSPACE; }
  hipStream_t st = as_stream(stream);
  GroupArgs G{};
  gemm::ReduceGroupArgs R{};
  gemm::TinyArgs T{};
  R.n = n_layers; R.accumulate = accumulate ? 1 : 0;
  if (adam) {      // the optimizer in the reduction's epilogue (clica_mlp_wgrad_split_adam)
    CLICA_CHECK_ARG(adam->param && adam->grad && adam->exp_avg && adam->exp_avg_sq && adam->step_dev && adam->count > 0 && !accumulate,
                    "clica_mlp_wgrad_split_adam: NULL arena / accumulate set");
    CLICA_CHECK_ARG((((uintptr_t)adam->param | (uintptr_t)adam->grad | (uintptr_t)adam->exp_avg | (uintptr_t)adam->exp_avg_sq) & 15) == 0,
                    "clica_mlp_wgrad_split_adam: arenas must be 16-byte aligned");
    CLICA_CHECK_ARG(adam->t_offset == 0 || adam->t_offset == 1, "clica_mlp_wgrad_split_adam: t_offset must be 0 or 1");
    CLICA_CHECK_ARG(!adam->split16_state || (adam->n_layers >= 1 && adam->n_layers <= 8), "clica_mlp_wgrad_split_adam: bad layer count");
    int64_t covered = 0;
    for (int l = 0; l < n_layers; ++l) {
      CLICA_CHECK_ARG(dW[l] && db[l] && lddw[l] == K[l], "clica_mlp_wgrad_split_adam: layer %d: dW must be contiguous, db present", l);
      CLICA_CHECK_ARG(dW[l] >= adam->grad && dW[l] + (int64_t)N[l] * K[l] <= adam->grad + adam->count && db[l] >= adam->grad &&
                      db[l] + N[l] <= adam->grad + adam->count, "clica_mlp_wgrad_split_adam: layer %d: dW / db outside the gradient arena", l);
      covered += (int64_t)N[l] * K[l] + N[l];
    }
    // every parameter of the arena must be one this call produces the gradient of (alignment padding aside: <= 3 floats per tensor)
    CLICA_CHECK_ARG(adam->count - covered >= 0 && adam->count - covered <= 8 * (int64_t)n_layers,
                    "clica_mlp_wgrad_split_adam: the arena holds %lld elements, the layers cover %lld", (long long)adam->count, (long long)covered);
    R.adam.p = adam->param; R.adam.m = adam->exp_avg; R.adam.v = adam->exp_avg_sq; R.adam.gbase = adam->grad;
    R.adam.lr = adam->lr; R.adam.b1 = adam->beta1; R.adam.b2 = adam->beta2; R.adam.eps = adam->eps; R.adam.gscale = adam->grad_scale;
    R.adam.step_dev = adam->step_dev; R.adam.t_offset = adam->t_offset;
    R.adam.s16_state = adam->split16_state; R.adam.s16_layers = adam->n_layers;
  }
  int item = 0, rblock = 0, ng = 0;
  for (int l = 0; l < n_layers; ++l) {
    CLICA_CHECK_ARG(dW[l] && N[l] >= 1 && K[l] >= 1 && lddw[l] >= K[l], "clica_mlp_wgrad_split: layer %d: bad argument", l);
    const bool tiny = gemm::wgrad_tiny_shape(N[l], K[l]);
    const bool from_chain = tail_slabs && (l == 0 || l == n_layers - 1);      // the chain's tail has left this layer's slabs (one per workgroup)
    const int sp = from_chain ? tail_splits(M) : layer_splits(p, N[l], K[l]);
    float* slab = (float*)((char*)workspace + slab_off[l]);
    float* dbslab = (float*)((char*)workspace + db_off[l]);
    if (from_chain) {
      CLICA_CHECK_ARG(db[l], "clica_mlp_wgrad_split_adam: layer %d: tail slabs need a bias gradient", l);
    } else if (tiny) {
      CLICA_CHECK_ARG(dZ[l] && X[l] && lddz[l] >= N[l] && ldx[l] >= K[l],
                      "clica_mlp_wgrad_split: layer %d (%d x %d) takes the fp32 tiny-dimension kernel: fp32 operands required", l, N[l], K[l]);
      gemm::Args g{};
      g.A = dZ[l]; g.lda = lddz[l]; g.B = X[l]; g.ldb = ldx[l]; g.C = slab; g.ldc = K[l]; g.M = N[l]; g.N = K[l]; g.Kc = M;
      g.k_per_split = p.tiny_kps; g.dbias_slab = db[l] ? dbslab : nullptr;
      T.p[T.n++] = g;
    } else {
      CLICA_CHECK_ARG(dZ_planes[l] && X_planes[l], "clica_mlp_wgrad_split: layer %d (%d x %d) needs the bf16 plane copies of both "
                      "operands (clica_mlp_fwd_split / clica_mlp_dgrad_split with a plane buffer)", l, N[l], K[l]);
      CLICA_CHECK_ARG(((reinterpret_cast<uintptr_t>(dZ_planes[l]) | reinterpret_cast<uintptr_t>(X_planes[l])) & 15) == 0,
                      "clica_mlp_wgrad_split: layer %d: plane buffers must be 16-byte aligned", l);
      Prob& g = G.p[ng];
      g.A = reinterpret_cast<const char*>(dZ_planes[l]); g.fuA = planes::units(N[l], 0);
      g.B = reinterpret_cast<const char*>(X_planes[l]); g.fuB = planes::units(K[l], 1);
      g.C = slab; g.ldc = K[l]; g.dbslab = db[l] ? dbslab : nullptr; g.M = N[l]; g.N = K[l];
      // the tile shape follows the SHAPE (with the bias column: it fixed the plan and the workspace); without a bias the last
      // column tile may simply have nothing to store
      tile_shape(N[l], K[l], &g.a_wide, &g.gx, &g.gy, true);
      g.groups = p.groups; g.gps = problem_gps(p, g.a_wide);
      if (state16) {
        const Split16Scales* st16 = reinterpret_cast<const Split16Scales*>(state16);
        CLICA_CHECK_ARG(a_index && d_index && a_index[l] >= 0 && a_index[l] < 9 && d_index[l] >= 0 && d_index[l] < 9,
                        "clica_mlp_wgrad_split16: layer %d: scale index out of range", l);
        g.scaleA = &st16->sD[d_index[l]]; g.scaleB = &st16->sA[a_index[l]];
      }
      G.first[ng] = item; item += g.gx * g.gy * sp;
      ++ng;
    }
    rblock += gemm::slab_reduce_entry(R, l, rblock, sp, slab, dbslab, dW[l], lddw[l], db[l], N[l], K[l]);
  }
  G.n = ng; G.first[ng] = G.total = item; R.first[n_layers] = rblock;
  if (T.n > 0) {
    int rct = gemm::launch_wgrad_tiny(T, p.tiny_splits, st);
    if (rct) return rct;
  }
  if (ng > 0) {
    if (state16) {
      static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_split_k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes_of<1>()), true);
      (void)once;
      hipLaunchKernelGGL(wgrad_split_k<1>, dim3((unsigned)item), dim3(THREADS), lds_bytes_of<1>(), st, G);
    } else {
      static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_split_k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes), true);
      (void)once;
      hipLaunchKernelGGL(wgrad_split_k<0>, dim3((unsigned)item), dim3(THREADS), kLdsBytes, st, G);
    }
    int rc = launch_status("clica_mlp_wgrad_split");
    if (rc) return rc;
  }
  return gemm::launch_slab_reduce_group(R, rblock, st);
}

extern "C" int clica_mlp_wgrad_split(int64_t M, int32_t n_layers, const void* const* dZ_planes, const void* const* X_planes,
                                     const float* const* dZ, const int64_t* lddz, const float* const* X, const int64_t* ldx,
                                     float* const* dW, const int64_t* lddw, float* const* db, const int32_t* N, const int32_t* K,
                                     int32_t accumulate, void* workspace, size_t workspace_bytes, clica_stream_t stream) {
  return mlp_wgrad_split_impl(M, n_layers, dZ_planes, X_planes, dZ, lddz, X, ldx, dW, lddw, db, N, K, accumulate, nullptr, nullptr, nullptr,
                              workspace, workspace_bytes, stream);
}
// f16x2 plane copies (two pieces per unit, written by clica_mlp_fwd_split16 / clica_mlp_dgrad_split16 of the same `state`):
// a_index[l] / d_index[l] = positions of layer l's input activation / of dZ_l in the state's activation / chain-gradient scale arrays
// (forward order: a_index = l; chain order: d_index = L - 1 - l for an L-layer encoder)
extern "C" int clica_mlp_wgrad_split16(int64_t M, int32_t n_layers, const void* const* dZ_planes, const void* const* X_planes,
                                       const float* const* dZ, const int64_t* lddz, const float* const* X, const int64_t* ldx,
                                       float* const* dW, const int64_t* lddw, float* const* db, const int32_t* N, const int32_t* K,
                                       int32_t accumulate, const void* state, const int32_t* a_index, const int32_t* d_index,
                                       void* workspace, size_t workspace_bytes, clica_stream_t stream) {
  CLICA_CHECK_ARG(state && a_index && d_index, "clica_mlp_wgrad_split16: NULL state / index arrays");
  return mlp_wgrad_split_impl(M, n_layers, dZ_planes, X_planes, dZ, lddz, X, ldx, dW, lddw, db, N, K, accumulate, state, a_index, d_index,
                              workspace, workspace_bytes, stream);
}
// clica_mlp_wgrad_split16 behind a backward chain that has left the slabs of the n-wide first / last layer itself
// (clica_mlp_dgrad_split_tail): no tiny-dimension launch -- the drop-in encoder's backward under torch autograd
extern "C" int clica_mlp_wgrad_split16_tail(int64_t M, int32_t n_layers, const void* const* dZ_planes, const void* const* X_planes,
                                            const float* const* dZ, const int64_t* lddz, const float* const* X, const int64_t* ldx,
                                            float* const* dW, const int64_t* lddw, float* const* db, const int32_t* N, const int32_t* K,
                                            int32_t accumulate, const void* state, const int32_t* a_index, const int32_t* d_index,
                                            int32_t tail_slabs, void* workspace, size_t workspace_bytes, clica_stream_t stream) {
  CLICA_CHECK_ARG(state && a_index && d_index, "clica_mlp_wgrad_split16_tail: NULL state / index arrays");
  return mlp_wgrad_split_impl(M, n_layers, dZ_planes, X_planes, dZ, lddz, X, ldx, dW, lddw, db, N, K, accumulate, state, a_index, d_index,
                              workspace, workspace_bytes, stream, nullptr, tail_slabs ? 1 : 0);
}
// Weight gradients AND the optimizer: the trailing reduction launch applies Adam to every element it has just reduced (and carries the
// f16x2 scale update in front when adam->split16_state is set), so a training step needs no optimizer launch.  state == NULL: bf16x3
// plane copies (clica_mlp_wgrad_split), else f16x2 (clica_mlp_wgrad_split16).
extern "C" int clica_mlp_wgrad_split_adam(int64_t M, int32_t n_layers, const void* const* dZ_planes, const void* const* X_planes,
                                          const float* const* dZ, const int64_t* lddz, const float* const* X, const int64_t* ldx,
                                          float* const* dW, const int64_t* lddw, float* const* db, const int32_t* N, const int32_t* K,
                                          const void* state, const int32_t* a_index, const int32_t* d_index, const clica_adam_desc* adam,
                                          int32_t tail_slabs, void* workspace, size_t workspace_bytes, clica_stream_t stream) {
  CLICA_CHECK_ARG(adam, "clica_mlp_wgrad_split_adam: NULL optimizer descriptor");
  CLICA_CHECK_ARG(!state || (a_index && d_index), "clica_mlp_wgrad_split_adam: NULL index arrays");
  return mlp_wgrad_split_impl(M, n_layers, dZ_planes, X_planes, dZ, lddz, X, ldx, dW, lddw, db, N, K, 0, state, a_index, d_index,
                              workspace, workspace_bytes, stream, adam, tail_slabs ? 1 : 0);
}
extern "C" int clica_mlp_chain_tail_supported(int32_t n_layers, const int32_t* N, const int32_t* K, int32_t* supported) {
  CLICA_CHECK_ARG(supported && N && K && n_layers >= 1 && n_layers <= MAXG, "clica_mlp_chain_tail_supported: bad argument");
  *supported = chain_tail_supported(n_layers, N, K) ? 1 : 0;
  return CLICA_OK;
}
extern "C" int clica_mlp_planes16_bytes(int64_t M, int32_t width, int32_t ones_column, size_t* bytes) {
  CLICA_CHECK_ARG(bytes && M > 0 && width >= 1, "clica_mlp_planes16_bytes: bad argument");
  *bytes = (size_t)planes::groups_alloc(M) * planes::units(width, ones_column) * 2 * planes::kPieceBytes;
  return CLICA_OK;
}


// ---- wide layers: forward / data gradient on the split bodies (gemm_split_k) ------------------------------------------------------
extern "C" int clica_mlp_planes_from_f32_t(const float* X, int64_t ldx, int64_t M, int32_t width, void* planes_out, clica_stream_t stream) {
  CLICA_CHECK_ARG(X && planes_out && M > 0 && width >= 1 && ldx >= width, "clica_mlp_planes_from_f32_t: bad argument");
  CLICA_CHECK_ARG((reinterpret_cast<uintptr_t>(planes_out) & 15) == 0, "clica_mlp_planes_from_f32_t: plane buffer must be 16-byte aligned");
  CLICA_CHECK_ARG(M < ((int64_t)1 << 31) - 64, "clica_mlp_planes_from_f32_t: M too large");
  const int units = planes::units((int)M, 0);
  const int64_t groups = planes::groups_alloc(width);
  hipLaunchKernelGGL(planes_from_f32_t_k, dim3((unsigned)ceil_div(groups * units, 2)), dim3(256), 0, as_stream(stream),
                     X, ldx, M, (int)width, units, reinterpret_cast<char*>(planes_out), groups);
  return launch_status("clica_mlp_planes_from_f32_t");
}

static bool g_epi_specialised = true;       // test / A-B hook: clica_set_tuning("gemm16_epilogue", 0) keeps the generic epilogue
namespace clica { namespace wsplit { void set_epi_specialised(int on) { g_epi_specialised = on != 0; } } }
static int launch_gemm_split(Prob& g, int64_t M, int32_t cols, hipStream_t st, const char* who, bool f16 = false) {
  // 128 x 256 tiles when the output is wide enough, else 256 x 128; with whole rounds of 256 x 256 tiles in front where they fit
  // (mixed tiling, see gemm_split_k)
  g.a_wide = cols >= 256 ? 0 : 1;
  int total;
  const int gx256 = (int)ceil_div((int64_t)cols, (int64_t)256);
  // mixed: whole rounds of 256 x 256 tiles (kNumCU / gx row tiles each) while a full round fits, 128 x 256 tiles for the rest
  const int64_t rows_per_big_round = (int64_t)(kNumCU / gx256) * 256;
  const int64_t big_rounds = (gx256 <= kNumCU && cols >= 256) ? M / rows_per_big_round : 0;
  if (big_rounds >= 1 && M - big_rounds * rows_per_big_round > 0) {
    g.a_wide = 3;
    g.gx = gx256;
    g.rows_big = (int)(big_rounds * rows_per_big_round);
    g.n_big = (int)(g.rows_big / 256) * g.gx;
    total = g.n_big + (int)ceil_div(M - g.rows_big, (int64_t)128) * g.gx;
    g.gy = 0;
  } else {
    const int bm = g.a_wide == 0 ? 128 : 256, bn = g.a_wide == 1 ? 128 : 256;
    g.gy = (int)ceil_div(M, (int64_t)bm); g.gx = (int)ceil_div((int64_t)cols, (int64_t)bn);
    total = g.gx * g.gy;
  }
  g.gps = g.groups;
  GroupArgs G{};
  G.n = 1; G.p[0] = g; G.first[0] = 0; G.first[1] = G.total = total;
  if (f16) {
    // the epilogue specialised per direction and output combination where the shape allows it (whole row / lane quads), else the generic one
    const bool quads = (g.M % 4 == 0) && (g.N % 4 == 0) && g_epi_specialised;
    const int code = quads ? epi_code(g.epi == 2, g.outT != nullptr, g.outN != nullptr, g.outF != nullptr) : 1;
#define CLICA_GEMM16_CASE(CODE)                                                                                                          \
    case CODE: {                                                                                                                         \
      static bool once_ = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_k<1, CODE>),                               \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes_of<1>()), true);        \
      (void)once_;                                                                                                                       \
      hipLaunchKernelGGL((gemm_split_k<1, CODE>), dim3((unsigned)G.total), dim3(THREADS), lds_bytes_of<1>(), st, G);                     \
      return launch_status(who);                                                                                                         \
    }
    switch (code) {
      CLICA_GEMM16_CASE(epi_code(false, true, true, false))       // forward inside the wide chain: both plane formats
      CLICA_GEMM16_CASE(epi_code(false, false, true, true))       // forward, last chain layer: N-planes for the next weight gradient + fp32
      CLICA_GEMM16_CASE(epi_code(false, false, false, true))      // forward, fp32 only
      CLICA_GEMM16_CASE(epi_code(true, true, true, false))        // data gradient inside the chain
      CLICA_GEMM16_CASE(epi_code(true, false, true, true))        // data gradient leaving the chain
      CLICA_GEMM16_CASE(epi_code(true, false, false, true))
      default: break;
    }
#undef CLICA_GEMM16_CASE
    static bool once16 = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes_of<1>()), true);
    (void)once16;
    hipLaunchKernelGGL(gemm_split_k<1>, dim3((unsigned)G.total), dim3(THREADS), lds_bytes_of<1>(), st, G);
    return launch_status(who);
  }
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes), true);
  (void)once;
  hipLaunchKernelGGL(gemm_split_k<0>, dim3((unsigned)G.total), dim3(THREADS), kLdsBytes, st, G);
  return launch_status(who);
}
static bool aligned16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int clica_linear_split_fwd(const void* xT_planes, const void* wT_planes, const float* bias, int64_t M, int32_t N, int32_t K,
                                      int32_t leaky, float slope, void* yT_planes, void* yN_planes, int32_t yN_ones,
                                      float* Y, int64_t ldy, clica_stream_t stream) {
  CLICA_CHECK_ARG(xT_planes && wT_planes && M > 0 && N >= 1 && K >= 1, "clica_linear_split_fwd: bad argument");
  CLICA_CHECK_ARG(M < ((int64_t)1 << 31) - 512, "clica_linear_split_fwd: M too large");
  CLICA_CHECK_ARG(yT_planes || yN_planes || Y, "clica_linear_split_fwd: no output");
  CLICA_CHECK_ARG(!Y || ldy >= N, "clica_linear_split_fwd: leading dimension too small");
  CLICA_CHECK_ARG(aligned16p(xT_planes) && aligned16p(wT_planes) && aligned16p(yT_planes) && aligned16p(yN_planes),
                  "clica_linear_split_fwd: plane buffers must be 16-byte aligned");
  Prob g{};
  g.A = reinterpret_cast<const char*>(xT_planes); g.fuA = planes::units((int)M, 0);
  g.B = reinterpret_cast<const char*>(wT_planes); g.fuB = planes::units(N, 0);
  g.M = (int)M; g.N = N; g.groups = (int)planes::groups_used(K);
  g.epi = 1; g.leaky = leaky ? 1 : 0; g.slope = slope; g.bias = bias;
  g.outT = reinterpret_cast<char*>(yT_planes); g.fuT = planes::units((int)M, 0);
  g.outN = reinterpret_cast<char*>(yN_planes); g.fuN = planes::units(N, yN_ones ? 1 : 0);
  g.outF = Y; g.ldf = ldy;
  return launch_gemm_split(g, M, N, as_stream(stream), "clica_linear_split_fwd");
}

extern "C" int clica_linear_split_dgrad(const void* dzT_planes, const void* wN_planes, const void* actT_planes, float slope,
                                        int64_t M, int32_t N, int32_t K, void* dxT_planes, void* dxN_planes,
                                        float* dX, int64_t lddx, clica_stream_t stream) {
  CLICA_CHECK_ARG(dzT_planes && wN_planes && M > 0 && N >= 1 && K >= 1, "clica_linear_split_dgrad: bad argument");
  CLICA_CHECK_ARG(M < ((int64_t)1 << 31) - 512, "clica_linear_split_dgrad: M too large");
  CLICA_CHECK_ARG(dxT_planes || dxN_planes || dX, "clica_linear_split_dgrad: no output");
  CLICA_CHECK_ARG(!dX || lddx >= K, "clica_linear_split_dgrad: leading dimension too small");
  CLICA_CHECK_ARG(aligned16p(dzT_planes) && aligned16p(wN_planes) && aligned16p(actT_planes) && aligned16p(dxT_planes) && aligned16p(dxN_planes),
                  "clica_linear_split_dgrad: plane buffers must be 16-byte aligned");
  Prob g{};
  g.A = reinterpret_cast<const char*>(dzT_planes); g.fuA = planes::units((int)M, 0);
  g.B = reinterpret_cast<const char*>(wN_planes); g.fuB = planes::units(K, 0);
  g.M = (int)M; g.N = K; g.groups = (int)planes::groups_used(N);
  g.epi = 2; g.slope = slope;
  g.signT = reinterpret_cast<const char*>(actT_planes); g.fuS = planes::units((int)M, 0);
  g.outT = reinterpret_cast<char*>(dxT_planes); g.fuT = planes::units((int)M, 0);
  g.outN = reinterpret_cast<char*>(dxN_planes); g.fuN = planes::units(K, 0);
  g.outF = dX; g.ldf = lddx;
  return launch_gemm_split(g, M, K, as_stream(stream), "clica_linear_split_dgrad");
}

// ---- f16x2 variants of the per-layer entry points (config 3's wide chain; include/clica.h "f16x2 arithmetic") -------------------------
// Tensors are named by (family, index) in the caller's Split16 state: family 0 = activations (index l = INPUT of layer l), 1 = gradients
// (index l = dZ_l), 2 = weights (index l).  Producers run on the scale in force and record their maximum for the next update.
static int s16_ok(const void* state, int family, int index, const char* who) {
  CLICA_CHECK_ARG(state && family >= 0 && family <= 2 && index >= 0 && index < s16::Split16State::NT, "%s: bad state / tensor (%d, %d)", who, family, index);
  return CLICA_OK;
}
extern "C" int clica_mlp_planes16_from_f32(const float* X, int64_t ldx, int64_t M, int32_t width, int32_t ones_column, void* planes_out,
                                           void* state, int32_t family, int32_t index, clica_stream_t stream) {
  CLICA_CHECK_ARG(X && planes_out && M > 0 && width >= 1 && ldx >= width, "clica_mlp_planes16_from_f32: bad argument");
  CLICA_CHECK_ARG((reinterpret_cast<uintptr_t>(planes_out) & 15) == 0, "clica_mlp_planes16_from_f32: plane buffer must be 16-byte aligned");
  int rc = s16_ok(state, family, index, "clica_mlp_planes16_from_f32"); if (rc) return rc;
  const int units = planes::units(width, ones_column);
  const int64_t groups = planes::groups_alloc(M);
  hipLaunchKernelGGL(planes16_from_f32_k, dim3((unsigned)ceil_div(groups * units, 2)), dim3(256), 0, as_stream(stream),
                     X, ldx, M, (int)width, ones_column ? 1 : 0, units, reinterpret_cast<char*>(planes_out), groups,
                     s16::s16_tensor(reinterpret_cast<s16::Split16State*>(state), family, index));
  return launch_status("clica_mlp_planes16_from_f32");
}
extern "C" int clica_mlp_planes16_from_f32_t(const float* X, int64_t ldx, int64_t M, int32_t width, void* planes_out,
                                             void* state, int32_t family, int32_t index, clica_stream_t stream) {
  CLICA_CHECK_ARG(X && planes_out && M > 0 && width >= 1 && ldx >= width, "clica_mlp_planes16_from_f32_t: bad argument");
  CLICA_CHECK_ARG((reinterpret_cast<uintptr_t>(planes_out) & 15) == 0, "clica_mlp_planes16_from_f32_t: plane buffer must be 16-byte aligned");
  CLICA_CHECK_ARG(M < ((int64_t)1 << 31) - 64, "clica_mlp_planes16_from_f32_t: M too large");
  int rc = s16_ok(state, family, index, "clica_mlp_planes16_from_f32_t"); if (rc) return rc;
  const int units = planes::units((int)M, 0);
  const int64_t groups = planes::groups_alloc(width);
  hipLaunchKernelGGL(planes16_from_f32_t_k, dim3((unsigned)ceil_div(groups * units, 2)), dim3(256), 0, as_stream(stream),
                     X, ldx, M, (int)width, units, reinterpret_cast<char*>(planes_out), groups,
                     s16::s16_tensor(reinterpret_cast<s16::Split16State*>(state), family, index));
  return launch_status("clica_mlp_planes16_from_f32_t");
}
extern "C" int clica_linear_split_fwd16(const void* xT_planes, const void* wT_planes, const float* bias, int64_t M, int32_t N, int32_t K,
                                        int32_t leaky, float slope, void* yT_planes, void* yN_planes, int32_t yN_ones,
                                        float* Y, int64_t ldy, void* state, int32_t layer, clica_stream_t stream) {
  CLICA_CHECK_ARG(xT_planes && wT_planes && M > 0 && N >= 1 && K >= 1, "clica_linear_split_fwd16: bad argument");
  CLICA_CHECK_ARG(M < ((int64_t)1 << 31) - 512, "clica_linear_split_fwd16: M too large");
  CLICA_CHECK_ARG(yT_planes || yN_planes || Y, "clica_linear_split_fwd16: no output");
  CLICA_CHECK_ARG(!Y || ldy >= N, "clica_linear_split_fwd16: leading dimension too small");
  CLICA_CHECK_ARG(aligned16p(xT_planes) && aligned16p(wT_planes) && aligned16p(yT_planes) && aligned16p(yN_planes),
                  "clica_linear_split_fwd16: plane buffers must be 16-byte aligned");
  int rc = s16_ok(state, 0, layer + 1, "clica_linear_split_fwd16"); if (rc) return rc;
  s16::Split16State* st16 = reinterpret_cast<s16::Split16State*>(state);
  Prob g{};
  g.A = reinterpret_cast<const char*>(xT_planes); g.fuA = planes::units((int)M, 0);
  g.B = reinterpret_cast<const char*>(wT_planes); g.fuB = planes::units(N, 0);
  g.M = (int)M; g.N = N; g.groups = (int)planes::groups_used(K);
  g.epi = 1; g.leaky = leaky ? 1 : 0; g.slope = slope; g.bias = bias;
  g.outT = reinterpret_cast<char*>(yT_planes); g.fuT = planes::units((int)M, 0);
  g.outN = reinterpret_cast<char*>(yN_planes); g.fuN = planes::units(N, yN_ones ? 1 : 0);
  g.outF = Y; g.ldf = ldy;
  g.scaleA = &st16->sA[layer]; g.scaleB = &st16->sW[layer];            // X = input of `layer`, W = its weights
  g.outS = s16::s16_tensor(st16, 0, layer + 1);                         // Y = input of layer + 1
  return launch_gemm_split(g, M, N, as_stream(stream), "clica_linear_split_fwd16", true);
}
extern "C" int clica_linear_split_dgrad16(const void* dzT_planes, const void* wN_planes, const void* actT_planes, float slope,
                                          int64_t M, int32_t N, int32_t K, void* dxT_planes, void* dxN_planes,
                                          float* dX, int64_t lddx, void* state, int32_t layer, clica_stream_t stream) {
  CLICA_CHECK_ARG(dzT_planes && wN_planes && M > 0 && N >= 1 && K >= 1, "clica_linear_split_dgrad16: bad argument");
  CLICA_CHECK_ARG(M < ((int64_t)1 << 31) - 512, "clica_linear_split_dgrad16: M too large");
  CLICA_CHECK_ARG(dxT_planes || dxN_planes || dX, "clica_linear_split_dgrad16: no output");
  CLICA_CHECK_ARG(!dX || lddx >= K, "clica_linear_split_dgrad16: leading dimension too small");
  CLICA_CHECK_ARG(aligned16p(dzT_planes) && aligned16p(wN_planes) && aligned16p(actT_planes) && aligned16p(dxT_planes) && aligned16p(dxN_planes),
                  "clica_linear_split_dgrad16: plane buffers must be 16-byte aligned");
  CLICA_CHECK_ARG(layer >= 1, "clica_linear_split_dgrad16: layer must be >= 1 (its output is dZ of layer - 1)");
  int rc = s16_ok(state, 1, layer, "clica_linear_split_dgrad16"); if (rc) return rc;
  s16::Split16State* st16 = reinterpret_cast<s16::Split16State*>(state);
  Prob g{};
  g.A = reinterpret_cast<const char*>(dzT_planes); g.fuA = planes::units((int)M, 0);
  g.B = reinterpret_cast<const char*>(wN_planes); g.fuB = planes::units(K, 0);
  g.M = (int)M; g.N = K; g.groups = (int)planes::groups_used(N);
  g.epi = 2; g.slope = slope;
  g.signT = reinterpret_cast<const char*>(actT_planes); g.fuS = planes::units((int)M, 0);
  g.outT = reinterpret_cast<char*>(dxT_planes); g.fuT = planes::units((int)M, 0);
  g.outN = reinterpret_cast<char*>(dxN_planes); g.fuN = planes::units(K, 0);
  g.outF = dX; g.ldf = lddx;
  g.scaleA = &st16->sD[layer]; g.scaleB = &st16->sW[layer];            // dZ_layer, W_layer
  g.outS = s16::s16_tensor(st16, 1, layer - 1);                         // dZ_{layer - 1}
  return launch_gemm_split(g, M, K, as_stream(stream), "clica_linear_split_dgrad16", true);
}
