// Fused Linear (+bias, +LeakyReLU) forward / dgrad / wgrad for the MLP encoder
// (get_mlp, /root/reference/encoders.py:36-48) on gfx950, exact fp32 on the matrix cores.
//
// One LDS-tiled GEMM template, C[M,N] = A_op[M,Kc] * B_op[Kc,N], instantiated for the three
// operand layouts the layer needs (all tensors row-major, contraction index Kc):
//   fwd   Y  = act(X W^T + b)        A = X  [M][Kc]  (Kc contiguous)   B = W  [N][Kc]  (Kc contiguous)
//   dgrad dX = (dY W) * act'(Xact)   A = dY [M][Kc]  (Kc contiguous)   B = W  [Kc][N]  (Kc strided)
//   wgrad dW = dY^T X, db = 1^T dY   A = dY [Kc][M]  (Kc strided)      B = X  [Kc][N]  (Kc strided)
//
// Math: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate; bit-identical to an fmaf chain, which is
// what keeps the 1e-5 parity bar -- bf16/xf32 inputs would not).  Each wave owns a (BM/WM)x(BN/WN)
// sub-tile as 32x32 accumulator blocks.  Because the contraction order is free, a lane reads FOUR
// consecutive k (one ds_read_b128) and the four MFMAs that follow pair k = 8s+4h+t of both
// operands (h = lane>>5), so a K-contiguous operand costs one LDS read per four MFMAs.
//
// Pipeline: global -> registers (next tile) overlaps the MFMAs of the current tile; registers ->
// LDS double buffer; one barrier per K tile.  wgrad splits the long contraction (Kc = rows of the
// batch) over blockIdx.z into fp32 slabs that a second tiny kernel sums deterministically.
#include "common.h"
#include <string.h>
#include "wgrad_shared.h"
#include "split16.h"
#include "adam_math.h"
#include <stdlib.h>
#include <algorithm>

namespace clica {
// skinny.hip: VALU kernels for layers with a tiny contraction or output width
bool skinny_fwd(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                int64_t M, int64_t N, int64_t K, int leaky, float slope, hipStream_t st);
bool skinny_dgrad(const float* dY, int64_t lddy, const float* W, int64_t ldw, const float* Xact, int64_t ldxa, float slope,
                  float* dX, int64_t lddx, int64_t M, int64_t N, int64_t K, hipStream_t st);

namespace conv16 {   // conv16.hip
void launch_slab_sum(const float* slab1, int n1, float* out1, const float* slab2, int n2, float* out2, int splits, int accumulate, hipStream_t st);
}
namespace gemm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int RED_THREADS = 256;

enum Epi { EPI_BIAS_ACT = 0, EPI_DACT = 1, EPI_SLAB = 2 };

// out-of-range tile pieces load from here: the zero comes straight from memory, so nothing has to
// touch the loaded registers before the LDS store and the loads stay in flight across the MFMAs
__device__ __attribute__((aligned(16))) float g_zero_page[4] = {0.f, 0.f, 0.f, 0.f};
// {1, 0, 0, 0}: the first padding column of the wgrad B operand reads this, so the MFMAs produce db = dZ^T 1 there
__device__ __attribute__((aligned(16))) float g_one_page[4] = {1.f, 0.f, 0.f, 0.f};

// Implicit-GEMM view of the k = 4, stride-2, pad-1 convolutions of BetaVAE_H (conv section at the end of this file): operand
// segments for the loaders and the row <-> pixel geometry the two scattering epilogues need.
struct ConvX {
  int64_t a_seg, a_jump;     // A operand (see Tile::load)
  int64_t b_seg, b_jump;     // B operand
  int mode;                  // 0: rows are stored as they are; 1: forward scatter into the next layer's space-to-depth
                             // tensor; 2: data-gradient scatter back into the previous layer's output-gradient pixels
  int hs, ws, ho, wo;        // GEMM row r = (image * hs + y) * ws + x; mode 1 stores rows with y < ho and x < wo only
  int dhs, dws, dho, dwo;    // destination pixel grid (rows of dhs x dws per image, valid dho x dwo)
  int c;                     // channels per pixel of the scattered tensor: mode 1 = N, mode 2 = N / 4
  float inv_pix, inv_ws;     // 1 / (hs * ws), 1 / ws for the epilogues' row -> pixel arithmetic (rows < 2^24)
  int fast;                  // contraction is a whole number of k-tiles and unsplit: CONTIG operands take Tile::load_fast
  int compact;               // forward only (needs fast): GEMM rows run over the OUTPUT pixels alone (hs = ho, ws = wo above) and a row's
  int ghs, gws;              // A pointer is taken at pixel (y, x) of the ghs x gws grid S is stored on -- no work on non-output rows
  unsigned* gate_out;        // mode 1: one bit per output element, (y > 0), word [row][col / 32] of this stage's row grid (or nullptr)
  const unsigned* gate_in;   // mode 2: the previous stage's gate bits on the destination grid (replaces the 4-byte gate read of xact)
};

// ---- tile loaders: global -> registers ------------------------------------------------------
// CONTIG: operand stored [rows][Kc]; tile = ROWS x BK, float4 along k.
// !CONTIG: operand stored [Kc][rows]; tile = BK x ROWS, float4 along rows.
template <int ROWS, bool CONTIG, int THREADS>
struct Tile {
  static constexpr int UNITS = ROWS * BK / 4;          // float4 units per tile
  static constexpr int PER_THREAD = UNITS / THREADS;
  static_assert(UNITS % THREADS == 0, "tile must divide over the workgroup");
  static constexpr int LDS_LD = CONTIG ? (BK + 4) : ROWS;   // +4 floats: conflict-free ds_read_b128
  static constexpr int LDS_FLOATS = CONTIG ? ROWS * (BK + 4) : BK * ROWS;

  // Branch-free: out-of-range pieces read the zero page, so all loads of a tile issue back-to-back
  // and nothing depends on them until the LDS store after the MFMAs.  VEC needs 16-byte aligned rows and a
  // contraction/row extent that is a multiple of 4 (every float4 is then fully in or fully out).
  template <bool VEC>
  static __device__ __forceinline__ void load(float4 (&r)[PER_THREAD], const float* __restrict__ src, int64_t ld,
                                              int64_t row0, int64_t nrows, int64_t k0, int64_t kend,
                                              const int64_t seg = 0, const int64_t jump = 0) {
    // (seg, jump): two-segment operand of the implicit-GEMM convolutions (conv section below) -- the index along the float4
    // direction (k when CONTIG, the row index otherwise) continues `jump` elements further on once it reaches `seg`
    // (a multiple of 4); jump = 0 is the plain operand and folds away.
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int u = threadIdx.x + i * THREADS;
      int64_t off, lim_a, lim_b;   // element offset of .x ; remaining valid elements along the float4
      bool ok;
      if (CONTIG) {
        const int row = u / (BK / 4), kq = u % (BK / 4);
        const int64_t gr = row0 + row, gk = k0 + 4 * kq;
        ok = gr < nrows; off = gr * ld + gk + (gk >= seg ? jump : 0); lim_a = kend - gk;
      } else {
        const int k = u / (ROWS / 4), rq = u % (ROWS / 4);
        const int64_t gk = k0 + k, gr = row0 + 4 * rq;
        ok = gk < kend; off = gk * ld + gr + (gr >= seg ? jump : 0); lim_a = nrows - gr;
      }
      (void)lim_b;
      float4 v;
      if (VEC) {
        const bool in = ok && lim_a > 0;
        v = *reinterpret_cast<const float4*>(in ? src + off : g_zero_page);
      } else {
        const bool i0 = ok && lim_a > 0, i1 = ok && lim_a > 1, i2 = ok && lim_a > 2, i3 = ok && lim_a > 3;
        v = make_float4(*(i0 ? src + off : g_zero_page), *(i1 ? src + off + 1 : g_zero_page),
                        *(i2 ? src + off + 2 : g_zero_page), *(i3 ? src + off + 3 : g_zero_page));
      }
      r[i] = v;
    }
  }

  // Conv fast path (CONTIG operands, contraction a multiple of BK, no splits): per-thread row pointers are set up ONCE, a tile load is
  // then one compare / select / add per thread and one 16-byte load per unit.  (The generic loader recomputes 64-bit row offsets, bounds and
  // zero-page selects per unit and tile: PMC counted 9 vector instructions per MFMA on the 32-wide conv stage, 145 per 16-MFMA k-tile.)
  // Rows past the end are clamped to the last row instead of zero-filled: their results are never stored.
  struct Fast { const float* p[PER_THREAD]; int thr; };
  static __device__ __forceinline__ void prep(Fast& f, const float* src, int64_t ld, int64_t row0, int64_t nrows, int64_t seg) {
    const int kq = threadIdx.x % (BK / 4);
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int row = (threadIdx.x + i * THREADS) / (BK / 4);
      const int64_t gr = min(row0 + row, nrows - 1);
      f.p[i] = src + gr * ld + 4 * kq;
    }
    f.thr = (int)min(seg, (int64_t)1 << 30) - 4 * kq;        // tile start k0 >= thr: this thread's float4 lies in the second run
  }
  static __device__ __forceinline__ void load_fast(float4 (&r)[PER_THREAD], const Fast& f, int k0, int jump) {
    const int o = k0 + (k0 >= f.thr ? jump : 0);
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) r[i] = *reinterpret_cast<const float4*>(f.p[i] + o);
  }

  static __device__ __forceinline__ void store(const float4 (&r)[PER_THREAD], float* lds) {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int u = threadIdx.x + i * THREADS;
      if (CONTIG) {
        const int row = u / (BK / 4), kq = u % (BK / 4);
        *reinterpret_cast<float4*>(&lds[row * LDS_LD + 4 * kq]) = r[i];
      } else {
        const int k = u / (ROWS / 4), rq = u % (ROWS / 4);
        *reinterpret_cast<float4*>(&lds[k * LDS_LD + 4 * rq]) = r[i];
      }
    }
  }

  // the four k-values (8s + 4h + t, t = 0..3) of row `row` for this lane's half h
  static __device__ __forceinline__ void frag(float (&f)[4], const float* lds, int row, int s, int h) {
    if (CONTIG) {
      const float4 v = *reinterpret_cast<const float4*>(&lds[row * LDS_LD + 8 * s + 4 * h]);
      f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) f[t] = lds[(8 * s + 4 * h + t) * LDS_LD + row];
    }
  }
};

// GEMM row -> (image, y, x) on the conv's hs x ws row grid without integer division: rows < 2^24 are exact in fp32, the rounded
// reciprocal can miss the quotient by one either way, one correction step each.
__device__ __forceinline__ void conv_row_to_pixel(const ConvX& cx, int64_t row, int& img, int& y, int& x) {
  const int pix = cx.hs * cx.ws;
  img = (int)((float)(int)row * cx.inv_pix);
  int rem = (int)row - img * pix;
  if (rem < 0) { --img; rem += pix; } else if (rem >= pix) { ++img; rem -= pix; }
  y = (int)((float)rem * cx.inv_ws); x = rem - y * cx.ws;
  if (x < 0) { --y; x += cx.ws; } else if (x >= cx.ws) { ++y; x -= cx.ws; }
}

// The workgroup's whole job for output tile (bx, by) and contraction split bz of problem g.
template <int BM, int BN, int WM, int WN, int STAGES, bool A_CONTIG, bool B_CONTIG, int EPI, bool VEC, bool CONV = false>
__device__ __forceinline__ void gemm_body(const Args& g, const int bx, const int by, const int bz, const ConvX cx = ConvX{}) {
  const int64_t aseg = CONV ? cx.a_seg : 0, ajump = CONV ? cx.a_jump : 0, bseg = CONV ? cx.b_seg : 0, bjump = CONV ? cx.b_jump : 0;
  constexpr int THREADS = 64 * WM * WN;
  using TA = Tile<BM, A_CONTIG, THREADS>;
  using TB = Tile<BN, B_CONTIG, THREADS>;
  constexpr int TM = BM / WM, TN = BN / WN;        // wave tile
  constexpr int NBM = TM / 32, NBN = TN / 32;      // 32x32 accumulator blocks per wave
  static_assert(TM % 32 == 0 && TN % 32 == 0, "wave tile must be a multiple of 32");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int STAGE = TA::LDS_FLOATS + TB::LDS_FLOATS;   // one pipeline stage: [A tile][B tile]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, l31 = lane & 31;
  const int64_t m0 = (int64_t)by * BM, n0 = (int64_t)bx * BN;
  int64_t kbeg = 0, kend = g.Kc;
  if (EPI == EPI_SLAB) {
    kbeg = (int64_t)bz * g.k_per_split;
    kend = min(g.Kc, kbeg + g.k_per_split);
  }

  f32x16 acc[NBM][NBN];
#pragma unroll
  for (int i = 0; i < NBM; ++i)
#pragma unroll
    for (int j = 0; j < NBN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float colsum = 0.f;  // wgrad: db partial for A_op row (threadIdx.x < BM), only blockIdx.x == 0

  // Pipeline: STAGES LDS stages + one register stage for the global loads + double-buffered MFMA
  // fragments.
  //   * registers hold tile t+1 at the top of iteration t; they are stored to LDS stage (t+1)%STAGES
  //     in the shadow of the first k-step's MFMAs, then the global loads of tile t+2 are issued --
  //     global latency has a whole iteration to hide in.
  //   * fragments of k-step s+1 are read from LDS while the MFMAs of k-step s run.
  //   * STAGES == 3: the only barrier of the iteration sits right after the stores; the stage being
  //     overwritten was last read two iterations ago, so the k-loop runs THROUGH tile boundaries
  //     (the first fragments of tile t+1 are prefetched during the last k-step of tile t).
  //   * STAGES == 2 (small-LDS shapes, two workgroups per CU): classic barrier at the end of the tile.
  static_assert(STAGES == 2 || STAGES == 3, "2 or 3 LDS stages");
  constexpr int KSTEPS = BK / 8;
  float4 ra[TA::PER_THREAD], rb[TB::PER_THREAD];
  typename TA::Fast fa; typename TB::Fast fb;
  const bool fast = CONV && cx.fast;                        // uniform: conv launch whose contraction is whole k-tiles, no splits
  if (CONV && A_CONTIG && fast) {
    TA::prep(fa, g.A, g.lda, m0, g.M, cx.a_jump ? cx.a_seg : (int64_t)1 << 30);
    if (cx.compact) {          // row -> output pixel -> its place on the storage grid
#pragma unroll
      for (int i = 0; i < TA::PER_THREAD; ++i) {
        const int row = (threadIdx.x + i * THREADS) / (BK / 4);
        int img, y, x;
        conv_row_to_pixel(cx, min(m0 + row, g.M - 1), img, y, x);
        fa.p[i] = g.A + (((int64_t)img * cx.ghs + y) * cx.gws + x) * g.lda + 4 * (threadIdx.x % (BK / 4));
      }
    }
  }
  if (CONV && B_CONTIG && fast) TB::prep(fb, g.B, g.ldb, n0, g.N, (int64_t)1 << 30);
  auto load_a = [&](int64_t k0) {
    if (CONV && A_CONTIG && fast) TA::load_fast(ra, fa, (int)k0, (int)cx.a_jump);
    else TA::template load<VEC>(ra, g.A, g.lda, m0, g.M, k0, kend, aseg, ajump);
  };
  auto load_b = [&](int64_t k0) {
    if (CONV && B_CONTIG && fast) TB::load_fast(rb, fb, (int)k0, 0);
    else TB::template load<VEC>(rb, g.B, g.ldb, n0, g.N, k0, kend, bseg, bjump);
  };
  const int ntiles = (int)((kend - kbeg + BK - 1) / BK);
  if (ntiles > 0) {
    load_a(kbeg);
    load_b(kbeg);
    TA::store(ra, smem);
    TB::store(rb, smem + TA::LDS_FLOATS);
    if (ntiles > 1) {
      load_a(kbeg + BK);
      load_b(kbeg + BK);
    }
  }
  __syncthreads();

  float af[2][NBM][4], bf[2][NBN][4];
  auto load_frags = [&](int buf, const float* a_s, int s) {
    const float* b_s = a_s + TA::LDS_FLOATS;
#pragma unroll
    for (int i = 0; i < NBM; ++i) TA::frag(af[buf][i], a_s, wm * TM + i * 32 + l31, s, h);
#pragma unroll
    for (int j = 0; j < NBN; ++j) TB::frag(bf[buf][j], b_s, wn * TN + j * 32 + l31, s, h);
  };
  if (ntiles > 0) load_frags(0, smem, 0);

  int cur = 0;   // LDS stage of tile t
  for (int t = 0; t < ntiles; ++t) {
    const int nxt = (cur + 1 == STAGES) ? 0 : cur + 1;
    const float* a_s = smem + cur * STAGE;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      // fragments for the NEXT k-step are in flight while this k-step's MFMAs run
      if (s + 1 < KSTEPS) load_frags((s + 1) & 1, a_s, s + 1);
      else if (STAGES == 3 && t + 1 < ntiles) load_frags(0, smem + nxt * STAGE, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int i = 0; i < NBM; ++i)
#pragma unroll
          for (int j = 0; j < NBN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][i][tt], bf[s & 1][j][tt], acc[i][j], 0, 0, 0);
      if (s == 0) {
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < ntiles) {
          TA::store(ra, smem + nxt * STAGE);
          TB::store(rb, smem + nxt * STAGE + TA::LDS_FLOATS);
        }
        if (t + 2 < ntiles) {
          const int64_t k0 = kbeg + (int64_t)(t + 2) * BK;
          load_a(k0);
          load_b(k0);
        }
        if (STAGES == 3) __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (EPI == EPI_SLAB && g.dbias_slab && bx == 0 && !A_CONTIG && threadIdx.x < BM) {
#pragma unroll 8
      for (int k = 0; k < BK; ++k) colsum += a_s[k * TA::LDS_LD + threadIdx.x];
    }
    if (STAGES == 2) {
      __syncthreads();
      if (t + 1 < ntiles) load_frags(0, smem + nxt * STAGE, 0);
    }
    cur = nxt;
  }

  // ---- epilogue: C/D layout of 32x32 blocks: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* Cbase = g.C;
  if (EPI == EPI_SLAB) Cbase += (int64_t)bz * g.M * g.ldc;
#pragma unroll
  for (int i = 0; i < NBM; ++i) {
#pragma unroll
    for (int j = 0; j < NBN; ++j) {
      const int64_t col = n0 + wn * TN + j * 32 + l31;
      if (col >= g.N) continue;
      float bv = 0.f;
      if (EPI == EPI_BIAS_ACT && g.bias) bv = g.bias[col];
      // consumed HERE, in uniform control flow: first used inside the row loops' bounds branches, the compiler repeated the wait for this
      // load -- a vmcnt(0), which also waits for every store issued so far -- in front of EVERY row's store (found in the disassembly of
      // conv16's streaming kernel, round 5: all 64 stores of this epilogue were preceded by one)
      asm volatile("" ::"v"(bv));
      if (CONV && cx.mode != 0) {
        // scattering epilogues of the convolutions: the GEMM row is a pixel of a (hs x ws) grid per image.  Two passes: destination
        // offsets and (mode 2) the gate values of all 16 rows first -- the gate loads are then in flight together instead of one
        // round trip in front of every store -- then the stores.
        const int q = (int)col / cx.c, ch = (int)col - q * cx.c;       // mode 2: column = (py, px, channel)
        const float* __restrict__ gate_src = g.xact;
        float* __restrict__ dst = g.C;
        int64_t off[16];
        float gate[16];
        int grid_row[16];                                                 // mode 1: the row's index on the stage's storage grid (gate words)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          off[r] = -1; gate[r] = 1.f;
          if (row >= g.M) continue;
          int img, y, x;
          conv_row_to_pixel(cx, row, img, y, x);
          if (cx.mode == 3) {
            // plain rows, but at the row's place on the storage grid: only output rows are computed and written (the others stay what
            // the caller put there once: zeros)
            if (y >= cx.ho || x >= cx.wo) continue;
            off[r] = (int64_t)((img * cx.ghs + y) * cx.gws + x) * g.ldc + col;
          } else if (cx.mode == 1) {
            // y, x = output pixel; it is element ((y+1)&1, (x+1)&1, col) of pixel ((y+1)/2, (x+1)/2) of the next layer's
            // padded space-to-depth input (4 * c channels)
            if (y >= cx.ho || x >= cx.wo) continue;
            const int Y = (y + 1) >> 1, X = (x + 1) >> 1, qq = ((y + 1) & 1) * 2 + ((x + 1) & 1);
            off[r] = (((int64_t)img * cx.dhs + Y) * cx.dws + X) * (4 * cx.c) + qq * cx.c + col;
            grid_row[r] = (img * cx.ghs + y) * cx.gws + x;
          } else {
            // y, x = pixel of this layer's space-to-depth input, column q = (py, px): the gradient of the previous layer's
            // output pixel (2y + py - 1, 2x + px - 1), gated by that output's ReLU (xact = the space-to-depth tensor itself)
            const int yy = 2 * y + (q >> 1) - 1, xx = 2 * x + (q & 1) - 1;
            if (yy < 0 || xx < 0 || yy >= cx.dho || xx >= cx.dwo) continue;
            off[r] = (((int64_t)img * cx.dhs + yy) * cx.dws + xx) * cx.c + ch;
            if (cx.gate_in) {
              // (requesting these words ahead of the k-loop instead was measured: 32 more live registers cost every data-gradient launch
              //  20-40 % -- 427 -> 596 us on the widest stage -- far more than the dependent round trip here)
              const unsigned word = cx.gate_in[(((int64_t)img * cx.dhs + yy) * cx.dws + xx) * (cx.c >> 5) + (ch >> 5)];
              gate[r] = (word >> (ch & 31)) & 1u ? 1.f : 0.f;
            } else if (gate_src) {
              gate[r] = gate_src[row * g.ldxa + col];
            }
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (off[r] < 0) continue;
          float v = acc[i][j][r];
          if (cx.mode == 1 || cx.mode == 3) {
            v += bv;
            if (g.leaky) v = v > 0.f ? v : v * g.slope;
          } else if (gate_src || cx.gate_in) {
            v *= (gate[r] > 0.f ? 1.f : g.slope);
          }
          dst[off[r]] = v;
          if (cx.mode == 1 && cx.gate_out) {
            // the 32 lanes of a half-wave hold 32 consecutive channels of one row: their (v > 0) bits are one word of the gate tensor
            const unsigned long long bits = __ballot(v > 0.f);
            if (l31 == 0) cx.gate_out[(int64_t)grid_row[r] * (g.N >> 5) + (col >> 5)] = (unsigned)(bits >> (32 * h));
          }
        }
        continue;
      }
      // EPI_DACT: the sixteen activations of the block are requested together and consumed once (one load and its round trip -- plus the
      // drain of the previous row's store -- in front of every store before)
      float xa[16];
      if (EPI == EPI_DACT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          xa[r] = g.xact ? g.xact[min(row, g.M - 1) * g.ldxa + col] : 1.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(xa[r]));
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= g.M) continue;
        float v = acc[i][j][r];
        if (EPI == EPI_BIAS_ACT) {
          v += bv;
          if (g.leaky) v = v > 0.f ? v : v * g.slope;
        } else if (EPI == EPI_DACT) {
          v *= (xa[r] > 0.f ? 1.f : g.slope);
        }
        Cbase[row * g.ldc + col] = v;
      }
    }
  }
  if (EPI == EPI_SLAB && g.dbias_slab && bx == 0 && !A_CONTIG && threadIdx.x < BM) {
    const int64_t row = m0 + threadIdx.x;
    if (row < g.M) g.dbias_slab[(int64_t)bz * g.M + row] = colsum;
  }
}

// XCD-aware order: the dispatcher places linear block b on XCD b % 8 (speed only, never correctness).
// Re-number so that each XCD walks a CONTIGUOUS range of the nwg work items: workgroups that share
// an operand panel then hit the same XCD-private L2.
__device__ __forceinline__ int xcd_contiguous(int b, int nwg) {
  const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

template <int BM, int BN, int WM, int WN, int STAGES, bool A_CONTIG, bool B_CONTIG, int EPI, bool VEC>
__global__ __launch_bounds__(64 * WM * WN) void gemm_k(Args g) {
  const int gx = gridDim.x;
  const int id = xcd_contiguous(blockIdx.y * gx + blockIdx.x, gx * gridDim.y);   // N fastest
  const int by = id / gx;
  gemm_body<BM, BN, WM, WN, STAGES, A_CONTIG, B_CONTIG, EPI, VEC>(g, id - by * gx, by, blockIdx.z);
}

// ---- weight-gradient body with direct global -> LDS tile loads ------------------------------------------
// Both wgrad operands are [batch rows][features] with the contraction along the batch rows, so a BK x 128
// tile is BK global row pieces of 512 B and its LDS image [BK][128] is exactly lane-contiguous: one
// global_load_lds_dwordx4 per wave moves two k-rows (1 KB) HBM/L2 -> LDS without touching a VGPR and
// without a ds_write.  Four LDS stages: the loads of tile t+3 are issued in iteration t and first
// waited for in iteration t+2, i.e. two full iterations (~4 us) of MFMA cover their latency; one barrier
// per iteration, placed after the first k-step so the fragment prefetch runs through tile boundaries.
// The loads are issued from inline asm: the compiler then does not see an LDS write through VMEM (for which it
// would insert s_waitcnt vmcnt(0) in front of every later ds_read) and the waits below are the exact ones.
typedef __attribute__((address_space(3))) float* lds_f32_ptr;
__device__ __forceinline__ void dma_1k(const float* lane_src, const float* lds_wave_dst) {
  const unsigned off = (unsigned)(size_t)(lds_f32_ptr)lds_wave_dst;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(lane_src), "s"(off) : "memory", "m0");
#pragma clang diagnostic pop
}
template <int N> __device__ __forceinline__ void wait_vm() {   // s_waitcnt vmcnt(N) only (gfx9 encoding)
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

// Debug build only (-DCLICA_WGRAD_TRACE, tools/wgrad_trace.py): s_memtime stamps per (workgroup, wave, phase).
#ifdef CLICA_WGRAD_TRACE
__device__ unsigned long long* g_wtrace = nullptr;
#define WG_STAMP(ph) do { if (g_wtrace && (threadIdx.x & 63) == 0) g_wtrace[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (ph)] = clock64(); } while (0)
#else
#define WG_STAMP(ph) do { } while (0)
#endif
constexpr int DMA_STAGES = 4;
template <int WM, int WN>
__device__ __forceinline__ void wgrad_dma_body(const Args& g, const int bx, const int by, const int bz) {
  WG_STAMP(0);
  constexpr int BM = 128, BN = 128, THREADS = 64 * WM * WN, WAVES = WM * WN;
  static_assert(WAVES == 8, "eight waves: each issues 2 + 2 one-KB loads per 32-deep tile");
  using TA = Tile<BM, false, THREADS>;
  using TB = Tile<BN, false, THREADS>;
  constexpr int TM = BM / WM, TN = BN / WN, NBM = TM / 32, NBN = TN / 32;
  constexpr int STAGE = TA::LDS_FLOATS + TB::LDS_FLOATS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave % WN, h = lane >> 5, l31 = lane & 31;
  const int64_t m0 = (int64_t)by * BM, n0 = (int64_t)bx * BN;
  const int64_t kbeg = (int64_t)bz * g.k_per_split, kend = min(g.Kc, kbeg + g.k_per_split);
  const int ntiles = (int)((kend - kbeg + BK - 1) / BK);

  f32x16 acc[NBM][NBN];
#pragma unroll
  for (int i = 0; i < NBM; ++i)
#pragma unroll
    for (int j = 0; j < NBN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float colsum = 0.f;

  // this lane's piece of the tile: k-rows 4*wave + 2*j + h (j = 0, 1), columns 4*l31 .. +3.  The four source
  // pointers are kept running (one 64-bit add per load and tile; lanes whose columns are out of range sit on
  // the zero page with stride 0): on this chip fp32 MFMA and the vector ALU share issue, address arithmetic
  // inside the k-loop is paid for in matrix throughput.
  const bool a_ok = m0 + 4 * l31 < g.M, b_ok = n0 + 4 * l31 < g.N;      // M, N multiples of 4 on this path
  // db for free: when the last column tile has padding, its first padding column (col == N) is fed ones, and
  // the matrix cores return the column sums of dZ there -- no per-iteration LDS column sum on the vector ALU
  const bool ones_col = g.dbias_slab && (g.N % BN != 0);
  const float* b_pad = (ones_col && n0 + 4 * l31 == g.N) ? g_one_page : g_zero_page;
  const float* pa[2]; const float* pb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t k = kbeg + 4 * wave + 2 * j + h;
    pa[j] = a_ok ? g.A + k * g.lda + m0 + 4 * l31 : g_zero_page;
    pb[j] = b_ok ? g.B + k * g.ldb + n0 + 4 * l31 : b_pad;
  }
  const int64_t sa = a_ok ? (int64_t)BK * g.lda : 0, sb = b_ok ? (int64_t)BK * g.ldb : 0;
  const int full_tiles = (int)((kend - kbeg) / BK);      // tiles with every k-row below kend
  auto issue = [&](int t) {
    float* st = smem + (t % DMA_STAGES) * STAGE;
    if (t < full_tiles) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int kr = 4 * wave + 2 * j;
        dma_1k(pa[j], st + kr * BM);
        dma_1k(pb[j], st + TA::LDS_FLOATS + kr * BN);
      }
    } else {               // the ragged last tile: rows at or beyond kend come from the zero page
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int kr = 4 * wave + 2 * j;
        const bool kin = kbeg + (int64_t)t * BK + kr + h < kend;
        dma_1k(kin ? pa[j] : g_zero_page, st + kr * BM);
        dma_1k(kin ? pb[j] : g_zero_page, st + TA::LDS_FLOATS + kr * BN);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) { pa[j] += sa; pb[j] += sb; }
  };
  constexpr int PER_TILE = 4;     // loads per wave per tile
  if (ntiles > 0) issue(0);
  if (ntiles > 1) issue(1);
  if (ntiles > 2) issue(2);
  if (ntiles > 2) wait_vm<2 * PER_TILE>(); else if (ntiles > 1) wait_vm<PER_TILE>(); else wait_vm<0>();
  __syncthreads();
  WG_STAMP(1);

  constexpr int KSTEPS = BK / 8;
  float af[2][NBM][4], bf[2][NBN][4];
  auto load_frags = [&](int buf, const float* a_s, int s) {
    const float* b_s = a_s + TA::LDS_FLOATS;
#pragma unroll
    for (int i = 0; i < NBM; ++i) TA::frag(af[buf][i], a_s, wm * TM + i * 32 + l31, s, h);
#pragma unroll
    for (int j = 0; j < NBN; ++j) TB::frag(bf[buf][j], b_s, wn * TN + j * 32 + l31, s, h);
  };
  if (ntiles > 0) load_frags(0, smem, 0);

  for (int t = 0; t < ntiles; ++t) {
    const float* a_s = smem + (t % DMA_STAGES) * STAGE;
    const float* a_n = smem + ((t + 1) % DMA_STAGES) * STAGE;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
#ifdef CLICA_WGRAD_TRACE
      if (t == 20 && s == 0) WG_STAMP(4);
      if (t == 20 && s == 1) WG_STAMP(7);
      if (t == 21 && s == 0) { if (g_wtrace && (threadIdx.x & 63) == 0) g_wtrace[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + 0] = clock64() - g_wtrace[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + 4]; }
#endif
      if (s + 1 < KSTEPS) load_frags((s + 1) & 1, a_s, s + 1);
      else if (t + 1 < ntiles) load_frags(0, a_n, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int i = 0; i < NBM; ++i)
#pragma unroll
          for (int j = 0; j < NBN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][i][tt], bf[s & 1][j][tt], acc[i][j], 0, 0, 0);
      if (s == 0) {
        __builtin_amdgcn_sched_barrier(0);
#ifdef CLICA_WGRAD_TRACE
        if (t == 20) WG_STAMP(5);
#endif
        // tile t+1 must be complete (it is read from the last k-step of this iteration on); tile t+2 may fly
        if (t + 2 < ntiles) wait_vm<PER_TILE>(); else wait_vm<0>();
        __syncthreads();            // ... for every wave; and every wave is done with stage (t-1) % 4
#ifdef CLICA_WGRAD_TRACE
        if (t == 20) WG_STAMP(6);
#endif
        // (tried: the four DMAs spread over k-steps 1..3, the two waves of a SIMD one step apart -- k-loop 218.9k vs 214.6k
        // cycles per 43-tile item, tools/wgrad_trace.py: issuing them together right behind the barrier is the better place)
        if (t + 3 < ntiles) issue(t + 3);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (g.dbias_slab && !ones_col && bx == 0 && threadIdx.x < BM) {     // no padding column to borrow: sum the LDS tile
#pragma unroll 8
      for (int k = 0; k < BK; ++k) colsum += a_s[k * TA::LDS_LD + threadIdx.x];
    }
  }

  WG_STAMP(2);
  float* Cbase = g.C + (int64_t)bz * g.M * g.ldc;
#pragma unroll
  for (int i = 0; i < NBM; ++i) {
#pragma unroll
    for (int j = 0; j < NBN; ++j) {
      const int64_t col = n0 + wn * TN + j * 32 + l31;
      const bool is_db = ones_col && col == g.N;
      if (col >= g.N && !is_db) continue;
      float* dst = is_db ? g.dbias_slab + (int64_t)bz * g.M : Cbase + col;
      const int64_t ld = is_db ? 1 : g.ldc;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= g.M) continue;
        dst[row * ld] = acc[i][j][r];
      }
    }
  }
  if (g.dbias_slab && !ones_col && bx == 0 && threadIdx.x < BM) {
    const int64_t row = m0 + threadIdx.x;
    if (row < g.M) g.dbias_slab[(int64_t)bz * g.M + row] = colsum;
  }
  WG_STAMP(3);
}
#ifdef CLICA_WGRAD_TRACE
extern "C" int clica_debug_wgrad_trace(unsigned long long* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(clica::gemm::g_wtrace), &buf, sizeof(buf));
}
#endif

// ---- the same direct global -> LDS scheme on a 256 x 128 (A_WIDE) or 128 x 256 output tile ------------------------------------
// Why a second tile shape: the 128 x 128 item spends ~18 % of its k-loop outside MFMA issue (tools/wgrad_trace.py: 4992 cycles
// per 32-deep tile vs 4096), and what it pays per tile -- 32 one-KB DMA pieces, 12 LDS fragment dwords per 8 MFMAs, one
// workgroup barrier -- does not grow with the tile's area.  A 256 x 128 tile does 2x the MFMAs per barrier with 1.5x the DMA
// pieces and 1.33x the fragment reads (16 dwords per 16 MFMAs), and the n = 10 encoder's 28 such tiles x 9 contraction splits
// are 252 equal items = ONE round of the 256 CUs (the 128 x 128 plan needs two rounds of 504, i.e. two prologues / epilogues
// per CU).  Eight waves, wave tile 64 x 64 (2 x 2 accumulator blocks of 32 x 32), three 48 KB LDS stages: the loads of tile
// t + 2 are issued behind the barrier of iteration t (their stage was read in iteration t - 1) and first waited for in t + 1.
// DMA piece layout: the wide operand's k-row is 256 floats = one 1 KB piece (lane l -> columns 4 l .. 4 l + 3); the narrow
// operand's piece is two k-rows of 128 floats (lanes 32..63 = the second row), as in wgrad_dma_body.
#ifndef CLICA_WGRAD_STAGGER
#define CLICA_WGRAD_STAGGER 1
#endif
#ifndef CLICA_WGRAD_ABLATE      // timing ablations (WRONG results): 1 no DMA in the loop, 2 no barrier, 4 no fragment reads
#define CLICA_WGRAD_ABLATE 0
#endif
template <bool A_WIDE>
__device__ __forceinline__ void wgrad_dma_body2(const Args& g, const int bx, const int by, const int bz) {
  constexpr int BM = A_WIDE ? 256 : 128, BN = A_WIDE ? 128 : 256;
  constexpr int WN = BN / 64, NBM = 2, NBN = 2, TM = 64, TN = 64;
  constexpr int STG = 3, STAGE = BK * (BM + BN);        // floats per stage: [A: BK x BM][B: BK x BN]
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave % WN, h = lane >> 5, l31 = lane & 31;
  const int64_t m0 = (int64_t)by * BM, n0 = (int64_t)bx * BN;
  const int64_t kbeg = (int64_t)bz * g.k_per_split, kend = min(g.Kc, kbeg + g.k_per_split);
  const int ntiles = (int)((kend - kbeg + BK - 1) / BK);
  WG_STAMP(0);
  f32x16 acc[NBM][NBN];
#pragma unroll
  for (int i = 0; i < NBM; ++i)
#pragma unroll
    for (int j = 0; j < NBN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // source pointers of this lane's pieces.  Wide operand: k-rows 4 wave + j (j = 0..3), columns 4 lane ..; narrow operand:
  // k-rows 4 wave + 2 j + h (j = 0, 1), columns 4 l31 ...  Lanes whose columns are out of range sit on the zero page (stride 0);
  // the B lane whose first column is exactly N sits on {1, 0, 0, 0}: the matrix cores return db = dZ^T 1 in that padding column.
  const bool ones_col = g.dbias_slab && (g.N % BN != 0);
  // no padding column to carry the ones (the layer's input width is a multiple of the tile: 128, 256, 512 ...): the first
  // column tile sums dZ's columns from the staged A tile instead, one output feature per thread
  const bool need_colsum = g.dbias_slab && !ones_col && bx == 0 && (int)threadIdx.x < BM;
  float colsum = 0.f;
  const float* wide = A_WIDE ? g.A : g.B; const int64_t ldw = A_WIDE ? g.lda : g.ldb;
  const float* narr = A_WIDE ? g.B : g.A; const int64_t ldn = A_WIDE ? g.ldb : g.lda;
  const int64_t w0 = (A_WIDE ? m0 : n0) + 4 * lane, wlim = A_WIDE ? g.M : g.N;
  const int64_t q0 = (A_WIDE ? n0 : m0) + 4 * l31, qlim = A_WIDE ? g.N : g.M;
  const bool w_ok = w0 < wlim, q_ok = q0 < qlim;
  const float* w_pad = (!A_WIDE && ones_col && w0 == g.N) ? g_one_page : g_zero_page;     // the wide operand is B when !A_WIDE
  const float* q_pad = (A_WIDE && ones_col && q0 == g.N) ? g_one_page : g_zero_page;
  const float* pw[4]; const float* pn[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) pw[j] = w_ok ? wide + (kbeg + 4 * wave + j) * ldw + w0 : w_pad;
#pragma unroll
  for (int j = 0; j < 2; ++j) pn[j] = q_ok ? narr + (kbeg + 4 * wave + 2 * j + h) * ldn + q0 : q_pad;
  const int64_t sw = w_ok ? (int64_t)BK * ldw : 0, sn = q_ok ? (int64_t)BK * ldn : 0;
  constexpr int WOFF = A_WIDE ? 0 : BK * BM;            // float offset of the wide / narrow operand inside a stage
  constexpr int NOFF = A_WIDE ? BK * BM : 0;
  constexpr int WLD = A_WIDE ? BM : BN, NLD = A_WIDE ? BN : BM;
  const int full_tiles = (int)((kend - kbeg) / BK);
  auto issue = [&](int t) {
    float* st = smem + (t % STG) * STAGE;
    const bool full = t < full_tiles;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kr = 4 * wave + j;
      const bool kin = full || kbeg + (int64_t)t * BK + kr < kend;
      dma_1k(kin ? pw[j] : g_zero_page, st + WOFF + kr * WLD);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kr = 4 * wave + 2 * j;
      const bool kin = full || kbeg + (int64_t)t * BK + kr + h < kend;
      dma_1k(kin ? pn[j] : g_zero_page, st + NOFF + kr * NLD);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) pw[j] += sw;
#pragma unroll
    for (int j = 0; j < 2; ++j) pn[j] += sn;
  };
  constexpr int PER_TILE = 6;     // loads per wave per tile
  if (ntiles > 0) issue(0);
  if (ntiles > 1) issue(1);
  if (ntiles > 1) wait_vm<PER_TILE>(); else wait_vm<0>();
  __syncthreads();

  constexpr int KSTEPS = BK / 8;
  float af[2][NBM][4], bf[2][NBN][4];
  auto load_frags = [&](int buf, const float* st, int s) {
    const float* a_s = st;                   // [BK][BM]
    const float* b_s = st + BK * BM;         // [BK][BN]
#pragma unroll
    for (int i = 0; i < NBM; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) af[buf][i][t] = a_s[(8 * s + 4 * h + t) * BM + wm * TM + i * 32 + l31];
#pragma unroll
    for (int j = 0; j < NBN; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) bf[buf][j][t] = b_s[(8 * s + 4 * h + t) * BN + wn * TN + j * 32 + l31];
  };
  if (ntiles > 0) load_frags(0, smem, 0);
  WG_STAMP(1);
  for (int t = 0; t < ntiles; ++t) {
    const float* st_c = smem + (t % STG) * STAGE;
    const float* st_n = smem + ((t + 1) % STG) * STAGE;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      if (t == 20 && s == 0) WG_STAMP(4);
      if (t == 20 && s == 1) WG_STAMP(7);
#if !(CLICA_WGRAD_ABLATE & 4)
      if (s + 1 < KSTEPS) load_frags((s + 1) & 1, st_c, s + 1);
      else if (t + 1 < ntiles) load_frags(0, st_n, 0);
#endif
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int i = 0; i < NBM; ++i)
#pragma unroll
          for (int j = 0; j < NBN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][i][tt], bf[s & 1][j][tt], acc[i][j], 0, 0, 0);
      if (s == 0) {
        __builtin_amdgcn_sched_barrier(0);
        if (t == 20) WG_STAMP(5);
        // tile t+1 must be complete (read from the last k-step of this iteration on): it is the only one in flight here
#if !(CLICA_WGRAD_ABLATE & 1)
        wait_vm<0>();
#endif
#if !(CLICA_WGRAD_ABLATE & 2)
        __syncthreads();            // ... for every wave; and every wave is done with stage (t-1) % 3
#endif
        if (t == 20) WG_STAMP(6);
#if CLICA_WGRAD_ABLATE & 1
#elif CLICA_WGRAD_STAGGER
        if (wave < 4 && t + 2 < ntiles) issue(t + 2);
#else
        if (t + 2 < ntiles) issue(t + 2);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
#if CLICA_WGRAD_STAGGER
      // The six DMA requests of a wave take 1600-2000 cycles to issue (tools/wgrad_trace.py: the texture path moves 48 KB per
      // tile and a wave issues in order), and right behind the barrier BOTH waves of every SIMD were in that phase together:
      // the matrix pipe idled for ~15 % of every tile.  Waves 0..3 still issue there; waves 4..7 (the other wave of each
      // SIMD) run k-step 1 first and issue afterwards, so one wave per SIMD always has MFMAs to issue.
      if (s == 1) {
        __builtin_amdgcn_sched_barrier(0);
#if !(CLICA_WGRAD_ABLATE & 1)
        if (wave >= 4 && t + 2 < ntiles) issue(t + 2);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    }
    if (need_colsum) {
#pragma unroll 8
      for (int k = 0; k < BK; ++k) colsum += st_c[k * BM + threadIdx.x];      // rows past kend were staged as zeros
    }
  }
  WG_STAMP(2);
  float* Cbase = g.C + (int64_t)bz * g.M * g.ldc;
#pragma unroll
  for (int i = 0; i < NBM; ++i) {
#pragma unroll
    for (int j = 0; j < NBN; ++j) {
      const int64_t col = n0 + wn * TN + j * 32 + l31;
      const bool is_db = ones_col && col == g.N;
      if (col >= g.N && !is_db) continue;
      float* dst = is_db ? g.dbias_slab + (int64_t)bz * g.M : Cbase + col;
      const int64_t ld = is_db ? 1 : g.ldc;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= g.M) continue;
        dst[row * ld] = acc[i][j][r];
      }
    }
  }
  if (need_colsum) {
    const int64_t row = m0 + threadIdx.x;
    if (row < g.M) g.dbias_slab[(int64_t)bz * g.M + row] = colsum;
  }
  WG_STAMP(3);
}
constexpr size_t kBody2LdsBytes = (size_t)3 * BK * (256 + 128) * sizeof(float);

// ---- weight gradient of a layer with one TINY dimension (the n-wide first / last encoder layer) on the vector ALU ----
// dW[N,K] = dZ^T X with S = min(N, K) <= 16 and Lg = max(N, K) <= THREADS.  As a 128 x 128 MFMA tile such a layer is
// > 90 % padding AND takes the slow register-staged body (its rows are not 16-byte aligned), which made its items the
// tail of the grouped launch (+37 us of 244 at n = 10).  Here: thread (group gq, large index l) keeps the S outputs
// (l, 0..S-1) in registers; per batch row it reads its own large-operand element and the row's S small-operand elements
// (wave-uniform ds_read_b128 broadcasts) and does S FMAs; the THREADS / Lg groups take interleaved rows and are summed
// through LDS in fixed order at the end.  Row tiles are register-prefetched one tile ahead.  A work item covers
// k_per_split batch rows and writes one slab, exactly like the MFMA items (same deterministic slab reduction).
// (MAXG, TINY_MAX_S, TINY_ROWS, TINY_MAX_LG: wgrad_shared.h)
// global -> LDS copy of `count` floats of a row block: flat float4 when the block is contiguous and aligned (the
// encoder's activations / gradients are), element-wise otherwise.  Eight loads are in flight per thread and pass.
template <int THREADS>
__device__ __forceinline__ void tiny_stage(float* dst, int dst_ld, const float* __restrict__ src, int64_t ld, int width,
                                           int64_t r0, int rows_valid, int rows_tile) {
  const bool flat = ld == width && dst_ld == width && ((reinterpret_cast<uintptr_t>(src + r0 * ld) & 15) == 0) && ((rows_valid * width) % 4 == 0);
  if (flat) {
    const float4* s4 = reinterpret_cast<const float4*>(src + r0 * ld);
    float4* d4 = reinterpret_cast<float4*>(dst);
    const int n4 = rows_valid * width / 4, t4 = rows_tile * width / 4;
    for (int i0 = threadIdx.x; i0 < t4; i0 += 8 * THREADS) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int i = i0 + u * THREADS; v[u] = i < n4 ? s4[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int i = i0 + u * THREADS; if (i < t4) d4[i] = v[u]; }
    }
    // a tile whose float count is not a multiple of 4 past n4 * 4: covered by rows_valid * width % 4 == 0 above
  } else {
    const int nt = rows_tile * width;
    for (int i0 = threadIdx.x; i0 < nt; i0 += 8 * THREADS) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * THREADS;
        const int r = i / width, c = i - r * width;
        const bool ok = i < nt && r < rows_valid;
        const float x = src[ok ? (r0 + r) * ld + c : 0];
        v[u] = ok ? x : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * THREADS;
        if (i < nt) { const int r = i / width, c = i - r * width; dst[r * dst_ld + c] = v[u]; }
      }
    }
  }
}
// S4 = ceil(S / 4): the small dimension in float4 units, a compile-time constant so that the row loop is branch-free and its
// LDS reads can be batched four rows deep (the first version waited for every ds_read_b128 in turn: ~11 of its 19 us).
template <int THREADS, int S4>
__device__ __forceinline__ void wgrad_tiny_body(const Args& g, const int bz) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = (int)g.M, K = (int)g.N;               // dW is [N][K]; A = dZ [rows][N], B = X [rows][K]
  const bool a_large = N >= K;                          // large operand: A (outputs (l, u) = dW[l][u]) or B (dW[u][l])
  const int Lg = a_large ? N : K, S = a_large ? K : N;
  const float* Lp = a_large ? g.A : g.B; const int64_t ldl = a_large ? g.lda : g.ldb;
  const float* Sp = a_large ? g.B : g.A; const int64_t lds_ = a_large ? g.ldb : g.lda;
  const int G = THREADS / Lg;                           // row-interleaved thread groups
  const int gq = threadIdx.x / Lg, l = threadIdx.x - gq * Lg;
  const bool active = gq < G;
  float* Ls = smem;                                     // [TINY_ROWS][Lg]
  float* Ss = smem + TINY_ROWS * Lg;                    // [TINY_ROWS][16]: columns >= S and rows beyond the chunk are zeros
  const int64_t kbeg = (int64_t)bz * g.k_per_split, kend = min(g.Kc, kbeg + g.k_per_split);
  float4 acc[S4];
#pragma unroll
  for (int u = 0; u < S4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  float accl = 0.f;                                     // column sum of the large operand (db when A is large)
  float accs = 0.f;                                     // column sum of the small operand, thread u < S (db when A is small)
  const bool want_db = g.dbias_slab != nullptr;
  const int nit = (TINY_ROWS + G - 1) / G;              // row iterations per tile and thread group
  // register prefetch of the NEXT tile (issued before this tile's arithmetic): the large operand's tile when it is one flat,
  // 16-byte aligned block (the encoder's activations / gradients are), and the small operand always
  constexpr int LQ = (TINY_ROWS * TINY_MAX_LG / 4 + THREADS - 1) / THREADS;        // float4 per thread
  constexpr int RPP = THREADS / TINY_MAX_S, SQ = TINY_ROWS / RPP;
  float4 lq[LQ]; float sq[SQ];
  const bool flat = ldl == Lg && ((reinterpret_cast<uintptr_t>(Lp) & 15) == 0) && (((int64_t)Lg * kbeg) % 4 == 0) &&
                    ((Lg * TINY_ROWS) % 4 == 0);
  const int t4 = TINY_ROWS * Lg / 4;
  const int sc = threadIdx.x & (TINY_MAX_S - 1), sr0 = threadIdx.x >> 4;
  auto fetch = [&](int64_t r0) {
    const int rv = (int)min((int64_t)TINY_ROWS, kend - r0);
    if (flat) {
      const float4* s4 = reinterpret_cast<const float4*>(Lp + r0 * ldl);
      const int n4 = rv * Lg / 4;          // rv * Lg is a multiple of 4 for full tiles; a ragged last tile takes the tail below
#pragma unroll
      for (int i = 0; i < LQ; ++i) {
        const int idx = threadIdx.x + i * THREADS;
        lq[i] = idx < n4 ? s4[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int ps = 0; ps < SQ; ++ps) {
      const int r = sr0 + ps * RPP;
      const bool ok = sc < S && r < rv;
      const float x = Sp[ok ? (r0 + r) * lds_ + sc : 0];
      sq[ps] = ok ? x : 0.f;
    }
  };
  auto stash = [&](int64_t r0) {
    const int rv = (int)min((int64_t)TINY_ROWS, kend - r0);
    if (flat && (rv * Lg) % 4 == 0) {
#pragma unroll
      for (int i = 0; i < LQ; ++i) {
        const int idx = threadIdx.x + i * THREADS;
        if (idx < t4) reinterpret_cast<float4*>(Ls)[idx] = lq[i];
      }
    } else {
      tiny_stage<THREADS>(Ls, Lg, Lp, ldl, Lg, r0, rv, TINY_ROWS);
    }
#pragma unroll
    for (int ps = 0; ps < SQ; ++ps) Ss[(sr0 + ps * RPP) * TINY_MAX_S + sc] = sq[ps];
  };
  if (kbeg < kend) fetch(kbeg);
  for (int64_t r0 = kbeg; r0 < kend; r0 += TINY_ROWS) {
    __syncthreads();                                    // every thread is done with the previous tile
    stash(r0);
    __syncthreads();
    if (r0 + TINY_ROWS < kend) fetch(r0 + TINY_ROWS);   // in flight during this tile's arithmetic
    if (active) {
      for (int i0 = 0; i0 < nit; i0 += 4) {             // four rows per pass: 4 + 4 S4 LDS reads issued before the FMAs
        float v[4]; float4 sv[4][S4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = gq + (i0 + j) * G;
          const bool ok = (i0 + j) < nit && r < TINY_ROWS;
          const int rc = ok ? r : 0;
          const float x = Ls[rc * Lg + l];
          v[j] = ok ? x : 0.f;
#pragma unroll
          for (int u = 0; u < S4; ++u) sv[j][u] = *reinterpret_cast<const float4*>(&Ss[rc * TINY_MAX_S + 4 * u]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          accl += v[j];
#pragma unroll
          for (int u = 0; u < S4; ++u) {
            acc[u].x = fmaf(v[j], sv[j][u].x, acc[u].x); acc[u].y = fmaf(v[j], sv[j][u].y, acc[u].y);
            acc[u].z = fmaf(v[j], sv[j][u].z, acc[u].z); acc[u].w = fmaf(v[j], sv[j][u].w, acc[u].w);
          }
        }
      }
    }
    if (want_db && !a_large && (int)threadIdx.x < S) {  // db = column sums of the small operand (rows beyond the chunk are zeros)
#pragma unroll 8
      for (int r = 0; r < TINY_ROWS; ++r) accs += Ss[r * TINY_MAX_S + threadIdx.x];
    }
  }
  // fixed-order sum over the G groups through LDS: red[gq][l][0..16]  (17 floats per (gq, l): conflict-free stride)
  __syncthreads();
  constexpr int RS = TINY_MAX_S + 1;
  float* red = smem;
  if (active) {
#pragma unroll
    for (int u = 0; u < S4; ++u) {
      float* d = &red[(gq * Lg + l) * RS + 4 * u];
      d[0] = acc[u].x; d[1] = acc[u].y; d[2] = acc[u].z; d[3] = acc[u].w;
    }
    red[(gq * Lg + l) * RS + TINY_MAX_S] = accl;
  }
  __syncthreads();
  float* Cbase = g.C + (int64_t)bz * g.M * g.ldc;
  for (int o = threadIdx.x; o < Lg * S; o += THREADS) {
    const int ll = o / S, u = o - ll * S;
    float t = 0.f;
    for (int q = 0; q < G; ++q) t += red[(q * Lg + ll) * RS + u];
    if (a_large) Cbase[(int64_t)ll * g.ldc + u] = t; else Cbase[(int64_t)u * g.ldc + ll] = t;
  }
  if (want_db) {
    float* dbs = g.dbias_slab + (int64_t)bz * g.M;
    if (a_large) {
      for (int ll = threadIdx.x; ll < Lg; ll += THREADS) {
        float t = 0.f;
        for (int q = 0; q < G; ++q) t += red[(q * Lg + ll) * RS + TINY_MAX_S];
        dbs[ll] = t;
      }
    } else if ((int)threadIdx.x < S) {
      dbs[threadIdx.x] = accs;
    }
  }
}
// One launch for all tiny-dimension layers of a group, BEFORE the grouped MFMA launch: grid (row chunks, problems).  The body is
// latency-bound per tile (a 64-row tile is ~0.2 us of FMAs behind a ~1-2 us fetch), so the rows are cut into many short
// chunks that all run at once on the otherwise idle chip (~5 us) -- as long items inside the MFMA launch they crawled
// behind its HBM traffic and became its tail.
__global__ __launch_bounds__(TINY_THREADS) void wgrad_tiny_k(TinyArgs T) {
  const Args& g = T.p[blockIdx.y];
  const int S = (int)(g.M < g.N ? g.M : g.N);
  switch ((S + 3) / 4) {
    case 1: wgrad_tiny_body<TINY_THREADS, 1>(g, (int)blockIdx.x); break;
    case 2: wgrad_tiny_body<TINY_THREADS, 2>(g, (int)blockIdx.x); break;
    case 3: wgrad_tiny_body<TINY_THREADS, 3>(g, (int)blockIdx.x); break;
    default: wgrad_tiny_body<TINY_THREADS, 4>(g, (int)blockIdx.x); break;
  }
}
static bool tiny_eligible(int64_t N, int64_t K, int threads) {
  const int64_t S = N < K ? N : K, Lg = N < K ? K : N;
  return S <= TINY_MAX_S && Lg <= TINY_MAX_LG && Lg <= threads;
}

// ---- grouped weight gradients: every layer's dW/db slabs in ONE launch ------------------------------
// Work item = (problem, contraction split, tile); all items cover the same number of batch rows, so the
// launch is a few full rounds of equal-length workgroups instead of one ragged launch (+ reduction) per layer.
struct GroupArgs {
  int n, total;
  int first[MAXG + 1];    // first work item of each problem
  int gx[MAXG], gy[MAXG]; // tiles along N / M
  int vec[MAXG];          // 1: direct global -> LDS 128 x 128 body; 3 / 4: its 256 x 128 / 128 x 256 variant; 0: register-staged body
  Args p[MAXG];
};

template <int BM, int BN, int WM, int WN, int STAGES>
__global__ __launch_bounds__(64 * WM * WN) void wgrad_group_k(GroupArgs G) {
  const int id = xcd_contiguous(blockIdx.x, G.total);
  int q = 0;
#pragma unroll
  for (int i = 1; i < MAXG; ++i) q += (i < G.n && id >= G.first[i]) ? 1 : 0;
  const int local = id - G.first[q];
  const int gx = G.gx[q], tiles = gx * G.gy[q];
  const int bz = local / tiles, t = local - bz * tiles;
  const int by = t / gx, bx = t - by * gx;
  if (G.vec[q] == 1) wgrad_dma_body<WM, WN>(G.p[q], bx, by, bz);
  else if (G.vec[q] == 3) wgrad_dma_body2<true>(G.p[q], bx, by, bz);
  else if (G.vec[q] == 4) wgrad_dma_body2<false>(G.p[q], bx, by, bz);
  else gemm_body<BM, BN, WM, WN, STAGES, false, false, EPI_SLAB, false>(G.p[q], bx, by, bz);
}

// dW[i][j] (+)= sum_s slab[s][i][j];  db[i] (+)= sum_s dbslab[s][i].
// A block owns 64 consecutive output units (float4 when the row length allows, else float); its
// four waves each sum a quarter of the splits (s = w, w+4, ...) with four loads in flight, then wave 0
// adds the four partial sums in wave order: fixed summation order, short dependent-load chains.
template <bool VEC4>
__device__ __forceinline__ void reduce_units(const float* __restrict__ src, int splits, int64_t total, int64_t unit0,
                                             float4 (*red)[64], float4& t, int64_t& e) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int U = VEC4 ? 4 : 1;
  constexpr int W = RED_THREADS / 64;
  e = (unit0 + lane) * U;
  float4 p[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) p[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < total && splits > 32) {
    // many short slabs (the tiny-dimension layers: ~100 row chunks from wgrad_tiny_k, 256 workgroup partials from the backward
    // chain's tail): sixteen loads in flight per wave and pass; the partial sums keep a fixed association (u, u + 4, u + 8, u + 12
    // into p[u]) so the result does not depend on timing
    for (int s0 = w; s0 < splits; s0 += 16 * W) {
      float4 q16[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int s = s0 + u * W;
        q16[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s < splits) {
          if (VEC4) q16[u] = *reinterpret_cast<const float4*>(src + (int64_t)s * total + e);
          else q16[u].x = src[(int64_t)s * total + e];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        p[u].x += (q16[u].x + q16[u + 4].x) + (q16[u + 8].x + q16[u + 12].x); p[u].y += (q16[u].y + q16[u + 4].y) + (q16[u + 8].y + q16[u + 12].y);
        p[u].z += (q16[u].z + q16[u + 4].z) + (q16[u + 8].z + q16[u + 12].z); p[u].w += (q16[u].w + q16[u + 4].w) + (q16[u + 8].w + q16[u + 12].w);
      }
    }
  } else if (e < total) {
    for (int s0 = w; s0 < splits; s0 += 4 * W) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int s = s0 + u * W;
        if (s < splits) {
          if (VEC4) {
            const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)s * total + e);
            p[u].x += v.x; p[u].y += v.y; p[u].z += v.z; p[u].w += v.w;
          } else {
            p[u].x += src[(int64_t)s * total + e];
          }
        }
      }
    }
  }
  t.x = (p[0].x + p[1].x) + (p[2].x + p[3].x); t.y = (p[0].y + p[1].y) + (p[2].y + p[3].y);
  t.z = (p[0].z + p[1].z) + (p[2].z + p[3].z); t.w = (p[0].w + p[1].w) + (p[2].w + p[3].w);
  red[w][lane] = t;
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int k = 1; k < W; ++k) { const float4 v = red[k][lane]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
  }
  __syncthreads();
}

template <bool VEC4>
__global__ __launch_bounds__(RED_THREADS) void slab_reduce_k(const float* __restrict__ slab, int splits, int64_t M, int64_t N,
                                                            float* __restrict__ dW, int64_t lddw,
                                                            const float* __restrict__ dbslab, float* __restrict__ db,
                                                            int accumulate, int dw_blocks) {
  __shared__ float4 red[RED_THREADS / 64][64];
  const int w = threadIdx.x >> 6;
  float4 t; int64_t e;
  if ((int)blockIdx.x < dw_blocks) {
    const int64_t total = M * N;
    reduce_units<VEC4>(slab, splits, total, (int64_t)blockIdx.x * 64, red, t, e);
    if (w == 0 && e < total) {
      const int64_t i = e / N, j = e - i * N;
      if (VEC4) {
        float4* dst = reinterpret_cast<float4*>(dW + i * lddw + j);
        if (accumulate) { const float4 o = *dst; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
        *dst = t;
      } else {
        float* dst = dW + i * lddw + j;
        *dst = accumulate ? (*dst + t.x) : t.x;
      }
    }
  } else {   // trailing blocks: the bias slab (M entries per split), same scheme
    reduce_units<false>(dbslab, splits, M, (int64_t)((int)blockIdx.x - dw_blocks) * 64, red, t, e);
    if (w == 0 && e < M) db[e] = accumulate ? (db[e] + t.x) : t.x;
  }
}

// grouped variant: the slabs of up to MAXG problems reduced by one launch
// (struct ReduceGroupArgs: wgrad_shared.h)
// Few, large slabs (wide layers: <= 8 contraction splits of a 2000 x 2000 gradient = 16 MB each): one thread per float4 of the
// result, all its <= 8 slab loads in flight at once, summed in split order (deterministic).  The 64-units-per-block scheme above
// is built for many short slabs and spends a whole workgroup (and an LDS pass) on 64 float4s: 138 us for a 2000 x 2000 layer with
// four splits (80 MB, 0.6 TB/s) against ~30 us here.
constexpr int FLAT_MAX_SPLITS = 24;      // (round 5: 8 -> 24.  The whole-stack plan cuts its 500 x 500 problems into 18 splits; on the
                                         //  64-units-per-block scheme they made this launch 15 us for 31 MB)
static inline bool reduce_flat(int splits, bool v4) { return v4 && splits <= FLAT_MAX_SPLITS; }
__device__ __forceinline__ void adam_apply4(const ReduceAdam& A, const adam::Consts& c, const float* dst, const float4& g, float4 pp, float4 mm, float4 vv) {
  const int64_t off = dst - A.gbase;
  adam::update(pp.x, g.x, mm.x, vv.x, c, A.b1, A.b2, A.eps, A.gscale); adam::update(pp.y, g.y, mm.y, vv.y, c, A.b1, A.b2, A.eps, A.gscale);
  adam::update(pp.z, g.z, mm.z, vv.z, c, A.b1, A.b2, A.eps, A.gscale); adam::update(pp.w, g.w, mm.w, vv.w, c, A.b1, A.b2, A.eps, A.gscale);
  *reinterpret_cast<float4*>(A.p + off) = pp; *reinterpret_cast<float4*>(A.m + off) = mm; *reinterpret_cast<float4*>(A.v + off) = vv;
}
__device__ __forceinline__ void adam_apply1(const ReduceAdam& A, const adam::Consts& c, const float* dst, float g) {
  const int64_t off = dst - A.gbase;
  float pp = A.p[off], mm = A.m[off], vv = A.v[off];
  adam::update(pp, g, mm, vv, c, A.b1, A.b2, A.eps, A.gscale);
  A.p[off] = pp; A.m[off] = mm; A.v[off] = vv;
}
__global__ __launch_bounds__(RED_THREADS) void slab_reduce_group_k(ReduceGroupArgs G) {
  __shared__ float4 red[RED_THREADS / 64][64];
  __shared__ adam::Consts s_c;
  const ReduceAdam& A = G.adam;
  const bool with_adam_ = A.p != nullptr;
  const int nfront = (with_adam_ && A.s16_state) ? s16::kS16UpdateBlocks : 0;
  if ((int)blockIdx.x < nfront) {
    s16::split16_update_tensor(reinterpret_cast<s16::Split16State*>(A.s16_state), A.s16_layers, (int)blockIdx.x, const_cast<int*>(A.step_dev));
    return;
  }
  const int blk = (int)blockIdx.x - nfront;
  // the guard (split16.h): the step is poisoned -> the gradients are still reduced (inspection), the optimizer is not applied
  const bool with_adam = with_adam_ && !(A.s16_state && s16::s16_step_poisoned(reinterpret_cast<const s16::Split16State*>(A.s16_state)));
  if (with_adam && threadIdx.x == RED_THREADS - 1) s_c = adam::consts_of(A.step_dev[0] + A.t_offset, A.lr, A.b1, A.b2);
  const int w = threadIdx.x >> 6;
  int q = 0;
#pragma unroll
  for (int i = 1; i < MAXG; ++i) q += (i < G.n && blk >= G.first[i]) ? 1 : 0;
  const int b = blk - G.first[q];
  const int64_t M = G.M[q], N = G.N[q];
  float4 t; int64_t e;
  if (b < G.dw_blocks[q]) {
    const int64_t total = M * N;
    if (G.vec4[q] && G.splits[q] <= FLAT_MAX_SPLITS) {       // flat: RED_THREADS float4 units per block (see slab_reduce_entry)
      e = ((int64_t)b * RED_THREADS + threadIdx.x) * 4;
      const bool live = e < total;
      if (!live && !with_adam) return;
      const int splits = live ? G.splits[q] : 0;
      const float* src = G.slab[q] + (live ? e : 0);
      const int64_t i = e / N, j = e - i * N;
      float4* dst = reinterpret_cast<float4*>(G.dW[q] + (live ? i * G.lddw[q] + j : 0));
      float4 v[FLAT_MAX_SPLITS];
#pragma unroll
      for (int sIdx = 0; sIdx < FLAT_MAX_SPLITS; ++sIdx)
        v[sIdx] = sIdx < splits ? *reinterpret_cast<const float4*>(src + (int64_t)sIdx * total) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 pp, mm, vv;                                     // the optimizer's operands travel with the slabs
      if (with_adam && live) {
        const int64_t off = reinterpret_cast<const float*>(dst) - A.gbase;
        pp = *reinterpret_cast<const float4*>(A.p + off); mm = *reinterpret_cast<const float4*>(A.m + off); vv = *reinterpret_cast<const float4*>(A.v + off);
      }
      t = v[0];
#pragma unroll
      for (int sIdx = 1; sIdx < FLAT_MAX_SPLITS; ++sIdx) { t.x += v[sIdx].x; t.y += v[sIdx].y; t.z += v[sIdx].z; t.w += v[sIdx].w; }
      if (live) {
        if (G.accumulate) { const float4 o = *dst; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
        *dst = t;
      }
      if (with_adam) {
        __syncthreads();                                     // s_c
        if (live) adam_apply4(A, s_c, reinterpret_cast<const float*>(dst), t, pp, mm, vv);
      }
      return;
    }
    if (G.vec4[q]) {
      reduce_units<true>(G.slab[q], G.splits[q], total, (int64_t)b * 64, red, t, e);
      if (w == 0 && e < total) {
        const int64_t i = e / N, j = e - i * N;
        float4* dst = reinterpret_cast<float4*>(G.dW[q] + i * G.lddw[q] + j);
        if (G.accumulate) { const float4 o = *dst; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
        *dst = t;
        if (with_adam) {
          const int64_t off = reinterpret_cast<const float*>(dst) - A.gbase;
          adam_apply4(A, s_c, reinterpret_cast<const float*>(dst), t, *reinterpret_cast<const float4*>(A.p + off),
                      *reinterpret_cast<const float4*>(A.m + off), *reinterpret_cast<const float4*>(A.v + off));
        }
      }
    } else {
      reduce_units<false>(G.slab[q], G.splits[q], total, (int64_t)b * 64, red, t, e);
      if (w == 0 && e < total) {
        const int64_t i = e / N, j = e - i * N;
        float* dst = G.dW[q] + i * G.lddw[q] + j;
        t.x = G.accumulate ? (*dst + t.x) : t.x;
        *dst = t.x;
        if (with_adam) adam_apply1(A, s_c, dst, t.x);
      }
    }
  } else {
    reduce_units<false>(G.dbslab[q], G.splits[q], M, (int64_t)(b - G.dw_blocks[q]) * 64, red, t, e);
    float* db = G.db[q];
    if (w == 0 && e < M) {
      t.x = G.accumulate ? (db[e] + t.x) : t.x;
      db[e] = t.x;
      if (with_adam) adam_apply1(A, s_c, db + e, t.x);
    }
  }
}

// one launch: ceil(units/64) blocks for dW followed by ceil(M/64) blocks for db
static void launch_slab_reduce(const float* slab, const float* dbslab, int splits, int64_t M, int64_t N, float* dW, int64_t lddw,
                               float* db, int accumulate, hipStream_t st) {
  const int64_t total = M * N;
  const bool v4 = (N % 4 == 0) && (lddw % 4 == 0) && ((reinterpret_cast<uintptr_t>(dW) & 15) == 0);
  const int dw_blocks = (int)ceil_div(v4 ? total / 4 : total, 64);
  const int db_blocks = db ? (int)ceil_div(M, 64) : 0;
  if (v4)
    hipLaunchKernelGGL(slab_reduce_k<true>, dim3((unsigned)(dw_blocks + db_blocks)), dim3(RED_THREADS), 0, st, slab, splits, M, N, dW, lddw,
                       dbslab, db, accumulate, dw_blocks);
  else
    hipLaunchKernelGGL(slab_reduce_k<false>, dim3((unsigned)(dw_blocks + db_blocks)), dim3(RED_THREADS), 0, st, slab, splits, M, N, dW, lddw,
                       dbslab, db, accumulate, dw_blocks);
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- tile configurations ------------------------------------------------------------------------
// id: BM x BN, waves WM x WN (64*WM*WN threads).  The 8-wave shapes keep two waves per SIMD on a
// CU that holds ONE workgroup, so LDS/global latency of one wave hides behind the MFMAs of the other.
struct Cfg { int bm, bn, wm, wn; };
constexpr Cfg kCfgs[] = {
    {192, 128, 2, 4},   // 0: 256 workgroups for M = 12288, N = 500 (one per CU, exactly one round)
    {128, 128, 2, 4},   // 1: wgrad / general
    {64, 128, 2, 2},    // 2: narrow outputs (N ~ 100): more workgroups along M
    {128, 128, 2, 2},   // 3: 4-wave fallback
    {192, 64, 2, 2},    // 4: 512 workgroups for M = 12288, N = 500: two independent 4-wave groups per CU
    {96, 128, 1, 4},    // 5: same count, other aspect
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

template <int BM, int BN, int WM, int WN, int STAGES, bool A_CONTIG, bool B_CONTIG, int EPI>
static int launch_cfg(const Args& g, int splits, hipStream_t st, const char* who) {
  constexpr int THREADS = 64 * WM * WN;
  // float4 loads need aligned rows and extents that keep every float4 fully in or out of range
  const bool vec = aligned16(g.A) && aligned16(g.B) && (g.lda % 4 == 0) && (g.ldb % 4 == 0) &&
                   (A_CONTIG ? (g.Kc % 4 == 0) : (g.M % 4 == 0)) && (B_CONTIG ? (g.Kc % 4 == 0) : (g.N % 4 == 0));
  dim3 grid((unsigned)ceil_div(g.N, BN), (unsigned)ceil_div(g.M, BM), (unsigned)splits), block(THREADS);
  constexpr size_t lds = STAGES * (Tile<BM, A_CONTIG, THREADS>::LDS_FLOATS + Tile<BN, B_CONTIG, THREADS>::LDS_FLOATS) * sizeof(float);
  if (vec) {
    auto k = gemm_k<BM, BN, WM, WN, STAGES, A_CONTIG, B_CONTIG, EPI, true>;
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    hipLaunchKernelGGL(k, grid, block, lds, st, g);
  } else {
    auto k = gemm_k<BM, BN, WM, WN, STAGES, A_CONTIG, B_CONTIG, EPI, false>;
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    hipLaunchKernelGGL(k, grid, block, lds, st, g);
  }
  return launch_status(who);
}

template <bool A_CONTIG, bool B_CONTIG, int EPI>
static int launch(int cfg, const Args& g, int splits, hipStream_t st, const char* who) {
  switch (cfg) {
    case 0: return launch_cfg<192, 128, 2, 4, 3, A_CONTIG, B_CONTIG, EPI>(g, splits, st, who);
    case 1: return launch_cfg<128, 128, 2, 4, 3, A_CONTIG, B_CONTIG, EPI>(g, splits, st, who);
    case 2: return launch_cfg<64, 128, 2, 2, 2, A_CONTIG, B_CONTIG, EPI>(g, splits, st, who);
    case 4: return launch_cfg<192, 64, 2, 2, 2, A_CONTIG, B_CONTIG, EPI>(g, splits, st, who);
    case 5: return launch_cfg<96, 128, 1, 4, 2, A_CONTIG, B_CONTIG, EPI>(g, splits, st, who);
    default: return launch_cfg<128, 128, 2, 2, 2, A_CONTIG, B_CONTIG, EPI>(g, splits, st, who);
  }
}

// Test / tuning hook (clica_set_tuning, include/clica.h): the per-layer entry points choose their body by SHAPE; a test that wants the other
// product path on a given shape (every shape through the MFMA template, another tile configuration) sets it here.  No environment reads:
// the launch path is called from autograd nodes in eager mode.
//   gemm_cfg_{fwd,dgrad,wgrad} = tile configuration id (-1: by shape);  skinny = 0 forces every shape through the MFMA template
struct Tuning { int cfg_fwd, cfg_dgrad, cfg_wgrad; bool skinny; };
static Tuning& tuning() { static Tuning t{-1, -1, -1, true}; return t; }
static bool use_skinny() { return tuning().skinny; }

// measured on MI355X (tools/gemm_bench.py): wide outputs -> 96x128 tiles, two 4-wave workgroups per
// CU; narrow outputs (N <= 128, one tile column) -> 64x128 for more workgroups along M
static int cfg_fwd(int64_t N) {
  const int e = tuning().cfg_fwd;
  return (e >= 0 && e < kNumCfgs) ? e : ((N > 128) ? 5 : 2);
}
// 64x128 measured best for every encoder layer shape (B operand is read with ds_read_b32)
static int cfg_dgrad() {
  const int e = tuning().cfg_dgrad;
  return (e >= 0 && e < kNumCfgs) ? e : 2;
}

// wgrad plan: output tiles x contraction splits ~ one round of workgroups, >= 4 K tiles per split
struct WgradPlan { int cfg, splits; int64_t k_per_split; };
static WgradPlan plan_wgrad(int64_t M /*rows of dW*/, int64_t N /*cols of dW*/, int64_t Kc) {
  WgradPlan p;
  p.cfg = (M > 128 && N > 128) ? 1 : 2;   // measured: 8-wave 128x128 wins on the square layers
  const int e = tuning().cfg_wgrad;
  if (e >= 0 && e < kNumCfgs) p.cfg = e;
  const Cfg c = kCfgs[p.cfg];
  const int64_t tiles = ceil_div(M, c.bm) * ceil_div(N, c.bn);
  int64_t want = kNumCU / tiles; if (want < 1) want = 1;
  const int64_t max_s = ceil_div(Kc, (int64_t)BK * 4);
  if (want > max_s) want = max_s;
  if (want < 1) want = 1;
  p.k_per_split = ceil_div(ceil_div(Kc, want), (int64_t)BK) * BK;
  p.splits = (int)ceil_div(Kc, p.k_per_split);
  return p;
}

// Grouped plan: 128x128 tiles for every layer, one common contraction split count chosen so that
// (tiles x splits) fills whole rounds of the 256 CUs (one 8-wave workgroup per CU) with few, long items.
struct GroupPlan { int splits; int64_t k_per_split; int tiles; int tiny_splits; int64_t tiny_kps; int n_tiny; };
constexpr int GBM = 128, GBN = 128;
// body of a problem from its shape alone (the workspace query and the launch must agree): 2 = tiny-dimension VALU kernel,
// 3 / 4 = 256 x 128 / 128 x 256 DMA tiles (a dimension beyond 128), 1 = 128 x 128 DMA tiles, 0 = register-staged (odd widths).
// Every MFMA item of a launch should cover the same work; mixed 128 x 128 and big-tile problems are legal, just not balanced.
static int group_kind(int32_t N, int32_t K) {
  if (tiny_eligible(N, K, TINY_THREADS)) return 2;
  if (N % 4 != 0 || K % 4 != 0) return 0;
  if (N > 128 || K > 128) return N >= K ? 3 : 4;
  return 1;
}
static void group_tile(int kind, int* bm, int* bn) {
  *bm = kind == 3 ? 256 : 128; *bn = kind == 4 ? 256 : 128;
}
static GroupPlan plan_wgrad_group(int64_t Mrows, int n, const int32_t* N, const int32_t* K) {
  GroupPlan p{};
  for (int l = 0; l < n; ++l) {
    const int kind = group_kind(N[l], K[l]);
    if (kind == 2) { ++p.n_tiny; continue; }
    int bm, bn; group_tile(kind, &bm, &bn);
    p.tiles += (int)(ceil_div(N[l], bm) * ceil_div(K[l], bn));
  }
  const int64_t max_s = std::max<int64_t>(1, std::min<int64_t>(64, ceil_div(Mrows, (int64_t)BK * 4)));
  double best = 1e300;
  p.splits = 1; p.k_per_split = ceil_div(Mrows, (int64_t)BK) * BK;
  for (int64_t s = 1; s <= max_s && p.tiles > 0; ++s) {
    const int64_t kps = ceil_div(ceil_div(Mrows, s), (int64_t)BK) * BK;
    const int64_t sp = ceil_div(Mrows, kps);
    const int64_t rounds = ceil_div((int64_t)p.tiles * sp, kNumCU);
    // rounds of kps-row items + fixed per-round prologue/epilogue + slab write/read traffic per split
    // (weights fitted on MI355X, tools/wgrad_probe.py)
    const double cost = (double)rounds * (kps + 32.0) + 8.0 * sp;
    if (cost < best) { best = cost; p.splits = (int)sp; p.k_per_split = kps; }
  }
  if (p.n_tiny > 0) wgrad_tiny_plan(Mrows, p.n_tiny, &p.tiny_splits, &p.tiny_kps);
  return p;
}
static bool group_is_tiny(const GroupPlan& p, int32_t N, int32_t K) { return p.n_tiny > 0 && group_kind(N, K) == 2; }
static size_t group_ws_layout(const GroupPlan& p, int n, const int32_t* N, const int32_t* K, size_t* slab_off, size_t* db_off) {
  size_t off = 0;
  for (int l = 0; l < n; ++l) {
    const size_t sp = group_is_tiny(p, N[l], K[l]) ? p.tiny_splits : p.splits;
    if (slab_off) slab_off[l] = off;
    off += align_up(sp * N[l] * K[l] * sizeof(float), 256);
    if (db_off) db_off[l] = off;
    off += align_up(sp * N[l] * sizeof(float), 256);
  }
  return off;
}


// ---- host-side launchers shared with wgrad_split.hip (wgrad_shared.h) ----------------------------------------------------
bool wgrad_tiny_shape(int32_t N, int32_t K) { return group_kind(N, K) == 2; }
void wgrad_tiny_plan(int64_t Mrows, int n_tiny, int* splits, int64_t* k_per_split) {
  // tiny-dimension layers: their own short launch, ~one workgroup per CU over all of them, 1..2 row tiles each
  int64_t ts = std::max<int64_t>(1, kNumCU / std::max(1, n_tiny));
  const int64_t cap = std::max<int64_t>(1, Mrows / (2 * TINY_ROWS));
  if (ts > cap) ts = cap;
  *k_per_split = ceil_div(ceil_div(Mrows, ts), (int64_t)TINY_ROWS) * TINY_ROWS;
  *splits = (int)ceil_div(Mrows, *k_per_split);
}
int launch_wgrad_tiny(const TinyArgs& T, int splits, hipStream_t st) {
  constexpr size_t tiny_lds = (size_t)(TINY_ROWS * TINY_MAX_LG + TINY_ROWS * TINY_MAX_S) * sizeof(float) + 1024;
  static bool once_t = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tiny_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tiny_lds), true);
  (void)once_t;
  // (forking this launch onto a side stream so that it runs UNDER the grouped launch instead of in front of it was measured:
  //  the grouped kernel slows down by more than the 12 us the fork hides, 1329 vs 1334 steps/s -- not kept)
  hipLaunchKernelGGL(wgrad_tiny_k, dim3((unsigned)splits, (unsigned)T.n), dim3(TINY_THREADS), tiny_lds, st, T);
  return launch_status("clica_mlp_wgrad(tiny)");
}
int launch_slab_reduce_group(const ReduceGroupArgs& R, int blocks, hipStream_t st) {
  const int front = (R.adam.p && R.adam.s16_state) ? s16::kS16UpdateBlocks : 0;      // the f16x2 scale update in front (ReduceAdam)
  hipLaunchKernelGGL(slab_reduce_group_k, dim3((unsigned)(blocks + front)), dim3(RED_THREADS), 0, st, R);
  return launch_status("clica_mlp_wgrad(reduce)");
}
int slab_reduce_entry(ReduceGroupArgs& R, int l, int first_block, int sp, const float* slab, const float* dbslab,
                      float* dW, int64_t lddw, float* db, int32_t N, int32_t K) {
  const bool v4 = (K % 4 == 0) && (lddw % 4 == 0) && aligned16(dW);
  R.vec4[l] = v4 ? 1 : 0; R.splits[l] = sp;
  R.dw_blocks[l] = (int)ceil_div(v4 ? (int64_t)N * K / 4 : (int64_t)N * K, reduce_flat(sp, v4) ? RED_THREADS : 64);
  R.first[l] = first_block;
  R.slab[l] = slab; R.dbslab[l] = dbslab; R.dW[l] = dW; R.db[l] = db;
  R.M[l] = N; R.N[l] = K; R.lddw[l] = lddw;
  return R.dw_blocks[l] + (db ? (int)ceil_div(N, 64) : 0);
}

}  // namespace gemm
}  // namespace clica

using namespace clica;
using namespace clica::gemm;

extern "C" int clica_mlp_wgrad_workspace_bytes(int64_t M, int32_t n_layers, const int32_t* N, const int32_t* K, size_t* bytes) {
  CLICA_CHECK_ARG(bytes && N && K && M > 0 && n_layers >= 1 && n_layers <= MAXG, "clica_mlp_wgrad_workspace_bytes: bad argument");
  for (int l = 0; l < n_layers; ++l) CLICA_CHECK_ARG(N[l] >= 1 && K[l] >= 1, "clica_mlp_wgrad_workspace_bytes: layer %d: bad size", l);
  *bytes = group_ws_layout(plan_wgrad_group(M, n_layers, N, K), n_layers, N, K, nullptr, nullptr);
  return CLICA_OK;
}

extern "C" int clica_mlp_wgrad(int64_t M, int32_t n_layers, const float* const* dZ, const int64_t* lddz,
                               const float* const* X, const int64_t* ldx, float* const* dW, const int64_t* lddw,
                               float* const* db, const int32_t* N, const int32_t* K, int32_t accumulate,
                               void* workspace, size_t workspace_bytes, clica_stream_t stream) {
  CLICA_CHECK_ARG(dZ && lddz && X && ldx && dW && lddw && db && N && K && workspace && M > 0, "clica_mlp_wgrad: NULL pointer / empty batch");
  CLICA_CHECK_ARG(n_layers >= 1 && n_layers <= MAXG, "clica_mlp_wgrad: %d layers (1..%d supported)", n_layers, MAXG);
  for (int l = 0; l < n_layers; ++l) {
    CLICA_CHECK_ARG(dZ[l] && X[l] && dW[l] && N[l] >= 1 && K[l] >= 1, "clica_mlp_wgrad: layer %d: bad argument", l);
    CLICA_CHECK_ARG(lddz[l] >= N[l] && ldx[l] >= K[l] && lddw[l] >= K[l], "clica_mlp_wgrad: layer %d: leading dimension too small", l);
  }
  const GroupPlan p = plan_wgrad_group(M, n_layers, N, K);
  size_t slab_off[MAXG], db_off[MAXG];
  const size_t need = group_ws_layout(p, n_layers, N, K, slab_off, db_off);
  if (need > workspace_bytes) { set_error("clica_mlp_wgrad: workspace %zu < %zu", workspace_bytes, need); return CLICA_E_WORKSPACE; }
  hipStream_t st = as_stream(stream);
  GroupArgs G{};
  ReduceGroupArgs R{};
  TinyArgs T{};
  R.n = n_layers; R.accumulate = accumulate ? 1 : 0;
  int item = 0, rblock = 0, ng = 0;
  for (int l = 0; l < n_layers; ++l) {
    const bool tiny = group_is_tiny(p, N[l], K[l]);
    const int sp = tiny ? p.tiny_splits : p.splits;
    float* slab = (float*)((char*)workspace + slab_off[l]);
    float* dbslab = (float*)((char*)workspace + db_off[l]);
    // dW[N,K] = dZ[M,N]^T X[M,K]: "M" = N, "N" = K, contraction over the batch rows
    Args g{};
    g.A = dZ[l]; g.lda = lddz[l]; g.B = X[l]; g.ldb = ldx[l]; g.C = slab; g.ldc = K[l]; g.M = N[l]; g.N = K[l]; g.Kc = M;
    g.k_per_split = tiny ? p.tiny_kps : p.k_per_split; g.dbias_slab = db[l] ? dbslab : nullptr;
    if (tiny) {
      T.p[T.n++] = g;
    } else {
      G.p[ng] = g;
      int kind = group_kind(N[l], K[l]);
      int bm, bn; group_tile(kind, &bm, &bn);              // the tile shape follows the SHAPE (it fixed the plan); a problem whose
      if (!(aligned16(g.A) && aligned16(g.B) && g.lda % 4 == 0 && g.ldb % 4 == 0)) kind = (bm == 128 && bn == 128) ? 0 : -1;
      CLICA_CHECK_ARG(kind >= 0, "clica_mlp_wgrad: layer %d: operands of a %d x %d layer must be 16-byte aligned with leading "
                      "dimensions that are multiples of 4", l, N[l], K[l]);   // pointers are unaligned can only take the 128 x 128 register-staged body
      G.gx[ng] = (int)ceil_div(K[l], bn); G.gy[ng] = (int)ceil_div(N[l], bm);
      G.vec[ng] = kind;
      G.first[ng] = item; item += G.gx[ng] * G.gy[ng] * sp;
      ++ng;
    }
    rblock += slab_reduce_entry(R, l, rblock, sp, slab, dbslab, dW[l], lddw[l], db[l], N[l], K[l]);
  }
  G.n = ng; G.first[ng] = G.total = item; R.first[n_layers] = rblock;
  if (T.n > 0) {
    int rct = launch_wgrad_tiny(T, p.tiny_splits, st);
    if (rct) return rct;
  }
  if (ng > 0) {
    constexpr int WM = 2, WN = 4, STAGES = 3, THREADS = 64 * WM * WN;
    constexpr size_t lds1 = DMA_STAGES * (Tile<GBM, false, THREADS>::LDS_FLOATS + Tile<GBN, false, THREADS>::LDS_FLOATS) * sizeof(float);
    constexpr size_t lds = lds1 > kBody2LdsBytes ? lds1 : kBody2LdsBytes;
    auto k = wgrad_group_k<GBM, GBN, WM, WN, STAGES>;
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    hipLaunchKernelGGL(k, dim3((unsigned)item), dim3(THREADS), lds, st, G);
    int rc = launch_status("clica_mlp_wgrad");
    if (rc) return rc;
  }
  return launch_slab_reduce_group(R, rblock, st);
}

extern "C" int clica_linear_fwd(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias,
                                float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                                int32_t leaky, float slope, clica_stream_t stream) {
  CLICA_CHECK_ARG(X && W && Y, "clica_linear_fwd: NULL pointer");
  CLICA_CHECK_ARG(M > 0 && N > 0 && K > 0, "clica_linear_fwd: M=%lld N=%lld K=%lld must be positive", (long long)M, (long long)N, (long long)K);
  CLICA_CHECK_ARG(ldx >= K && ldw >= K && ldy >= N, "clica_linear_fwd: leading dimension too small");
  if (use_skinny() && skinny_fwd(X, ldx, W, ldw, bias, Y, ldy, M, N, K, leaky, slope, as_stream(stream)))
    return launch_status("clica_linear_fwd(skinny)");
  Args g{}; g.A = X; g.lda = ldx; g.B = W; g.ldb = ldw; g.C = Y; g.ldc = ldy; g.M = M; g.N = N; g.Kc = K;
  g.bias = bias; g.slope = slope; g.leaky = leaky;
  const int cfg = cfg_fwd(N);
  return launch<true, true, EPI_BIAS_ACT>(cfg, g, 1, as_stream(stream), "clica_linear_fwd");
}

extern "C" int clica_linear_dgrad(const float* dY, int64_t lddy, const float* W, int64_t ldw,
                                  const float* Xact, int64_t ldxa, float slope,
                                  float* dX, int64_t lddx, int64_t M, int64_t N, int64_t K,
                                  clica_stream_t stream) {
  CLICA_CHECK_ARG(dY && W && dX, "clica_linear_dgrad: NULL pointer");
  CLICA_CHECK_ARG(M > 0 && N > 0 && K > 0, "clica_linear_dgrad: sizes must be positive");
  CLICA_CHECK_ARG(lddy >= N && ldw >= K && lddx >= K && (!Xact || ldxa >= K), "clica_linear_dgrad: leading dimension too small");
  if (use_skinny() && skinny_dgrad(dY, lddy, W, ldw, Xact, ldxa, slope, dX, lddx, M, N, K, as_stream(stream)))
    return launch_status("clica_linear_dgrad(skinny)");
  // dX[M,K] = dY[M,N] W[N,K]: contraction over N; B_op[kc=n][j=k] = W[n][k] (Kc strided)
  Args g{}; g.A = dY; g.lda = lddy; g.B = W; g.ldb = ldw; g.C = dX; g.ldc = lddx; g.M = M; g.N = K; g.Kc = N;
  g.xact = Xact; g.ldxa = ldxa; g.slope = slope;
  const int cfg = cfg_dgrad();
  return launch<true, false, EPI_DACT>(cfg, g, 1, as_stream(stream), "clica_linear_dgrad");
}

extern "C" int clica_linear_plan(int32_t op, int64_t M, int64_t N, int64_t K, int32_t* tile_m, int32_t* tile_n,
                                 int32_t* waves, int32_t* splits) {
  CLICA_CHECK_ARG(op >= 0 && op <= 2 && M > 0 && N > 0 && K > 0, "clica_linear_plan: bad argument");
  int cfg, sp = 1;
  if (op == 0) cfg = cfg_fwd(N);
  else if (op == 1) cfg = cfg_dgrad();
  else { const WgradPlan p = plan_wgrad(N, K, M); cfg = p.cfg; sp = p.splits; }
  if (tile_m) *tile_m = kCfgs[cfg].bm;
  if (tile_n) *tile_n = kCfgs[cfg].bn;
  if (waves) *waves = kCfgs[cfg].wm * kCfgs[cfg].wn;
  if (splits) *splits = sp;
  return CLICA_OK;
}

namespace clica { namespace lp { void set_dot_mfma(int on); void set_fused_finalize(int on); } }      // lp_loss.hip: SimCLRLoss contraction path; training forward in one launch
namespace clica { namespace wsplit { void set_epi_specialised(int on); } }      // wgrad_split.hip: gemm_split_k<1, code> vs the generic epilogue
extern "C" int clica_set_tuning(const char* key, int32_t value) {
  CLICA_CHECK_ARG(key != nullptr, "clica_set_tuning: key is NULL");
  gemm::Tuning& t = gemm::tuning();
  if (!strcmp(key, "skinny")) t.skinny = value != 0;
  else if (!strcmp(key, "gemm_cfg_fwd")) t.cfg_fwd = value;
  else if (!strcmp(key, "gemm_cfg_dgrad")) t.cfg_dgrad = value;
  else if (!strcmp(key, "gemm_cfg_wgrad")) t.cfg_wgrad = value;
  else if (!strcmp(key, "dot_mfma")) clica::lp::set_dot_mfma(value);
  else if (!strcmp(key, "gemm16_epilogue")) clica::wsplit::set_epi_specialised(value);
  else if (!strcmp(key, "lp_fused_finalize")) clica::lp::set_fused_finalize(value);
  else if (!strcmp(key, "reset")) { t = gemm::Tuning{-1, -1, -1, true}; clica::lp::set_dot_mfma(1); clica::wsplit::set_epi_specialised(1); clica::lp::set_fused_finalize(1); }
  else { set_error("clica_set_tuning: unknown key '%s'", key); return CLICA_E_INVALID; }
  return CLICA_OK;
}

extern "C" int clica_linear_wgrad_workspace_bytes(int64_t M, int64_t N, int64_t K, size_t* bytes) {
  CLICA_CHECK_ARG(bytes && M > 0 && N > 0 && K > 0, "clica_linear_wgrad_workspace_bytes: bad argument");
  // worst case over the configurations the planner (or the tuning hook) may pick
  const int64_t max_s = ceil_div(M, (int64_t)BK * 4) < kNumCU ? ceil_div(M, (int64_t)BK * 4) : kNumCU;
  *bytes = align_up((size_t)max_s * N * K * sizeof(float), 256) + align_up((size_t)max_s * N * sizeof(float), 256);
  const WgradPlan p = plan_wgrad(N, K, M);
  const size_t exact = align_up((size_t)p.splits * N * K * sizeof(float), 256) + align_up((size_t)p.splits * N * sizeof(float), 256);
  if (tuning().cfg_wgrad < 0) *bytes = exact;
  return CLICA_OK;
}

extern "C" int clica_linear_wgrad(const float* dY, int64_t lddy, const float* X, int64_t ldx,
                                  float* dW, int64_t lddw, float* db, int64_t M, int64_t N, int64_t K,
                                  int32_t accumulate, void* workspace, size_t workspace_bytes,
                                  clica_stream_t stream) {
  CLICA_CHECK_ARG(dY && X && dW && workspace, "clica_linear_wgrad: NULL pointer");
  CLICA_CHECK_ARG(M > 0 && N > 0 && K > 0, "clica_linear_wgrad: sizes must be positive");
  CLICA_CHECK_ARG(lddy >= N && ldx >= K && lddw >= K, "clica_linear_wgrad: leading dimension too small");
  hipStream_t st = as_stream(stream);
  const WgradPlan p = plan_wgrad(N, K, M);
  const size_t slab_bytes = align_up((size_t)p.splits * N * K * sizeof(float), 256);
  const size_t need = slab_bytes + align_up((size_t)p.splits * N * sizeof(float), 256);
  if (need > workspace_bytes) { set_error("clica_linear_wgrad: workspace %zu < %zu", workspace_bytes, need); return CLICA_E_WORKSPACE; }
  float* slab = (float*)workspace;
  float* dbslab = (float*)((char*)workspace + slab_bytes);
  // dW[N,K] = dY[M,N]^T X[M,K]: "M" = N, "N" = K, contraction over the batch rows M
  Args g{}; g.A = dY; g.lda = lddy; g.B = X; g.ldb = ldx; g.C = slab; g.ldc = K; g.M = N; g.N = K; g.Kc = M;
  g.k_per_split = p.k_per_split;
  g.dbias_slab = db ? dbslab : nullptr;
  int rc;
  const bool vec = aligned16(g.A) && aligned16(g.B) && g.lda % 4 == 0 && g.ldb % 4 == 0 && N % 4 == 0 && K % 4 == 0;
  if (vec && p.cfg == 1) {
    // 128x128 8-wave tiles: the direct global -> LDS body of the grouped kernel, as a one-problem group
    GroupArgs G{};
    G.n = 1; G.p[0] = g; G.vec[0] = 1;
    G.gx[0] = (int)ceil_div(K, GBN); G.gy[0] = (int)ceil_div(N, GBM);
    G.first[0] = 0; G.first[1] = G.total = G.gx[0] * G.gy[0] * p.splits;
    constexpr int WM = 2, WN = 4, STAGES = 3, THREADS = 64 * WM * WN;
    constexpr size_t lds = DMA_STAGES * (Tile<GBM, false, THREADS>::LDS_FLOATS + Tile<GBN, false, THREADS>::LDS_FLOATS) * sizeof(float);
    constexpr size_t lds_max = lds > kBody2LdsBytes ? lds : kBody2LdsBytes;      // one attribute value for every call site of this kernel
    auto k = wgrad_group_k<GBM, GBN, WM, WN, STAGES>;
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max), true);
    (void)once;
    hipLaunchKernelGGL(k, dim3((unsigned)G.total), dim3(THREADS), lds, st, G);
    rc = launch_status("clica_linear_wgrad");
  } else {
    rc = launch<false, false, EPI_SLAB>(p.cfg, g, p.splits, st, "clica_linear_wgrad");
  }
  if (rc) return rc;
  launch_slab_reduce(slab, dbslab, p.splits, N, K, dW, lddw, db, accumulate ? 1 : 0, st);
  return launch_status("clica_linear_wgrad(reduce)");
}

// =====================================================================================================================
// Convolutions of the KITTI-masks encoder (BetaVAE_H, /root/reference/kitti_masks/model.py:41-56): four Conv2d(k = 4,
// stride 2, pad 1) + ReLU stages (the fifth, k = 4 on a 4 x 4 map, is a plain Linear over the flattened map) as
// IMPLICIT GEMMs on the fp32-MFMA template above.  Layout: channels-last.  The input of a stage is kept as the padded
// "space-to-depth" tensor  S[image][sy][sx][(py, px, c)] = in[image][2 sy + py - 1][2 sx + px - 1][c]  (hs = H/2 + 1 by
// ws = W/2 + 1 pixels of 4C channels, zero border), in which the 4 x 4 stride-2 window of output pixel (oy, ox) is the 2 x 2
// stride-1 window (oy + dy, ox + dx): two runs of 8C contiguous floats, the second one ws * 4C floats behind the first.
// So with GEMM rows numbered over the WHOLE hs x ws grid (r = (image * hs + y) * ws + x; rows with y = hs - 1 or
// x = ws - 1 are not outputs and are computed for nothing: (ws/(ws-1))^2 of the useful work),
//   forward   out[r][co] = sum_k A[r][k] Wg[co][k],      A[r][k] = S_flat[r * 4C + k + (k >= 8C ? (ws - 2) * 4C : 0)]
//   wgrad     dWg[co][k] = sum_r dO[r][co] A[r][k]       (dO = 0 on the non-output rows), db = column sums of dO
//   dgrad     dS[r][j]   = sum_{dy,dx,co} dO[r - dy ws - dx][co] Wg[co][(dy, dx, j)]   -- the same two-run operand on dO
// and nothing is ever materialised: the loaders jump between the runs (Tile::load's seg / jump), the forward epilogue adds
// bias + ReLU and scatters a row straight into the NEXT stage's space-to-depth tensor, the dgrad epilogue applies the ReLU
// gate (S itself is the saved activation) and scatters into the previous stage's dO grid.  Borders and non-output rows are
// never written, so buffers zeroed once stay valid (the host keeps them: cl_ica_amd/conv.py).
// The first stage (C = 1 or 3 input channels, K = 16C) runs on an explicit patch matrix (clica_conv_im2col_k4s2, 64 B per
// output pixel for the masks) -- its A operand has no run long enough to tile.
namespace clica {
namespace gemm {

template <int BM, int BN, int WM, int WN, int STAGES, bool A_CONTIG, bool B_CONTIG, int EPI>
__global__ __launch_bounds__(64 * WM * WN) void conv_gemm_k(Args g, ConvX cx) {
  // XCD-contiguous numbering over ALL three grid dimensions (N fastest, then M, then the contraction split): the column tiles of one split
  // of the weight gradient read the same rows of S shifted by 0 / 1 / ws / ws + 1 pixels -- as neighbours on one XCD they share its L2
  // (numbered per z-slice they sat on four different XCDs: PMC L2 hit 0.03, 758 MB fetched for 379 MB of operands)
  const int gx = gridDim.x, gxy = gx * gridDim.y;
  const int id = xcd_contiguous(blockIdx.z * gxy + blockIdx.y * gx + blockIdx.x, gxy * gridDim.z);
  const int bz = id / gxy, rem = id - bz * gxy, by = rem / gx;
  gemm_body<BM, BN, WM, WN, STAGES, A_CONTIG, B_CONTIG, EPI, true, true>(g, rem - by * gx, by, bz, cx);
}

template <int BM, int BN, int WM, int WN, int STAGES, bool A_CONTIG, bool B_CONTIG, int EPI>
static int launch_conv(const Args& g, const ConvX& cx, int splits, hipStream_t st, const char* who) {
  constexpr int THREADS = 64 * WM * WN;
  dim3 grid((unsigned)ceil_div(g.N, BN), (unsigned)ceil_div(g.M, BM), (unsigned)splits), block(THREADS);
  constexpr size_t lds = STAGES * (Tile<BM, A_CONTIG, THREADS>::LDS_FLOATS + Tile<BN, B_CONTIG, THREADS>::LDS_FLOATS) * sizeof(float);
  auto k = conv_gemm_k<BM, BN, WM, WN, STAGES, A_CONTIG, B_CONTIG, EPI>;
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
  (void)once;
  hipLaunchKernelGGL(k, grid, block, lds, st, g, cx);
  return launch_status(who);
}

// forward of one stage from either operand form
static int conv_fwd_launch(const float* A, int64_t lda, int64_t seg, int64_t jump, const float* Wg, const float* bias, int64_t rows,
                           int32_t K, int32_t Cout, int32_t hs, int32_t ws, int32_t ho, int32_t wo, int32_t relu, int32_t scatter,
                           float* out, uint32_t* gate_bits, hipStream_t st, const char* who, int compact = 0) {
  const int ghs = hs, gws = ws;
  const bool fast_ok = K % BK == 0 && lda % 4 == 0 && jump % 4 == 0 && seg % 4 == 0 && jump < (1 << 30);
  if (compact && fast_ok && scatter) { rows = rows / ((int64_t)hs * ws) * ((int64_t)ho * wo); hs = ho; ws = wo; } else { compact = 0; }
  if (scatter == 2 && !compact) { set_error("%s: scatter = 2 needs a contraction of whole k-tiles", who); return CLICA_E_INVALID; }
  Args g{}; g.A = A; g.lda = lda; g.B = Wg; g.ldb = K; g.C = out; g.ldc = Cout; g.M = rows; g.N = Cout; g.Kc = K;
  g.bias = bias; g.leaky = relu ? 1 : 0; g.slope = 0.f;
  ConvX cx{}; cx.a_seg = seg; cx.a_jump = jump; cx.mode = scatter == 2 ? 3 : (scatter ? 1 : 0);
  cx.hs = hs; cx.ws = ws; cx.ho = ho; cx.wo = wo; cx.dhs = ho / 2 + 1; cx.dws = wo / 2 + 1; cx.c = Cout;
  cx.inv_pix = 1.f / (float)(hs * ws); cx.inv_ws = 1.f / (float)ws;
  cx.gate_out = gate_bits;
  cx.fast = fast_ok ? 1 : 0;
  cx.compact = compact; cx.ghs = ghs; cx.gws = gws;
  if (gate_bits && !(scatter == 1 && Cout % 32 == 0)) { set_error("%s: gate bits need the scattering epilogue and Cout %% 32 == 0", who); return CLICA_E_INVALID; }
  if (scatter && rows >= (1 << 24)) { set_error("%s: %lld rows (the scattering epilogue handles < 2^24)", who, (long long)rows); return CLICA_E_INVALID; }
  // few rows (the k = 4 stage on the 4 x 4 map: images x 1600 -> 256): small tiles so that the launch still covers the chip
  if (Cout > 64 && rows <= 8192) return launch_conv<64, 64, 2, 2, 2, true, true, EPI_BIAS_ACT>(g, cx, 1, st, who);
  // Cout = 32 measured on the 17 x 17 stage: 128 x 32 / 4 waves 286 us; 64 x 32 / 2 waves 312; 256 x 32 / 4 waves 392; 64 x 32 with 3 stages 392
  if (Cout <= 32) return launch_conv<128, 32, 4, 1, 2, true, true, EPI_BIAS_ACT>(g, cx, 1, st, who);   // 3 workgroups x 4 waves per CU (LDS-limited)
  if (Cout <= 64) return launch_conv<128, 64, 2, 2, 2, true, true, EPI_BIAS_ACT>(g, cx, 1, st, who);
  return launch_conv<64, 128, 2, 2, 2, true, true, EPI_BIAS_ACT>(g, cx, 1, st, who);
}

// Data gradient of the widest stage (C = Cout = 32: a [rows x 128] x [128 x 128] product, 43 % of the stack's data-gradient work) as a
// PERSISTENT kernel.  With a 128-deep contraction the tiled GEMM above is all prologue and epilogue (PMC: matrix pipes busy 0.34): every
// 64-row workgroup re-reads the whole 64 KB operand matrix Wd and pays its load / barrier / scatter latencies for 4 k-tiles of work.  Here a
// workgroup (8 waves, two per CU) keeps Wd in LDS for its whole life and walks over 256-row tiles.  A wave reads its A fragments -- row r,
// contraction index (1 - dy, 1 - dx, co) = dO row r - ws - 1 + (1 - dy) ws + (1 - dx) -- straight from global memory two k-steps ahead (16
// bytes per lane and step out of the same 19 KB of dO per 256 rows: L1 / L2 hits); there is no barrier after the first one, so waves drift
// apart and one wave's scattering epilogue runs under the others' matrix work.  (A version that also staged the dO block in LDS, double-
// buffered with one barrier per tile, measured 358-370 us against this one's 347; the tiled GEMM 427.)
constexpr int DG_ROWS = 256;
__global__ __launch_bounds__(512) void conv_dgrad32_stream_k(const float* __restrict__ A0 /* dO - (ws + 1) * 32 */, const float* __restrict__ Wd,
                                                             int64_t rows, float* __restrict__ dst, ConvX cx) {
  extern __shared__ __attribute__((aligned(16))) float sB[];   // [128][128]
  const int ws = cx.ws;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, l31 = lane & 31;
  for (int u = threadIdx.x; u < 128 * 32; u += 512) reinterpret_cast<float4*>(sB)[u] = reinterpret_cast<const float4*>(Wd)[u];
  __syncthreads();
  const int64_t ntiles = (rows + DG_ROWS - 1) / DG_ROWS;
  const int64_t last = rows + ws;                            // last readable row of A0
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * DG_ROWS;
    const int64_t rbase = r0 + wave * 32 + l31;
    auto lda = [&](int st) {
      const int64_t gr = min(rbase + (st >> 3) * ws + ((st & 7) >> 2), last);
      return *reinterpret_cast<const float4*>(A0 + gr * 32 + 8 * (st & 3) + 4 * h);
    };
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float4 aq[3];
    aq[0] = lda(0); aq[1] = lda(1);
    float b4[2][4][4];
    auto bfr = [&](int buf, int st) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) b4[buf][j][t] = sB[(8 * st + 4 * h + t) * 128 + j * 32 + l31];
    };
    bfr(0, 0);
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      if (st + 2 < 16) aq[(st + 2) % 3] = lda(st + 2);
      if (st + 1 < 16) bfr((st + 1) & 1, st + 1);
      __builtin_amdgcn_sched_barrier(0);
      const float4 av = aq[st % 3];
      const float a4[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t], b4[st & 1][j][t], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = r0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row >= rows) continue;
      int img, y, x;
      conv_row_to_pixel(cx, row, img, y, x);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int yy = 2 * y + (j >> 1) - 1, xx = 2 * x + (j & 1) - 1;
        if (yy < 0 || xx < 0 || yy >= cx.dho || xx >= cx.dwo) continue;
        const int64_t pix = ((int64_t)img * cx.dhs + yy) * cx.dws + xx;
        const unsigned word = cx.gate_in[pix];
        dst[pix * 32 + l31] = ((word >> l31) & 1u) ? acc[j][r] : 0.f;
      }
    }
  }
}

struct ConvWgradPlan { int bm, splits; int64_t k_per_split; };
static ConvWgradPlan plan_conv_wgrad(int64_t rows, int32_t Cout, int32_t K) {
  ConvWgradPlan p;
  p.bm = Cout <= 32 ? 32 : 64;
  const int64_t tiles = ceil_div(Cout, p.bm) * ceil_div(K, 128);
  // ~three small workgroups per CU.  (Capping the contraction chains at ~1024 rows -- 578 splits on the widest stage -- was tried when that
  // stage's dW read 1.2e-5 against fp64: the figure did not move, it was a ReLU gate tie, see tests/test_gpu_conv.py; the extra slabs only
  // cost the reduction 25 us.)
  int64_t want = std::max<int64_t>(1, 3 * kNumCU / tiles);
  want = std::min<int64_t>(want, std::max<int64_t>(1, rows / (8 * BK)));
  p.k_per_split = ceil_div(ceil_div(rows, want), (int64_t)BK) * BK;
  p.splits = (int)ceil_div(rows, p.k_per_split);
  return p;
}

// x [images][C][H][W] (NCHW, as the data loader hands it over) -> patches [images * H/2 * W/2][16 C], column (ky * 4 + kx) * C + c.
// One thread per (output pixel, tap): consecutive lanes write consecutive floats (C = 1) and read consecutive input columns.
__global__ __launch_bounds__(256) void conv_im2col_k4s2_k(const float* __restrict__ x, unsigned pixels, int C, int H, int W,
                                                           float* __restrict__ patches) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  const unsigned pixel = t >> 4, tap = t & 15u;
  if (pixel >= pixels) return;
  const unsigned Wo = W / 2, Ho = H / 2;
  const unsigned prow = pixel / Wo, ox = pixel - prow * Wo, img = prow / Ho, oy = prow - img * Ho;
  const int y = 2 * (int)oy + (int)(tap >> 2) - 1, xx = 2 * (int)ox + (int)(tap & 3u) - 1;
  const bool in = y >= 0 && y < H && xx >= 0 && xx < W;
  const float* src = x + ((int64_t)img * C * H + y) * W + xx;
  float* dst = patches + ((int64_t)pixel * 16 + tap) * C;
  for (int c = 0; c < C; ++c) dst[c] = in ? src[(int64_t)c * H * W] : 0.f;
}

// One input channel (the masks): a thread owns one window ROW of one output pixel -- four taps, one 16-byte store (the generic kernel's
// 4-byte stores were instruction-bound: 69 us for 167 MB).
__global__ __launch_bounds__(256) void conv_im2col_k4s2_c1_k(const float* __restrict__ x, unsigned pixels, int H, int W, float* __restrict__ patches) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  const unsigned pixel = t >> 2, ky = t & 3u;
  if (pixel >= pixels) return;
  const unsigned Wo = W / 2, Ho = H / 2;
  const unsigned prow = pixel / Wo, ox = pixel - prow * Wo, img = prow / Ho, oy = prow - img * Ho;
  const int y = 2 * (int)oy + (int)ky - 1, x0 = 2 * (int)ox - 1;
  const bool rin = y >= 0 && y < H;
  const float* rowp = x + ((int64_t)img * H + (rin ? y : 0)) * W;
  const bool i0 = rin && x0 >= 0, i3 = rin && x0 + 3 < W;
  const float v0 = rowp[i0 ? x0 : 0], v1 = rowp[x0 + 1], v2 = rowp[x0 + 2], v3 = rowp[i3 ? x0 + 3 : 0];
  *reinterpret_cast<float4*>(patches + (int64_t)pixel * 16 + 4 * ky) = make_float4(i0 ? v0 : 0.f, rin ? v1 : 0.f, rin ? v2 : 0.f, i3 ? v3 : 0.f);
}

// First stage's weight / bias gradient from the patch matrix: dWg[Cout][K] = dO[rows][Cout]^T P[rows][K] with Cout x K small (32 x 16 for the
// masks) and rows in the millions -- HBM-bound (dO 128 B + P 64 B per row).  A thread owns a 4 x 4 block of dWg; the (Cout/4)(K/4)
// threads of a group read one row's dO and P as float4s (the same 128 / 64 bytes for the whole group: one cache line each), the
// G = 256 / group size groups of a workgroup take rows r = g (mod G) of the workgroup's row range, are summed through LDS, and every
// workgroup writes ONE slab for the shared deterministic slab reduction.
__global__ __launch_bounds__(256) void conv_wgrad_patches_k(const float* __restrict__ dO, const float* __restrict__ P, int64_t rows,
                                                             int Cout, int K, int64_t rows_per_block, float* __restrict__ slab,
                                                             float* __restrict__ dbslab) {
  extern __shared__ float red[];                       // [G][Cout * K + Cout]
  const int ncg = Cout / 4, gsz = ncg * (K / 4), G = 256 / gsz;
  const int g = threadIdx.x / gsz, u = threadIdx.x - g * gsz, cg = u % ncg, kg = u / ncg;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float acc[4][4] = {};
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g < G) {
#pragma unroll 4
    for (int64_t r = r0 + g; r < r1; r += G) {
      const float4 d = *reinterpret_cast<const float4*>(dO + r * Cout + 4 * cg);
      const float4 p = *reinterpret_cast<const float4*>(P + r * K + 4 * kg);
      const float dv[4] = {d.x, d.y, d.z, d.w}, pv[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(dv[i], pv[j], acc[i][j]);
      bsum.x += d.x; bsum.y += d.y; bsum.z += d.z; bsum.w += d.w;
    }
    float* mine = red + (size_t)g * (Cout * K + Cout);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) mine[(4 * cg + i) * K + 4 * kg + j] = acc[i][j];
    if (kg == 0) { mine[Cout * K + 4 * cg] = bsum.x; mine[Cout * K + 4 * cg + 1] = bsum.y; mine[Cout * K + 4 * cg + 2] = bsum.z; mine[Cout * K + 4 * cg + 3] = bsum.w; }
  }
  __syncthreads();
  const int per = Cout * K + Cout;
  for (int e = threadIdx.x; e < per; e += 256) {
    float v = 0.f;
    for (int gg = 0; gg < G; ++gg) v += red[(size_t)gg * per + e];
    if (e < Cout * K) slab[(int64_t)blockIdx.x * Cout * K + e] = v;
    else if (dbslab) dbslab[(int64_t)blockIdx.x * Cout + (e - Cout * K)] = v;
  }
}

// First stage's forward from the patch matrix on the vector ALUs: K = 16 C is far too short a contraction for a 32-deep MFMA k-tile
// (the GEMM path spent 220 us per 2 M pixels on tile prologues and epilogues), and the stage is HBM-bound anyway: 64 B of patch in,
// 128 B of activation out per pixel.  A thread owns 8 output channels (its K x 8 weights live in registers) and walks over pixels;
// the four threads of a pixel write its 32 channels as one 128-byte piece of the next stage's space-to-depth tensor.
template <int K>
__global__ __launch_bounds__(256) void conv_fwd_patches_valu_k(const float* __restrict__ P, const float* __restrict__ Wg,
                                                                const float* __restrict__ bias, unsigned pixels, int Cout, int ho, int wo,
                                                                int relu, float* __restrict__ out, unsigned* __restrict__ gate_out,
                                                                unsigned* __restrict__ amax_slots) {
  const int groups = Cout / 8;                          // threads per pixel
  float amax = 0.f;                                     // conv16.hip: the maximum of what this thread stored (its consumer's f16 scale)
  const int cg = threadIdx.x % groups;
  // channel of this thread's c-th output: two runs of four, [4 cg, 4 cg + 4) and [Cout/2 + 4 cg, ...), so that each of the thread's two
  // 16-byte stores is contiguous with its neighbours' (a pixel's group writes 64 contiguous bytes per store instruction, not 16 of every 32)
  const int half = Cout / 2;
  auto chan = [&](int c) { return c < 4 ? 4 * cg + c : half + 4 * cg + (c - 4); };
  float w[8][K], b[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    b[c] = bias ? bias[chan(c)] : 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) w[c][k] = Wg[chan(c) * K + k];
  }
  const unsigned per_block = 256 / groups, stride = gridDim.x * per_block;
  const int dhs = ho / 2 + 1, dws = wo / 2 + 1;
  // the next pixel's patch is requested before this pixel's arithmetic (load -> use was an exposed round trip per pixel: PMC had the waves
  // waiting 61 % of their cycles at three waves per SIMD)
  float4 nxt[K / 4];
  const unsigned first = blockIdx.x * per_block + threadIdx.x / groups;
  if (first < pixels) {
#pragma unroll
    for (int k = 0; k < K; k += 4) nxt[k / 4] = *reinterpret_cast<const float4*>(P + (int64_t)first * K + k);
  }
  for (unsigned pixel = first; pixel < pixels; pixel += stride) {
    float a[K];
#pragma unroll
    for (int k = 0; k < K; k += 4) { a[k] = nxt[k / 4].x; a[k + 1] = nxt[k / 4].y; a[k + 2] = nxt[k / 4].z; a[k + 3] = nxt[k / 4].w; }
    if (pixel + stride < pixels) {
#pragma unroll
      for (int k = 0; k < K; k += 4) nxt[k / 4] = *reinterpret_cast<const float4*>(P + (int64_t)(pixel + stride) * K + k);
    }
    float o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) acc = fmaf(a[k], w[c][k], acc);       // k ascending, like the MFMA chain of the GEMM path
      acc += b[c];
      o[c] = (relu && !(acc > 0.f)) ? 0.f : acc;
      amax = fmaxf(amax, fabsf(o[c]));
    }
    const unsigned prow = pixel / (unsigned)wo, x = pixel - prow * (unsigned)wo, img = prow / (unsigned)ho, y = prow - img * (unsigned)ho;
    const unsigned Y = (y + 1) >> 1, X = (x + 1) >> 1, qq = ((y + 1) & 1) * 2 + ((x + 1) & 1);
    float* dst = out + (((int64_t)img * dhs + Y) * dws + X) * (4 * Cout) + qq * Cout + 4 * cg;
    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(dst + half) = make_float4(o[4], o[5], o[6], o[7]);
    if (gate_out) {               // Cout == 32: the four threads of a pixel combine their 8 bits into the pixel's gate word
      unsigned bits = 0;
#pragma unroll
      for (int c = 0; c < 8; ++c) bits |= (o[c] > 0.f ? 1u : 0u) << chan(c);
      bits |= __shfl_xor(bits, 1);
      bits |= __shfl_xor(bits, 2);
      if (cg == 0) gate_out[pixel] = bits;
    }
  }
  if (amax_slots) {                                     // one atomic per wave, 256 slots (conv16.hip: commit_amax)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    if ((threadIdx.x & 63) == 0) {
      amax = (amax <= 3.0e38f) ? amax : 3.4e38f;
      if (amax > 0.f) atomicMax(amax_slots + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 255), __float_as_uint(amax));
    }
  }
}

// The same two first-stage kernels for ONE input channel reading the IMAGE instead of a patch matrix (round 5): the 4 x 4 window of an
// output pixel is twelve aligned 8-byte loads out of 33 MB of masks that live in L1 / L2, where the patch matrix cost 134 MB written by
// the im2col launch and 134 MB read by each of the two kernels (42 + ~45 + ~40 us of a 1.7 ms step).  Window value (ky, kx) of output
// pixel (oy, ox) = x[img][2 oy - 1 + ky][2 ox - 1 + kx], zero outside the H x W image: columns 2 ox - 2 .. 2 ox + 3 as three float2.
// branch-free: out-of-image taps are read from a clamped address and zeroed by a select (as conditional loads the window cost twelve
// exec-mask branches per pixel: the kernel was instruction-bound at ~800 instructions per pixel and thread)
// the raw loads only (clamped addresses); win_mask() applies the zeroing when the values are USED -- a select on a value that has just
// been requested makes the compiler wait for it on the spot, which is what a prefetch must not do
__device__ __forceinline__ float4 win_row_raw(const float* __restrict__ ximg, int yy, int ox, int H, int W) {
  const int yc = min(max(yy, 0), H - 1);
  const float* rowp = ximg + (int64_t)yc * W + 2 * ox;
  const float l = rowp[ox > 0 ? -1 : 0];
  const float2 m = *reinterpret_cast<const float2*>(rowp);
  const float r = rowp[2 * ox + 2 < W ? 2 : 0];
  return make_float4(l, m.x, m.y, r);
}
__device__ __forceinline__ float4 win_mask(const float4 v, int yy, int ox, int H, int W) {
  const bool rin = yy >= 0 && yy < H, lin = ox > 0, rgt = 2 * ox + 2 < W;
  return make_float4((rin && lin) ? v.x : 0.f, rin ? v.y : 0.f, rin ? v.z : 0.f, (rin && rgt) ? v.w : 0.f);
}
__device__ __forceinline__ float4 win_row(const float* __restrict__ ximg, int yy, int ox, int H, int W) {
  const int yc = min(max(yy, 0), H - 1);
  const bool rin = yy == yc, lin = ox > 0, rgt = 2 * ox + 2 < W;
  const float* rowp = ximg + (int64_t)yc * W + 2 * ox;
  const float l = rowp[lin ? -1 : 0];
  const float2 m = *reinterpret_cast<const float2*>(rowp);
  const float r = rowp[rgt ? 2 : 0];
  return make_float4((rin && lin) ? l : 0.f, rin ? m.x : 0.f, rin ? m.y : 0.f, (rin && rgt) ? r : 0.f);
}
__device__ __forceinline__ void pixel_decode(unsigned pixel, int ho, int wo, int ho_shift, int wo_shift, unsigned& img, unsigned& oy, unsigned& ox) {
  if (wo_shift >= 0 && ho_shift >= 0) {
    ox = pixel & (unsigned)(wo - 1); const unsigned prow = pixel >> wo_shift; oy = prow & (unsigned)(ho - 1); img = prow >> ho_shift;
  } else {
    const unsigned prow = pixel / (unsigned)wo; ox = pixel - prow * (unsigned)wo; img = prow / (unsigned)ho; oy = prow - img * (unsigned)ho;
  }
}
__global__ __launch_bounds__(256) void conv_fwd_image_valu_k(const float* __restrict__ X, const float* __restrict__ Wg, const float* __restrict__ bias,
                                                              unsigned pixels, int Cout, int H, int W, int ho_shift, int wo_shift, int relu,
                                                              float* __restrict__ out, unsigned* __restrict__ gate_out, unsigned* __restrict__ amax_slots,
                                                              int ablate) {
  constexpr int K = 16;
  const int ho = H / 2, wo = W / 2;
  const int groups = Cout / 8;                          // threads per pixel
  const int cg = threadIdx.x % groups;
  const int half = Cout / 2;
  auto chan = [&](int c) { return c < 4 ? 4 * cg + c : half + 4 * cg + (c - 4); };      // as in conv_fwd_patches_valu_k
  float amax = 0.f;
  float w[8][K], b[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    b[c] = bias ? bias[chan(c)] : 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) w[c][k] = Wg[chan(c) * K + k];
  }
  const unsigned per_block = 256 / groups, stride = gridDim.x * per_block;
  const int dhs = ho / 2 + 1, dws = wo / 2 + 1;
  // The four threads of a pixel (a lane quad: cg = lane & 3) fetch ONE window row each and hand the values round with quad-permute
  // moves: a thread pays a quarter of the address / clamp / select work (each thread fetching the whole window: ~400 vector
  // instructions per pixel for 64 packed FMAs).  Two pixels in flight per thread.
  struct Pre { float4 v; unsigned img, y, x; };
  Pre P0, P1;
  __shared__ __attribute__((aligned(16))) float stage[64 * 36];      // [wave][pixel of the wave][32 channels + 4 pad]
  const int lane = threadIdx.x & 63, pl = lane >> 2;
  // every condition of the loop is uniform over the wave (a wave's 16 pixels start at `base`): the stores below cross lanes
  const unsigned base0 = blockIdx.x * per_block + (threadIdx.x >> 6) * 16;
  auto fetch = [&](Pre& P, unsigned base) {
    const unsigned pixel = min(base + (unsigned)pl, pixels - 1u);
    pixel_decode(pixel, ho, wo, ho_shift, wo_shift, P.img, P.y, P.x);
    P.v = win_row_raw(X + (int64_t)P.img * H * W, 2 * (int)P.y - 1 + cg, (int)P.x, H, W);
  };
  auto quad = [](float v, int src) -> float {       // value of lane (lane & ~3) + src
    const int i = __builtin_bit_cast(int, v);
    int r;
    switch (src) {
      case 0: r = __builtin_amdgcn_update_dpp(i, i, 0x00, 0xf, 0xf, false); break;
      case 1: r = __builtin_amdgcn_update_dpp(i, i, 0x55, 0xf, 0xf, false); break;
      case 2: r = __builtin_amdgcn_update_dpp(i, i, 0xaa, 0xf, 0xf, false); break;
      default: r = __builtin_amdgcn_update_dpp(i, i, 0xff, 0xf, 0xf, false); break;
    }
    return __builtin_bit_cast(float, r);
  };
  auto process = [&](Pre& P, unsigned base) {
    const unsigned pixel = base + (unsigned)pl;
    const bool valid = pixel < pixels;
    float a[K];
    const float4 pv = win_mask(P.v, 2 * (int)P.y - 1 + cg, (int)P.x, H, W);
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) { a[4 * ky] = quad(pv.x, ky); a[4 * ky + 1] = quad(pv.y, ky); a[4 * ky + 2] = quad(pv.z, ky); a[4 * ky + 3] = quad(pv.w, ky); }
    const unsigned img = P.img, y = P.y, x = P.x;
    if (base + 4 * stride < pixels) fetch(P, base + 4 * stride);
    float o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) acc = fmaf(a[k], w[c][k], acc);
      acc += b[c];
      o[c] = (relu && !(acc > 0.f)) ? 0.f : acc;
      amax = fmaxf(amax, fabsf(o[c]));
    }
    const unsigned Y = (y + 1) >> 1, Xs = (x + 1) >> 1, qq = ((y + 1) & 1) * 2 + ((x + 1) & 1);
    // The wave's 16 pixels x 128 B go through LDS so that a store instruction writes WHOLE 128-byte lines (eight lanes per pixel):
    // with each thread storing its two 16-byte runs directly, an instruction wrote half of every line it touched and the kernel sat at
    // 2.1 TB/s whatever its arithmetic looked like (three versions of the loads: 121 ... 128 us).
    const int myoff = valid ? (int)((((int64_t)img * dhs + Y) * dws + Xs) * (4 * Cout) + qq * Cout) : -1;
    float* mine = stage + (threadIdx.x >> 2) * 36;
    __builtin_amdgcn_wave_barrier();
    *reinterpret_cast<float4*>(mine + 4 * cg) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(mine + half + 4 * cg) = make_float4(o[4], o[5], o[6], o[7]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int p2 = hh * 8 + (lane >> 3);
      const float4 v = *reinterpret_cast<const float4*>(stage + ((threadIdx.x >> 6) * 16 + p2) * 36 + 4 * (lane & 7));
      const int off = __shfl(myoff, 4 * p2, 64);
      if (off >= 0 && !(ablate & 1)) *reinterpret_cast<float4*>(out + off + 4 * (lane & 7)) = v;
    }
    if (gate_out && valid && !(ablate & 2)) {
      unsigned bits = 0;
#pragma unroll
      for (int c = 0; c < 8; ++c) bits |= (o[c] > 0.f ? 1u : 0u) << chan(c);
      bits |= __shfl_xor(bits, 1);
      bits |= __shfl_xor(bits, 2);
      if (cg == 0) gate_out[pixel] = bits;
    }
  };
  // The 128 weights must have ARRIVED before the loop: left pending, their waits end up inside the loop body as s_waitcnt vmcnt(0) -- and
  // on this ISA a wait for loads with stores in flight is a wait for EVERYTHING anyway (loads and stores share one counter and complete
  // out of order with respect to each other, so the compiler emits vmcnt(0)): every pixel paid a full round trip for its stores to be
  // acknowledged, the kernel wrote its 268 MB at 2.1 TB/s whatever its loads looked like (five versions: 121 ... 157 us).  Hence:
  // weights pinned here, and FOUR pixels in flight per thread, refilled together -- one drain per four pixels.
#pragma unroll
  for (int c = 0; c < 8; ++c) {
#pragma unroll
    for (int k = 0; k < K; ++k) asm volatile("" ::"v"(w[c][k]));
    asm volatile("" ::"v"(b[c]));
  }
  Pre P2, P3;
  if (base0 < pixels) fetch(P0, base0);
  if (base0 + stride < pixels) fetch(P1, base0 + stride);
  if (base0 + 2 * stride < pixels) fetch(P2, base0 + 2 * stride);
  if (base0 + 3 * stride < pixels) fetch(P3, base0 + 3 * stride);
  for (unsigned base = base0; base < pixels; base += 4 * stride) {
    process(P0, base);
    if (base + stride < pixels) process(P1, base + stride);
    if (base + 2 * stride < pixels) process(P2, base + 2 * stride);
    if (base + 3 * stride < pixels) process(P3, base + 3 * stride);
  }
  if (amax_slots) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    if ((threadIdx.x & 63) == 0) {
      amax = (amax <= 3.0e38f) ? amax : 3.4e38f;
      if (amax > 0.f) atomicMax(amax_slots + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 255), __float_as_uint(amax));
    }
  }
}

// dWg[Cout][16] = dO[rows][Cout]^T window(rows), db = column sums of dO: conv_wgrad_patches_k with the window row read from the image and
// an 8 x 4 block of dWg per thread (eight channels x one window row: 32 FMAs per row for one row decode, one window row and two
// 16-byte dO loads; 4 x 4 blocks measured 117 us, 16 x 4 blocks -- fewer waves per SIMD -- 129, this shape 105).
// The (Cout/8) x 4 threads of a group share a row; the groups of a workgroup take rows r = g (mod G); a wave's groups are summed with
// lane shuffles, the four waves through LDS, every workgroup writes one slab.
__global__ __launch_bounds__(256) void conv_wgrad_image_k(const float* __restrict__ dO, const float* __restrict__ X, int64_t rows, int Cout, int H, int W,
                                                           int ho_shift, int wo_shift, int64_t rows_per_block, float* __restrict__ slab,
                                                           float* __restrict__ dbslab) {
  extern __shared__ float red[];                       // [4 waves][Cout * 16 + Cout]
  constexpr int K = 16;
  const int ho = H / 2, wo = W / 2;
  const int ncg = Cout / 8, gsz = ncg * 4, G = 256 / gsz;           // gsz = 16 for Cout = 32 (a power of two <= 64 is required)
  const int g = threadIdx.x / gsz, u = threadIdx.x - g * gsz, cg = u % ncg, kg = u / ncg;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float acc[8][4] = {};
  float bsum[8] = {};
#pragma unroll 4
  for (int64_t r = r0 + g; r < r1; r += G) {
    const float4 d0 = *reinterpret_cast<const float4*>(dO + r * Cout + 8 * cg);
    const float4 d1 = *reinterpret_cast<const float4*>(dO + r * Cout + 8 * cg + 4);
    unsigned img, oy, ox;
    pixel_decode((unsigned)r, ho, wo, ho_shift, wo_shift, img, oy, ox);
    const float4 wv = win_row(X + (int64_t)img * H * W, 2 * (int)oy - 1 + kg, (int)ox, H, W);
    const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w}, pv[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(dv[i], pv[j], acc[i][j]);
      bsum[i] += dv[i];
    }
  }
  // groups of one wave: lanes l, l + gsz, l + 2 gsz ... hold the same (cg, kg)
  for (int off = gsz; off < 64; off <<= 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] += __shfl_xor(acc[i][j], off, 64);
      bsum[i] += __shfl_xor(bsum[i], off, 64);
    }
  }
  const int per = Cout * K + Cout;
  if ((threadIdx.x & 63) < gsz) {
    float* mine = red + (size_t)(threadIdx.x >> 6) * per;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) mine[(8 * cg + i) * K + 4 * kg + j] = acc[i][j];
      if (kg == 0) mine[Cout * K + 8 * cg + i] = bsum[i];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < per; e += 256) {
    const float v = (red[e] + red[per + e]) + (red[2 * per + e] + red[3 * per + e]);
    if (e < Cout * K) slab[(int64_t)blockIdx.x * Cout * K + e] = v;
    else if (dbslab) dbslab[(int64_t)blockIdx.x * Cout + (e - Cout * K)] = v;
  }
}

// The same weight gradient for Cout = 32 on the fp32 matrix cores (round 5, last session).  The vector-ALU kernel above is
// instruction-bound like the forward was (16 threads x ~65 instructions per output pixel: ~63 us of issue for the 2048-mask batch, 102 us
// measured against ~60 us for its 268 MB of dO at the rate the other streaming kernels reach).  dWg[co][k] = sum_pixel dO[pixel][co] x
// window(pixel)[k] is a [32 x pixels] x [pixels x 16] product: v_mfma_f32_16x16x4_f32 contracts FOUR pixels per instruction -- exact fp32
// products, fp32 accumulation: no split, no scales -- and both operands are one register per lane straight from memory:
//   A[i][kq]: lane (i = l & 15, kq = l >> 4) loads the float2 dO[pixel + kq][2 i, 2 i + 1] (the 16 lanes of a pixel cover its whole 128-byte
//             row); the .x values are the A operand of the even channels' instruction, the .y values of the odd channels'
//   B[kq][j]: lane (j = l & 15 = (ky, kx), kq) loads window value (ky, kx) of pixel + kq from the image (L1 / L2 resident), zero outside
// i.e. ~25 vector instructions per FOUR pixels and wave instead of ~1 000 lane-instructions per pixel.  A wave walks over runs of 32 pixels
// (eight steps, the next run's loads in flight under this run's instructions), the four waves of a workgroup meet through LDS; slab
// format and reduction are the vector-ALU kernel's.  db: the lanes add what they load, the four kq lanes of a channel pair meet by shuffles.
typedef float f32x4w __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void conv_wgrad_image_mfma_k(const float* __restrict__ dO, const float* __restrict__ X, int64_t rows, int H, int W,
                                                                int ho_shift, int wo_shift, int64_t rows_per_block, float* __restrict__ slab,
                                                                float* __restrict__ dbslab) {
  constexpr int Cout = 32, K = 16, U = 8, PER = Cout * K + Cout;
  __shared__ float red[4][PER];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4, ky = i >> 2, kx = i & 3;
  const int ho = H / 2, wo = W / 2;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  struct Run { float2 a[U]; float b[U]; unsigned mask; };      // mask: bit u = pixel inside the block's rows, bit 8 + u = window tap inside the image
  auto fetch = [&](Run& R, int64_t base) {
    R.mask = 0u;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t pixel = base + 4 * u + kq;
      const bool ok = pixel < r1;
      const int64_t pc = ok ? pixel : r1 - 1;
      R.a[u] = *reinterpret_cast<const float2*>(dO + pc * Cout + 2 * i);
      unsigned img, oy, ox;
      pixel_decode((unsigned)pc, ho, wo, ho_shift, wo_shift, img, oy, ox);
      const int yy = 2 * (int)oy - 1 + ky, xx = 2 * (int)ox - 1 + kx;
      const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
      R.b[u] = X[(int64_t)img * H * W + yc * W + xc];
      R.mask |= (ok ? 1u : 0u) << u;
      R.mask |= ((yy == yc && xx == xc) ? 1u : 0u) << (8 + u);
    }
  };
  f32x4w acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  float bs0 = 0.f, bs1 = 0.f;
  auto consume = [&](const Run& R) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool oka = (R.mask >> u) & 1u, okb = (R.mask >> (8 + u)) & 1u;
      const float ax = oka ? R.a[u].x : 0.f, ay = oka ? R.a[u].y : 0.f, bv = okb ? R.b[u] : 0.f;
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, bv, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ay, bv, acc1, 0, 0, 0);
      bs0 += ax; bs1 += ay;
    }
  };
  constexpr int RUN = 4 * U;                           // pixels per run
  Run Ra, Rb;
  int64_t base = r0 + (int64_t)wave * RUN;
  if (base < r1) fetch(Ra, base);
  for (; base < r1; base += 8 * RUN) {                 // (four waves x two runs per turn)
    const int64_t nb = base + 4 * RUN, nn = base + 8 * RUN;
    if (nb < r1) fetch(Rb, nb);
    consume(Ra);
    if (nb < r1) {
      if (nn < r1) fetch(Ra, nn);
      consume(Rb);
    }
  }
  // the four kq lanes of a channel pair
  bs0 += __shfl_xor(bs0, 16, 64); bs1 += __shfl_xor(bs1, 16, 64);
  bs0 += __shfl_xor(bs0, 32, 64); bs1 += __shfl_xor(bs1, 32, 64);
  // D[row = 4 kq + r][col = i]: row = channel pair, col = window index
  float* mine = red[wave];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int cp = 4 * kq + r;
    mine[(2 * cp) * K + i] = acc0[r];
    mine[(2 * cp + 1) * K + i] = acc1[r];
  }
  if (kq == 0) { mine[Cout * K + 2 * i] = bs0; mine[Cout * K + 2 * i + 1] = bs1; }
  __syncthreads();
  for (int e = threadIdx.x; e < PER; e += 256) {
    const float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    if (e < Cout * K) slab[(int64_t)blockIdx.x * Cout * K + e] = v;
    else if (dbslab) dbslab[(int64_t)blockIdx.x * Cout + (e - Cout * K)] = v;
  }
}

// Weight re-ordering between nn.Conv2d's [co][c][ky][kx] and the GEMM layouts (forward rows, data-gradient rows, the zero-padded rows of
// the 4 x 4 stage) and back for the gradients: up to MAXG index-mapped copies in ONE launch, dst[e] = map[e] >= 0 ? src[map[e]] : 0.
// The maps are permutations the host builds once per shape (cl_ica_amd/conv.py applies its layout functions to an index tensor).
constexpr int MAXGATHER = 16;
struct GatherArgs { int n, accumulate; const float* src[MAXGATHER]; const int* map[MAXGATHER]; float* dst[MAXGATHER]; int count[MAXGATHER]; };
__global__ __launch_bounds__(256) void conv_gather_k(GatherArgs G) {
  const int s = blockIdx.y;
  const float* __restrict__ src = G.src[s];
  const int* __restrict__ map = G.map[s];            // nullptr: identity
  float* __restrict__ dst = G.dst[s];
  for (int e = blockIdx.x * 256 + threadIdx.x; e < G.count[s]; e += gridDim.x * 256) {
    const int m = map ? map[e] : e;
    const float v = m >= 0 ? src[m] : 0.f;
    dst[e] = G.accumulate ? dst[e] + v : v;
  }
}

struct PatchWgradPlan { int blocks; int64_t rows_per_block; };
static bool patch_wgrad_ok(int32_t Cout, int32_t K) {
  if (Cout % 4 || K % 4) return false;
  const int gsz = (Cout / 4) * (K / 4);
  return gsz <= 256 && 256 % gsz == 0 && (size_t)(256 / gsz) * (Cout * K + Cout) * sizeof(float) <= 64 * 1024;
}
static PatchWgradPlan plan_patch_wgrad(int64_t rows) {
  PatchWgradPlan p;
  p.rows_per_block = std::max<int64_t>(256, ceil_div(rows, (int64_t)kNumCU * 4));      // (x 8: twice the slabs for the reduction to walk, the kernel no faster)
  p.blocks = (int)ceil_div(rows, p.rows_per_block);
  return p;
}

}  // namespace gemm
}  // namespace clica

extern "C" int clica_conv_im2col_k4s2(const float* x, int64_t images, int32_t C, int32_t H, int32_t W, float* patches,
                                      clica_stream_t stream) {
  CLICA_CHECK_ARG(x && patches && images > 0 && C >= 1 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0,
                  "clica_conv_im2col_k4s2: bad argument (even H, W required)");
  const int64_t pixels = images * (H / 2) * (W / 2);
  CLICA_CHECK_ARG(pixels * 16 < ((int64_t)1 << 32), "clica_conv_im2col_k4s2: %lld output pixels (< 2^28 supported)", (long long)pixels);
  if (C == 1 && aligned16(patches))
    hipLaunchKernelGGL(conv_im2col_k4s2_c1_k, dim3((unsigned)ceil_div(pixels * 4, 256)), dim3(256), 0, as_stream(stream), x, (unsigned)pixels,
                       (int)H, (int)W, patches);
  else
    hipLaunchKernelGGL(conv_im2col_k4s2_k, dim3((unsigned)ceil_div(pixels * 16, 256)), dim3(256), 0, as_stream(stream), x, (unsigned)pixels,
                       (int)C, (int)H, (int)W, patches);
  return launch_status("clica_conv_im2col_k4s2");
}

extern "C" int clica_conv_k4s2_fwd_patches_amax(const float* patches, const float* Wg, const float* bias, int64_t images, int32_t K,
                                                int32_t Cout, int32_t ho, int32_t wo, int32_t relu, int32_t scatter, float* out,
                                                uint32_t* gate_bits, uint32_t* amax_slots, clica_stream_t stream);
extern "C" int clica_conv_k4s2_fwd_patches(const float* patches, const float* Wg, const float* bias, int64_t images, int32_t K,
                                           int32_t Cout, int32_t ho, int32_t wo, int32_t relu, int32_t scatter, float* out,
                                           uint32_t* gate_bits, clica_stream_t stream) {
  return clica_conv_k4s2_fwd_patches_amax(patches, Wg, bias, images, K, Cout, ho, wo, relu, scatter, out, gate_bits, nullptr, stream);
}
extern "C" int clica_conv_k4s2_fwd_patches_amax(const float* patches, const float* Wg, const float* bias, int64_t images, int32_t K,
                                                int32_t Cout, int32_t ho, int32_t wo, int32_t relu, int32_t scatter, float* out,
                                                uint32_t* gate_bits, uint32_t* amax_slots, clica_stream_t stream) {
  CLICA_CHECK_ARG(patches && Wg && out && images > 0 && K >= 4 && K % 4 == 0 && Cout >= 1 && ho >= 1 && wo >= 1,
                  "clica_conv_k4s2_fwd_patches: bad argument");
  CLICA_CHECK_ARG(!scatter || (ho % 2 == 0 && wo % 2 == 0), "clica_conv_k4s2_fwd_patches: scatter needs an even output grid");
  CLICA_CHECK_ARG(aligned16(patches) && aligned16(Wg), "clica_conv_k4s2_fwd_patches: operands must be 16-byte aligned");
  if (scatter == 1 && K == 16 && Cout == 32 && aligned16(out) && images * ho * wo < ((int64_t)1 << 32)) {
    // short contraction (one input channel): vector-ALU kernel, bound by the 128 B per pixel it writes
    hipLaunchKernelGGL(conv_fwd_patches_valu_k<16>, dim3((unsigned)kNumCU * 8), dim3(256), 0, as_stream(stream), patches, Wg, bias,
                       (unsigned)(images * ho * wo), (int)Cout, (int)ho, (int)wo, (int)relu, out, gate_bits, amax_slots);
    return launch_status("clica_conv_k4s2_fwd_patches(valu)");
  }
  CLICA_CHECK_ARG(!amax_slots, "clica_conv_k4s2_fwd_patches_amax: the maximum is recorded by the K = 16, Cout = 32 first-stage kernel only");
  return conv_fwd_launch(patches, K, 0, 0, Wg, bias, images * ho * wo, K, Cout, ho, wo, ho, wo, relu, scatter, out, gate_bits, as_stream(stream),
                         "clica_conv_k4s2_fwd_patches");
}

extern "C" int clica_conv_k4s2_fwd(const float* S, const float* Wg, const float* bias, int64_t images, int32_t C, int32_t Cout,
                                   int32_t hs, int32_t ws, int32_t relu, int32_t scatter, float* out, uint32_t* gate_bits,
                                   clica_stream_t stream) {
  CLICA_CHECK_ARG(S && Wg && out && images > 0 && C >= 1 && C % 4 == 0 && Cout >= 1 && hs >= 2 && ws >= 2,
                  "clica_conv_k4s2_fwd: bad argument (C must be a multiple of 4)");
  CLICA_CHECK_ARG(scatter != 1 || ((hs - 1) % 2 == 0 && (ws - 1) % 2 == 0), "clica_conv_k4s2_fwd: scatter = 1 needs an even output grid");
  CLICA_CHECK_ARG(aligned16(S) && aligned16(Wg), "clica_conv_k4s2_fwd: operands must be 16-byte aligned");
  // scattering stages run their GEMM over the output pixels only (no work on the grid's non-output rows: 13 / 27 % of the 17 x 17 / 9 x 9 stages)
  return conv_fwd_launch(S, 4 * (int64_t)C, 8 * (int64_t)C, (int64_t)(ws - 2) * 4 * C, Wg, bias, images * hs * ws, 16 * C, Cout,
                         hs, ws, hs - 1, ws - 1, relu, scatter, out, gate_bits, as_stream(stream), "clica_conv_k4s2_fwd", 1);
}

extern "C" int clica_conv_k4s2_dgrad(const float* dO, const float* Wd, const float* S, int64_t images, int32_t C, int32_t Cout,
                                     int32_t hs, int32_t ws, float* dPrev, int32_t dhs, int32_t dws, const uint32_t* gate_bits,
                                     clica_stream_t stream) {
  CLICA_CHECK_ARG(dO && Wd && dPrev && images > 0 && C >= 1 && C % 4 == 0 && Cout >= 1 && Cout % 4 == 0 && hs >= 2 && ws >= 2,
                  "clica_conv_k4s2_dgrad: bad argument (C, Cout must be multiples of 4)");
  CLICA_CHECK_ARG(dhs >= 2 * (hs - 1) && dws >= 2 * (ws - 1), "clica_conv_k4s2_dgrad: destination grid smaller than 2 (hs - 1) x 2 (ws - 1)");
  CLICA_CHECK_ARG(aligned16(dO) && aligned16(Wd), "clica_conv_k4s2_dgrad: operands must be 16-byte aligned");
  // dS[r][j] = sum_k A[r][k] Wd[k][j],  k = (1 - dy, 1 - dx, co):  A[r][k] = dO_flat[(r - ws - 1) Cout + k + (k >= 2 Cout ? (ws - 2) Cout : 0)]
  Args g{}; g.A = dO - (int64_t)(ws + 1) * Cout; g.lda = Cout; g.B = Wd; g.ldb = 4 * (int64_t)C; g.C = dPrev; g.ldc = C;
  g.M = images * hs * ws; g.N = 4 * (int64_t)C; g.Kc = 4 * (int64_t)Cout;
  g.xact = S; g.ldxa = 4 * (int64_t)C; g.slope = 0.f;
  ConvX cx{}; cx.a_seg = 2 * (int64_t)Cout; cx.a_jump = (int64_t)(ws - 2) * Cout; cx.mode = 2;
  cx.hs = hs; cx.ws = ws; cx.ho = hs; cx.wo = ws; cx.dhs = dhs; cx.dws = dws; cx.dho = 2 * (hs - 1); cx.dwo = 2 * (ws - 1); cx.c = C;
  cx.inv_pix = 1.f / (float)(hs * ws); cx.inv_ws = 1.f / (float)ws;
  cx.gate_in = gate_bits;
  cx.fast = ((4 * Cout) % BK == 0 && cx.a_jump < (1 << 30)) ? 1 : 0;
  CLICA_CHECK_ARG(!gate_bits || C % 32 == 0, "clica_conv_k4s2_dgrad: gate bits need C %% 32 == 0");
  CLICA_CHECK_ARG(g.M < (1 << 24), "clica_conv_k4s2_dgrad: %lld rows (the scattering epilogue handles < 2^24)", (long long)g.M);
  if (C == 32 && Cout == 32 && gate_bits && g.M >= 8 * DG_ROWS) {
    auto k = conv_dgrad32_stream_k;
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024), true);
    (void)once;
    const int64_t ntiles = ceil_div(g.M, DG_ROWS);
    hipLaunchKernelGGL(k, dim3((unsigned)std::min<int64_t>(ntiles, 2 * kNumCU)), dim3(512), 64 * 1024, as_stream(stream), g.A, Wd, g.M, dPrev, cx);
    return launch_status("clica_conv_k4s2_dgrad(persistent)");
  }
  // tile shape measured on config 5 (steps/s): 64 x 128 / 4 waves 391; 128 x 128 / 8 waves 402 and 393 (2 x 4, 4 x 2 waves); three LDS stages 352
  // -- within one box's spread except the last; the smallest LDS footprint is kept
  return launch_conv<64, 128, 2, 2, 2, true, false, EPI_DACT>(g, cx, 1, as_stream(stream), "clica_conv_k4s2_dgrad");
}

extern "C" int clica_conv_k4s2_wgrad_workspace_bytes(int64_t rows, int32_t Cout, int32_t K, size_t* bytes) {
  CLICA_CHECK_ARG(bytes && rows > 0 && Cout >= 1 && K >= 1, "clica_conv_k4s2_wgrad_workspace_bytes: bad argument");
  const ConvWgradPlan p = plan_conv_wgrad(rows, Cout, K);
  *bytes = align_up((size_t)p.splits * Cout * K * sizeof(float), 256) + align_up((size_t)p.splits * Cout * sizeof(float), 256);
  return CLICA_OK;
}

extern "C" int clica_conv_k4s2_wgrad(const float* dO, const float* S, int64_t images, int32_t C, int32_t Cout, int32_t hs, int32_t ws,
                                     float* dWg, float* db, int32_t accumulate, void* workspace, size_t workspace_bytes,
                                     clica_stream_t stream) {
  CLICA_CHECK_ARG(dO && S && dWg && workspace && images > 0 && C >= 1 && C % 4 == 0 && Cout >= 1 && Cout % 4 == 0 && hs >= 2 && ws >= 2,
                  "clica_conv_k4s2_wgrad: bad argument (C, Cout must be multiples of 4)");
  CLICA_CHECK_ARG(aligned16(dO) && aligned16(S), "clica_conv_k4s2_wgrad: operands must be 16-byte aligned");
  const int64_t rows = images * hs * ws;
  const int32_t K = 16 * C;
  const ConvWgradPlan p = plan_conv_wgrad(rows, Cout, K);
  const size_t slab_bytes = align_up((size_t)p.splits * Cout * K * sizeof(float), 256);
  const size_t need = slab_bytes + align_up((size_t)p.splits * Cout * sizeof(float), 256);
  if (need > workspace_bytes) { set_error("clica_conv_k4s2_wgrad: workspace %zu < %zu", workspace_bytes, need); return CLICA_E_WORKSPACE; }
  float* slab = (float*)workspace;
  float* dbslab = (float*)((char*)workspace + slab_bytes);
  hipStream_t st = as_stream(stream);
  // dWg[Cout][16C] = dO[rows][Cout]^T A[rows][16C]: "M" = Cout, "N" = 16C, contraction over the rows; B is the two-run operand
  Args g{}; g.A = dO; g.lda = Cout; g.B = S; g.ldb = 4 * (int64_t)C; g.C = slab; g.ldc = K; g.M = Cout; g.N = K; g.Kc = rows;
  g.k_per_split = p.k_per_split; g.dbias_slab = db ? dbslab : nullptr;
  ConvX cx{}; cx.b_seg = 8 * (int64_t)C; cx.b_jump = (int64_t)(ws - 2) * 4 * C;
  int rc = p.bm == 32 ? launch_conv<32, 128, 1, 4, 2, false, false, EPI_SLAB>(g, cx, p.splits, st, "clica_conv_k4s2_wgrad")
                      : launch_conv<64, 128, 2, 2, 2, false, false, EPI_SLAB>(g, cx, p.splits, st, "clica_conv_k4s2_wgrad");
  if (rc) return rc;
  launch_slab_reduce(slab, dbslab, p.splits, Cout, K, dWg, K, db, accumulate ? 1 : 0, st);
  return launch_status("clica_conv_k4s2_wgrad(reduce)");
}

extern "C" int clica_conv_k4s2_wgrad_patches_workspace_bytes(int64_t rows, int32_t Cout, int32_t K, size_t* bytes) {
  CLICA_CHECK_ARG(bytes && rows > 0 && Cout >= 1 && K >= 1, "clica_conv_k4s2_wgrad_patches_workspace_bytes: bad argument");
  CLICA_CHECK_ARG(patch_wgrad_ok(Cout, K), "clica_conv_k4s2_wgrad_patches: Cout = %d, K = %d not supported (multiples of 4 with (Cout/4)(K/4) "
                  "dividing 256); use clica_mlp_wgrad on the patch matrix", Cout, K);
  const PatchWgradPlan p = plan_patch_wgrad(rows);
  *bytes = align_up((size_t)p.blocks * Cout * K * sizeof(float), 256) + align_up((size_t)p.blocks * Cout * sizeof(float), 256);
  return CLICA_OK;
}

extern "C" int clica_conv_k4s2_wgrad_patches(const float* dO, const float* patches, int64_t rows, int32_t Cout, int32_t K, float* dWg,
                                             float* db, int32_t accumulate, void* workspace, size_t workspace_bytes, clica_stream_t stream) {
  CLICA_CHECK_ARG(dO && patches && dWg && workspace && rows > 0, "clica_conv_k4s2_wgrad_patches: NULL pointer / no rows");
  CLICA_CHECK_ARG(patch_wgrad_ok(Cout, K), "clica_conv_k4s2_wgrad_patches: Cout = %d, K = %d not supported (multiples of 4 with (Cout/4)(K/4) "
                  "dividing 256); use clica_mlp_wgrad on the patch matrix", Cout, K);
  CLICA_CHECK_ARG(aligned16(dO) && aligned16(patches), "clica_conv_k4s2_wgrad_patches: operands must be 16-byte aligned");
  const PatchWgradPlan p = plan_patch_wgrad(rows);
  const size_t slab_bytes = align_up((size_t)p.blocks * Cout * K * sizeof(float), 256);
  const size_t need = slab_bytes + align_up((size_t)p.blocks * Cout * sizeof(float), 256);
  if (need > workspace_bytes) { set_error("clica_conv_k4s2_wgrad_patches: workspace %zu < %zu", workspace_bytes, need); return CLICA_E_WORKSPACE; }
  float* slab = (float*)workspace;
  float* dbslab = (float*)((char*)workspace + slab_bytes);
  hipStream_t st = as_stream(stream);
  const int G = 256 / ((Cout / 4) * (K / 4));
  const size_t lds = (size_t)G * (Cout * K + Cout) * sizeof(float);
  hipLaunchKernelGGL(conv_wgrad_patches_k, dim3((unsigned)p.blocks), dim3(256), lds, st, dO, patches, rows, (int)Cout, (int)K, p.rows_per_block,
                     slab, db ? dbslab : (float*)nullptr);
  int rc = launch_status("clica_conv_k4s2_wgrad_patches");
  if (rc) return rc;
  // thousands of small slabs: the 16-way split reduction of conv16.hip (the shared slab_reduce_k walks them as four chains: 31 us here)
  if ((Cout * K) % 4 == 0 && Cout % 4 == 0 && aligned16(dWg) && (!db || aligned16(db)))
    conv16::launch_slab_sum(slab, Cout * K, dWg, db ? dbslab : nullptr, Cout, db, p.blocks, accumulate ? 1 : 0, st);
  else
    launch_slab_reduce(slab, dbslab, p.blocks, Cout, K, dWg, K, db, accumulate ? 1 : 0, st);
  return launch_status("clica_conv_k4s2_wgrad_patches(reduce)");
}

static int shift_of(int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; }

extern "C" int clica_conv_k4s2_fwd_image(const float* x, const float* Wg, const float* bias, int64_t images, int32_t H, int32_t W, int32_t Cout,
                                         int32_t relu, float* out, uint32_t* gate_bits, uint32_t* amax_slots, clica_stream_t stream) {
  CLICA_CHECK_ARG(x && Wg && out && images > 0 && H >= 4 && W >= 4 && H % 4 == 0 && W % 4 == 0 && Cout == 32,
                  "clica_conv_k4s2_fwd_image: bad argument (one input channel, Cout = 32, H and W multiples of 4)");
  CLICA_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 7) == 0 && aligned16(out), "clica_conv_k4s2_fwd_image: x must be 8-byte, out 16-byte aligned");
  const int64_t pixels = images * (H / 2) * (W / 2);
  CLICA_CHECK_ARG(pixels < ((int64_t)1 << 31), "clica_conv_k4s2_fwd_image: %lld output pixels (< 2^31 supported)", (long long)pixels);
  constexpr int ablate = 0;
  hipLaunchKernelGGL(conv_fwd_image_valu_k, dim3((unsigned)kNumCU * 8), dim3(256), 0, as_stream(stream), x, Wg, bias, (unsigned)pixels, (int)Cout,
                     (int)H, (int)W, shift_of(H / 2), shift_of(W / 2), (int)relu, out, gate_bits, amax_slots, ablate);
  return launch_status("clica_conv_k4s2_fwd_image");
}

extern "C" int clica_conv_k4s2_wgrad_image(const float* dO, const float* x, int64_t images, int32_t H, int32_t W, int32_t Cout, float* dWg, float* db,
                                           int32_t accumulate, void* workspace, size_t workspace_bytes, clica_stream_t stream) {
  CLICA_CHECK_ARG(dO && x && dWg && workspace && images > 0 && H >= 4 && W >= 4 && H % 4 == 0 && W % 4 == 0,
                  "clica_conv_k4s2_wgrad_image: NULL pointer / bad shape (H and W multiples of 4)");
  const int32_t K = 16;
  CLICA_CHECK_ARG(Cout == 8 || Cout == 16 || Cout == 32 || Cout == 64 || Cout == 128, "clica_conv_k4s2_wgrad_image: Cout = %d not supported (8 ... 128, a power of two)", Cout);
  CLICA_CHECK_ARG(aligned16(dO) && (reinterpret_cast<uintptr_t>(x) & 7) == 0, "clica_conv_k4s2_wgrad_image: dO must be 16-byte, x 8-byte aligned");
  const int64_t rows = images * (H / 2) * (W / 2);
  CLICA_CHECK_ARG(rows < ((int64_t)1 << 32), "clica_conv_k4s2_wgrad_image: %lld rows (< 2^32 supported)", (long long)rows);
  const PatchWgradPlan p = plan_patch_wgrad(rows);
  const size_t slab_bytes = align_up((size_t)p.blocks * Cout * K * sizeof(float), 256);
  const size_t need = slab_bytes + align_up((size_t)p.blocks * Cout * sizeof(float), 256);
  if (need > workspace_bytes) { set_error("clica_conv_k4s2_wgrad_image: workspace %zu < %zu (clica_conv_k4s2_wgrad_patches_workspace_bytes)", workspace_bytes, need); return CLICA_E_WORKSPACE; }
  float* slab = (float*)workspace;
  float* dbslab = (float*)((char*)workspace + slab_bytes);
  hipStream_t st = as_stream(stream);
  const size_t lds = (size_t)4 * (Cout * K + Cout) * sizeof(float);
  if (Cout == 32)
    hipLaunchKernelGGL(conv_wgrad_image_mfma_k, dim3((unsigned)p.blocks), dim3(256), 0, st, dO, x, rows, (int)H, (int)W, shift_of(H / 2), shift_of(W / 2),
                       p.rows_per_block, slab, db ? dbslab : (float*)nullptr);
  else
    hipLaunchKernelGGL(conv_wgrad_image_k, dim3((unsigned)p.blocks), dim3(256), lds, st, dO, x, rows, (int)Cout, (int)H, (int)W, shift_of(H / 2), shift_of(W / 2),
                       p.rows_per_block, slab, db ? dbslab : (float*)nullptr);
  int rc = launch_status("clica_conv_k4s2_wgrad_image");
  if (rc) return rc;
  if ((Cout * K) % 4 == 0 && Cout % 4 == 0 && aligned16(dWg) && (!db || aligned16(db)))
    conv16::launch_slab_sum(slab, Cout * K, dWg, db ? dbslab : nullptr, Cout, db, p.blocks, accumulate ? 1 : 0, st);
  else
    launch_slab_reduce(slab, dbslab, p.blocks, Cout, K, dWg, K, db, accumulate ? 1 : 0, st);
  return launch_status("clica_conv_k4s2_wgrad_image(reduce)");
}

// ---- data gradient of the FIRST stage: d loss / d image (kitti_masks/model.py:41-56 is plain nn.Conv2d, differentiable w.r.t. its input) ----
// The training step never needs it (the encoder's input is data); a caller that does differentiate through the images (saliency, an
// adversarial probe) got nn.Conv2d / MIOpen until round 5.  dX[img][c][y][x] = sum over the <= 2 x 2 output pixels whose 4 x 4 / stride-2 /
// pad-1 window covers (y, x) of dO[pixel][co] W[co][c][ky][kx]: one thread per input pixel, the (Cout x C x 16) weights in LDS, the four
// dO rows it needs (Cout floats each) read as float4s -- neighbouring threads share them through L1 / L2.  HBM-bound: dO read once
// (images x ho x wo x Cout x 4 B), dX written once.
namespace clica {
namespace convin {
constexpr int MAXW = 32 * 4 * 16;      // Cout x C x 16 floats of weights in LDS (Cout = 32, C <= 4)
__global__ __launch_bounds__(256) void dgrad_input_k(const float* __restrict__ dO, const float* __restrict__ Wc, int64_t images, int C, int Cout,
                                                    int H, int Wd, float* __restrict__ dX) {
  __shared__ float w[MAXW];
  for (int i = threadIdx.x; i < Cout * C * 16; i += 256) w[i] = Wc[i];
  __syncthreads();
  const int ho = H / 2, wo = Wd / 2;
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= images * H * Wd) return;
  const int x = (int)(pix % Wd), y = (int)((pix / Wd) % H);
  const int64_t img = pix / ((int64_t)Wd * H);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int oy = ((y + 1) >> 1) - a, ky = ((y + 1) & 1) + 2 * a;      // 2 oy + ky - 1 = y
    if (oy < 0 || oy >= ho) continue;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int ox = ((x + 1) >> 1) - b, kx = ((x + 1) & 1) + 2 * b;
      if (ox < 0 || ox >= wo) continue;
      const float4* row = reinterpret_cast<const float4*>(dO + ((img * ho + oy) * wo + ox) * Cout);
      for (int c4 = 0; c4 < Cout / 4; ++c4) {
        const float4 g = row[c4];
        const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* wr = w + ((c4 * 4 + e) * C) * 16 + ky * 4 + kx;
          for (int c = 0; c < C; ++c) acc[c] = fmaf(gv[e], wr[c * 16], acc[c]);
        }
      }
    }
  }
  for (int c = 0; c < C; ++c) dX[((img * C + c) * H + y) * Wd + x] = acc[c];
}
}  // namespace convin
}  // namespace clica

extern "C" int clica_conv_k4s2_dgrad_input(const float* dO, const float* W, int64_t images, int32_t C, int32_t Cout, int32_t H, int32_t Wd,
                                           float* dX, clica_stream_t stream) {
  CLICA_CHECK_ARG(dO && W && dX && images > 0, "clica_conv_k4s2_dgrad_input: NULL pointer / empty batch");
  CLICA_CHECK_ARG(C >= 1 && C <= 4 && Cout >= 4 && Cout % 4 == 0 && Cout * C * 16 <= convin::MAXW && H >= 2 && Wd >= 2 && H % 2 == 0 && Wd % 2 == 0,
                  "clica_conv_k4s2_dgrad_input: unsupported shape (C = %d <= 4, Cout = %d with Cout * C <= 128, even H x W = %d x %d)", C, Cout, H, Wd);
  CLICA_CHECK_ARG((reinterpret_cast<uintptr_t>(dO) & 15) == 0, "clica_conv_k4s2_dgrad_input: dO must be 16-byte aligned");
  const int64_t n = images * H * Wd;
  hipLaunchKernelGGL(convin::dgrad_input_k, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), dO, W, images, (int)C, (int)Cout,
                     (int)H, (int)Wd, dX);
  return launch_status("clica_conv_k4s2_dgrad_input");
}

extern "C" int clica_conv_gather(int32_t n, const float* const* src, const int32_t* const* map, float* const* dst, const int32_t* count,
                                 int32_t accumulate, clica_stream_t stream) {
  CLICA_CHECK_ARG(n >= 1 && n <= MAXGATHER && src && map && dst && count, "clica_conv_gather: 1..%d segments", MAXGATHER);
  GatherArgs G{}; G.n = n; G.accumulate = accumulate ? 1 : 0;
  int most = 0;
  for (int i = 0; i < n; ++i) {
    CLICA_CHECK_ARG(src[i] && dst[i] && count[i] >= 0, "clica_conv_gather: segment %d: bad argument", i);
    G.src[i] = src[i]; G.map[i] = map[i]; G.dst[i] = dst[i]; G.count[i] = count[i];
    most = std::max(most, (int)count[i]);
  }
  if (most == 0) return CLICA_OK;
  hipLaunchKernelGGL(conv_gather_k, dim3((unsigned)std::min<int64_t>(ceil_div(most, 256), 256), (unsigned)n), dim3(256), 0, as_stream(stream), G);
  return launch_status("clica_conv_gather");
}
