// Whole-encoder forward in ONE launch: the Linear(+bias)(+LeakyReLU) stack of get_mlp
// (/root/reference/encoders.py:36-48) with the activation panel resident on chip.
//
// Why: at n = 10 (widths 10-100-500-500-500-500-100-10, main_mlp.py:297-307) a per-layer GEMM launch
// pays a fixed ~15 us (first-tile HBM latency, the 24.6 MB epilogue store, the launch boundary) next to
// ~45 us of MFMA time.  Here every workgroup owns 48 rows of the 2B-row batch for the whole stack:
//   * the 48 x width activation panel lives in LDS (48 x 520 floats = 100 KB); a layer reads it as the
//     MFMA A operand and, after a barrier, overwrites it with its own output (accumulators hold the
//     results meanwhile), which is also streamed to HBM once (coalesced, straight from the panel) because
//     the backward needs the saved activations;
//   * weights are the B operand.  A wave owns whole 16-column blocks of the output, so a B fragment is
//     private to the wave: it is loaded global -> VGPR directly (16 B per lane, L2-resident: every CU
//     streams the same <= 1 MB matrix) in fragment order, requested one 32-deep k-iteration ahead into the other of two
//     register buffers (ping-pong, see layer_gemm); no LDS staging, no barrier in the k-loop;
//   * math: v_mfma_f32_16x16x4_f32 (exact fp32, same as the tiled GEMMs), 3 row blocks x <= 4 column
//     blocks per wave, k-permuted so one 16-byte read feeds four MFMAs on both operands.
// One launch replaces seven; the panel never round-trips through HBM between layers.
#include "common.h"
#include "planes.h"
#include "split16.h"
#include "wgrad_shared.h"
#include "sampler_dev.h"
#include "lp_guard.h"
#include <stdlib.h>

namespace clica {
namespace fmlp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWS = 48;            // rows per workgroup = 3 MFMA row blocks
constexpr int RB = ROWS / 16;
#ifndef CLICA_FMLP_WAVES
#define CLICA_FMLP_WAVES 8
#endif
constexpr int WAVES = CLICA_FMLP_WAVES;
constexpr int THREADS = 64 * WAVES;
constexpr int MAXW = 512;           // widest layer the panel holds
#ifndef CLICA_FMLP_LDPAD
#define CLICA_FMLP_LDPAD 8
#endif
constexpr int LDP = MAXW + CLICA_FMLP_LDPAD;       // panel leading dimension: conflict-free ds_read_b128 of 16 rows x 4 k-groups
constexpr int CBW = (MAXW / 16 + WAVES - 1) / WAVES;   // column blocks per wave (4)
constexpr int MAXL = 8;
#ifndef FP32_STORE_AUX
#define FP32_STORE_AUX 2      // activation stores of the fp32 kernel: 2 = streaming (nt)
#endif

struct Layer {
  const float* W; int64_t ldw; const float* bias;
  float* out; int64_t ldo;           // HBM copy of the layer output (saved activation / final result / dZ)
  const float* aux; int64_t ldaux;   // dact mode: saved activation [M, N] whose sign gives act' (slow path, see mask_in)
  unsigned long long* mask_out;      // forward: per-lane sign bits of this layer's output in accumulator order (or nullptr)
  const unsigned long long* mask_in; // dact mode: the sign bits the forward stored for the same [M, N] panel (or nullptr)
  int N, K, leaky;
  int dact;                          // 0: out = act(acc + bias)   1: out = acc * act'(aux)   (backward data chain)
  // split-bf16 kernel only: HBM copy of the layer output as three bf16 planes in the weight-gradient kernel's operand
  // format (wgrad_split.hip, clica_mlp_planes_bytes), or nullptr; `out` may then be nullptr (no fp32 copy)
  unsigned short* planes; int pl_units; int pl_ones;
};
struct Args {
  const float* X; int64_t ldx; int64_t M; int L; float slope;
  const float* packed;              // fragment-order weights (clica_mlp_pack) or nullptr
  int64_t pack_off[MAXL];           // float offset of each layer inside `packed`
  // optional mixing-net prologue (clica_mlp_fwd_mixed): X is the latent block Z, the stack's input is g(Z)
  const float* mixW; int mixL; float mix_slope; float* xout; int64_t ldxo;
  Layer layer[MAXL];
};
constexpr int MIX_MAX_N = 16;       // widest mixing net the prologue handles (ROWS * n values over THREADS threads, <= 2 each)

__device__ __attribute__((aligned(16))) float g_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

// Debug build only (-DCLICA_FMLP_TRACE, tools/fmlp_trace.py): s_memtime stamps per (workgroup, wave, layer, phase).
#ifdef CLICA_FMLP_TRACE
__device__ unsigned long long* g_trace = nullptr;
#define FMLP_STAMP(l, ph) do { if (g_trace && lane == 0) g_trace[(((size_t)blockIdx.x * WAVES + wave) * (MAXL + 1) + (l)) * 8 + (ph)] = clock64(); } while (0)
#else
#define FMLP_STAMP(l, ph) do { } while (0)
#endif

// B fragment of column block `cb` for k in [k0 + 4q, k0 + 4q + 4): W[n = cb*16 + (lane&15)][k..k+3]
template <bool VEC>
__device__ __forceinline__ float4 load_b(const Layer& ly, int n, int k) {
  if (VEC) {
    const bool in = n < ly.N && k < ly.K;        // K % 4 == 0 on this path: a float4 is fully in or out
    return *reinterpret_cast<const float4*>(in ? ly.W + (int64_t)n * ly.ldw + k : g_zero_page);
  }
  const float* row = ly.W + (int64_t)(n < ly.N ? n : 0) * ly.ldw;
  const bool r = n < ly.N;
  return make_float4(*((r && k < ly.K) ? row + k : g_zero_page), *((r && k + 1 < ly.K) ? row + k + 1 : g_zero_page),
                     *((r && k + 2 < ly.K) ? row + k + 2 : g_zero_page), *((r && k + 3 < ly.K) ? row + k + 3 : g_zero_page));
}

// One k-iteration covers 32 k: lane group q reads k0+4q..+3 and k0+16+4q..+3, i.e. the two 16-byte loads
// of a lane group complete a full 128-byte line of each weight row (half-line requests would fetch
// every L2 line twice: the 32 KB per-step working set of the 8 waves does not survive in L1).
constexpr int KI = 32;
#ifndef CLICA_FMLP_PINGPONG
#define CLICA_FMLP_PINGPONG 1
#endif
// NC = number of 16-column blocks this wave owns in this layer (compile time: the MFMA stream must be
// branch-free; NC is wave-uniform and selected by a scalar switch in the caller).
// Measured with the -DCLICA_FMLP_TRACE build (tools/fmlp_trace.py, s_memtime per wave / layer / phase) on a 500 x 500 layer:
// ideal MFMA time 98.3k cycles per SIMD; the k-loop of waves 0-3 ends at 82.5k, of waves 4-7 at 107.7k (the two waves of a SIMD
// share its matrix pipe, arbitration is priority then AGE: the older wave takes ~60 % of it and the younger finishes alone),
// then barrier 1.2k + epilogue 5.1k + barrier 1.2k + activation-store issue 3.6k = 120k per layer (82 % of ideal).
// Tried and NOT kept (each re-measured with the trace and with back-to-back launches):
//   * s_setprio for the younger half (static: swaps winner and loser, 112k / 86k; alternating per k-iteration: 102k / 115k;
//     first half of the loop only: 106k / 114k) -- the pair needs ~111k either way: the loop's own load-issue / wait overhead
//     is ~12 % with both waves live, arbitration only decides which wave shows it;
//   * weight fragments two k-iterations ahead (winner 75k, loser still 114k);
//   * pinning the eight weight loads + six panel reads to the FRONT of the MFMA block (sched_barrier / sched_group_barrier):
//     130k-150k -- issuing 8 x 1 KB loads back to back stalls the wave's own MFMA issue for ~1000 cycles, the compiler's
//     placement (loads spread over the LAST third of the block) is the better one;
//   * a rotated k-start per workgroup (spread the 256 CUs' requests for the same weight line): -1.5 % cycles inside the
//     kernel, +2..9 % wall on back-to-back launches.
// The chip holds 2.37 GHz under this load (tools/clock_probe.py) and the bare instruction pattern issues at 95-99 % of the
// 157.3 TFLOP/s peak (tools/proto/mfma_rate.hip), so the remaining ~30 % is this kernel's structure, not a clock ceiling.
template <bool VEC, bool PACKED, int NC>
__device__ __forceinline__ void layer_gemm(const Layer& ly, const float* __restrict__ pk, const float* panel, int wave, int lane,
                                           f32x4 (&acc)[RB][CBW], const float4 (&bpre)[2][CBW]) {
  const int i15 = lane & 15, q = lane >> 4;
  const int kiters = (ly.K + KI - 1) / KI;
  int nrow[CBW];
#pragma unroll
  for (int c = 0; c < CBW; ++c) nrow[c] = (wave + c * WAVES) * 16 + i15;     // column blocks w, w+8, w+16, w+24
  float4 bcur[2][CBW], bnxt[2][CBW], acur[2][RB], anxt[2][RB];
  auto fetch_b = [&](float4 (&b)[2][CBW], int k0) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (PACKED) {   // fragment order: one fully contiguous 1 KB per load instruction, pre-padded with zeros
        const float* base = pk + ((int64_t)((wave + c * WAVES) * kiters + (k0 >> 5)) * 2 * 64 + lane) * 4;
        b[0][c] = *reinterpret_cast<const float4*>(base);
        b[1][c] = *reinterpret_cast<const float4*>(base + 256);
      } else {
        b[0][c] = load_b<VEC>(ly, nrow[c], k0 + 4 * q);
        b[1][c] = load_b<VEC>(ly, nrow[c], k0 + 16 + 4 * q);
      }
    }
  };
  auto fetch_a = [&](float4 (&a)[2][RB], int k0) {
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      a[0][r] = *reinterpret_cast<const float4*>(&panel[(r * 16 + i15) * LDP + k0 + 4 * q]);
      a[1][r] = *reinterpret_cast<const float4*>(&panel[(r * 16 + i15) * LDP + k0 + 16 + 4 * q]);
    }
  };
  [[maybe_unused]] auto mma_and_rotate = [&]() {
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const float4 b4 = bcur[hlf][c];
          const float bv = t == 0 ? b4.x : (t == 1 ? b4.y : (t == 2 ? b4.z : b4.w));
#pragma unroll
          for (int r = 0; r < RB; ++r) {
            const float4 a4 = acur[hlf][r];
            const float av = t == 0 ? a4.x : (t == 1 ? a4.y : (t == 2 ? a4.z : a4.w));
            acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[r][c], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
#pragma unroll
      for (int c = 0; c < NC; ++c) bcur[hlf][c] = bnxt[hlf][c];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        // keep the panel fragment ONE 16-byte LDS read: without a whole-vector use the optimiser scalarises it
        // into four ds_read_b32 (2-way bank conflicts at this row stride) plus an address add each, sunk between the MFMAs
        f32x4 v = {anxt[hlf][r].x, anxt[hlf][r].y, anxt[hlf][r].z, anxt[hlf][r].w};
        asm volatile("" : "+v"(v));
        acur[hlf][r] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  };
  if (PACKED) {     // iteration 0's weights were requested before the previous layer's epilogue (see the kernel)
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf)
#pragma unroll
      for (int c = 0; c < NC; ++c) bcur[hlf][c] = bpre[hlf][c];
  } else {
    fetch_b(bcur, 0);
  }
  fetch_a(acur, 0);
#if CLICA_FMLP_PINGPONG
  // Ping-pong form: the k-loop is unrolled by two with the roles of the two operand buffers swapped, so there is no
  // register rotation at the end of an iteration (the rotation's moves made the compiler wait for ALL of the next
  // iteration's loads, vmcnt(0) / lgkmcnt(0), a few MFMAs after issuing them: the whole L2 / LDS round trip was exposed
  // in every iteration and only the SIMD's other wave covered it).  sched_group_barrier pins one weight load behind
  // every sixth MFMA of the first half of the block and one panel read behind every 2 NC-th MFMA of the second half.
  auto mma = [&](const float4 (&b)[2][CBW], const float4 (&a)[2][RB]) {
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const float4 b4 = b[hlf][c];
          const float bv = t == 0 ? b4.x : (t == 1 ? b4.y : (t == 2 ? b4.z : b4.w));
#pragma unroll
          for (int r = 0; r < RB; ++r) {
            const float4 a4 = a[hlf][r];
            const float av = t == 0 ? a4.x : (t == 1 ? a4.y : (t == 2 ? a4.z : a4.w));
            acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[r][c], 0, 0, 0);
          }
        }
  };
#ifndef CLICA_FMLP_LOAD_GAP
#define CLICA_FMLP_LOAD_GAP 6
#endif
#ifndef CLICA_FMLP_DS_GAP
#define CLICA_FMLP_DS_GAP (2 * NC)
#endif
#ifndef CLICA_FMLP_DS_FIRST
#define CLICA_FMLP_DS_FIRST 0
#endif
  auto pin = [&]() {
    if (CLICA_FMLP_DS_FIRST) {
#pragma unroll
      for (int i = 0; i < 2 * RB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, CLICA_FMLP_DS_GAP, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
      }
    }
#pragma unroll
    for (int i = 0; i < 2 * NC; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, CLICA_FMLP_LOAD_GAP, 0);      // MFMAs
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read
    }
    if (!CLICA_FMLP_DS_FIRST) {
#pragma unroll
      for (int i = 0; i < 2 * RB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, CLICA_FMLP_DS_GAP, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
      }
    }
  };
  auto kof = [&](int ki) { return (ki < kiters ? ki : kiters - 1) * KI; };
  // (Measured on this loop: the older wave of a SIMD wins the matrix-pipe arbitration, leaves the loop at ~70 k cycles of a
  //  500 x 500 layer and the younger one follows at ~107 k.  Raising the younger wave's priority for the first 9/16 .. 13/16 of
  //  its iterations makes both finish together -- 99 k / 105 k, 104 k / 101 k -- but the LAST wave still arrives at ~105 k:
  //  the pair needs ~105 k cycles for 98.3 k of MFMA work however the pipe is shared; s_setprio not kept.)
  int ki = 0;
  for (; ki + 1 < kiters; ki += 2) {
    fetch_b(bnxt, kof(ki + 1)); fetch_a(anxt, kof(ki + 1));
    mma(bcur, acur);
    pin();
    fetch_b(bcur, kof(ki + 2)); fetch_a(acur, kof(ki + 2));
    mma(bnxt, anxt);
    pin();
  }
  if (ki < kiters) mma(bcur, acur);
#else
  // UNCONDITIONAL prefetch of the next iteration's operands (past the end: a harmless re-read of the last
  // iteration's own operands): with a conditional issue the compiler cannot count the loads in flight and
  // drains them all (s_waitcnt vmcnt(0)) in front of the MFMAs, which serialises fetch and math.
  // Iteration 0 is peeled: the wait in front of ITS MFMAs may leave the previous layer's activation stores
  // (issued after the early weight request) in flight, which a loop-carried wait count could not express.
  {
    const int kn = kiters > 1 ? KI : 0;
    fetch_b(bnxt, kn);
    fetch_a(anxt, kn);
    mma_and_rotate();
  }
  for (int ki = 1; ki < kiters; ++ki) {
    const int kn = (ki + 1 < kiters) ? (ki + 1) * KI : ki * KI;
    fetch_b(bnxt, kn);
    fetch_a(anxt, kn);
    mma_and_rotate();
  }
#endif
}

// iteration-0 weight fragments of a layer, for every column block slot of this wave (slots beyond the layer's
// last block re-read block 0: unconditional loads, their values are never used)
__device__ __forceinline__ void request_first_b(const float* __restrict__ pk, int K, int N, int wave, int lane, float4 (&b)[2][CBW]) {
  const int kiters = (K + KI - 1) / KI, ncb_real = (N + 15) / 16;
#pragma unroll
  for (int c = 0; c < CBW; ++c) {
    const int cb = wave + c * WAVES;
    const float* base = pk + ((int64_t)((cb < ncb_real ? cb : 0) * kiters) * 2 * 64 + lane) * 4;
    b[0][c] = *reinterpret_cast<const float4*>(base);
    b[1][c] = *reinterpret_cast<const float4*>(base + 256);
  }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr int kRsrcWord3 = 0x00020000;       // raw buffer, 32-bit data format (gfx9 family)
constexpr unsigned kOobOffset = 0x80000000u; // beyond any num_records: the access is dropped / reads zero

// Epilogue of one layer into the panel, specialised at compile time (no scalar branches / kernarg re-reads
// per element).  C/D layout of the 16x16 blocks: col = lane & 15, row = (lane >> 4) * 4 + reg.  Columns
// N..round_up(N, KI) are written as zeros: they are the next layer's k-padding.  Returns this lane's sign bits.
enum EpiMode { EPI_BIAS = 0, EPI_BIAS_LEAKY = 1, EPI_DACT_MASK = 2, EPI_DACT_NONE = 3 };
template <int MODE, bool BITS>
__device__ __forceinline__ unsigned long long epilogue_to_panel(float* panel, const f32x4 (&acc)[RB][CBW], const float (&bias)[CBW],
                                                                const unsigned long long mbits, const float slope,
                                                                const int N, const int wave, const int lane) {
  const int ncb = ((N + KI - 1) & ~(KI - 1)) / 16;
  unsigned lo = 0u, hi = 0u;
  const unsigned mlo = (unsigned)mbits, mhi = (unsigned)(mbits >> 32);
#pragma unroll
  for (int c = 0; c < CBW; ++c) {
    const int cb = wave + c * WAVES;
    if (cb < ncb) {                                   // wave-uniform
      const int col = cb * 16 + (lane & 15);
      const bool colin = col < N;
      float* dst = panel + ((lane >> 4) * 4) * LDP + col;
#pragma unroll
      for (int r = 0; r < RB; ++r) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int bit = (c * RB + r) * 4 + e;       // compile time
          float v = acc[r][c][e];
          if (MODE == EPI_BIAS || MODE == EPI_BIAS_LEAKY) v += bias[c];
          if (MODE == EPI_BIAS_LEAKY) v = v > 0.f ? v : v * slope;
          if (MODE == EPI_DACT_MASK) {   // dZ_{l-1} = (dZ_l W_l) * LeakyReLU'(a_{l-1}); sign(a) = sign(pre-activation)
            const unsigned m = bit < 32 ? (mlo >> bit) : (mhi >> (bit - 32));
            v = (m & 1u) ? v : v * slope;
          }
          if (BITS) {
            if (bit < 32) lo |= (v > 0.f) ? (1u << bit) : 0u;
            else hi |= (v > 0.f) ? (1u << (bit - 32)) : 0u;
          }
          dst[(r * 16 + e) * LDP] = colin ? v : 0.f;
        }
      }
    }
  }
  return (unsigned long long)lo | ((unsigned long long)hi << 32);
}

#ifndef CLICA_FMLP_DEFER_STORE
#define CLICA_FMLP_DEFER_STORE 1
#endif
// panel -> HBM copy of one layer's output by a subset of the workgroup's threads (tid in [0, nthreads))
__device__ __forceinline__ void store_panel(const float* panel, const Layer& ly, int64_t row0, int nrows, int tid, int nthreads) {
  const bool ovec = ((reinterpret_cast<uintptr_t>(ly.out) & 15) == 0) && (ly.ldo % 4 == 0) && (ly.N % 4 == 0);
  if (ovec) {
    const int n4 = ly.N / 4;
    const __amdgpu_buffer_rsrc_t orsrc =
        __builtin_amdgcn_make_buffer_rsrc(ly.out + row0 * ly.ldo, 0, (int)(((int64_t)(nrows - 1) * ly.ldo + ly.N) * 4), kRsrcWord3);
    for (int idx = tid; idx < nrows * n4; idx += nthreads) {
      const int r = idx / n4, c4 = idx - r * n4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(&panel[r * LDP + 4 * c4]);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orsrc, (unsigned)((r * (int)ly.ldo + 4 * c4) * 4), 0, FP32_STORE_AUX);
    }
  } else {
    for (int idx = tid; idx < nrows * ly.N; idx += nthreads) {
      const int r = idx / ly.N, c = idx - r * ly.N;
      ly.out[(row0 + r) * ly.ldo + c] = panel[r * LDP + c];
    }
  }
}

// PACKED: weights come in fragment order (g.packed); AUX: some backward link has no sign bits and re-reads its
// saved activation (slow epilogue).  Separate instantiations keep the hot <true, false> kernel small.
template <bool PACKED, bool AUX>
__global__ __launch_bounds__(THREADS) void mlp_fwd_k(Args g) {
  extern __shared__ __attribute__((aligned(16))) float panel[];     // [ROWS][LDP]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // provably wave-uniform: scalar branches
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int nrows = (int)min((int64_t)ROWS, g.M - row0);

  // Sign bits / first weight fragments of layer 0 (for later layers they are requested ahead of the previous
  // layer's epilogue).  Raw-buffer accesses with num_records = 0 when the pointer is NULL: unconditional
  // instructions, so every vector-memory operation of the kernel is statically countable.
  const int mslot = (wave * 64 + lane) * 8;                       // byte offset inside this workgroup's 4 KB of sign bits
  auto mask_rsrc = [&](const unsigned long long* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(p) + (p ? (int64_t)blockIdx.x * WAVES * 64 : 0), 0,
                                             p ? WAVES * 64 * 8 : 0, kRsrcWord3);
  };
  // bias of this lane's column in each of the wave's column blocks (NULL bias / columns >= N read as 0)
  auto request_bias = [&](const Layer& ly, float (&b)[CBW]) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ly.bias), 0, ly.bias ? ly.N * 4 : 0, kRsrcWord3);
#pragma unroll
    for (int c = 0; c < CBW; ++c)
      b[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, ((wave + c * WAVES) * 16 + (lane & 15)) * 4, 0, 0));
  };
  float4 bpre[2][CBW];
  float bias[CBW];
  if (PACKED) request_first_b(g.packed + g.pack_off[0], g.layer[0].K, g.layer[0].N, wave, lane, bpre);
  u32x2 mraw = __builtin_amdgcn_raw_buffer_load_b64(mask_rsrc(g.layer[0].dact ? g.layer[0].mask_in : nullptr), mslot, 0, 0);
  request_bias(g.layer[0], bias);

  // input panel, zero-padded to a multiple of KI columns (the k-loop runs in KI-deep iterations)
  if (g.mixW) {
    // x = g(z): the frozen mixing MLP (n x n bias-free layers, LeakyReLU(mix_slope) between them; same arithmetic
    // order as clica_mixing_fwd, hence the same bits), computed on this workgroup's 48 latent rows in a corner of
    // the still empty panel; x goes to the panel AND to HBM (layer 0's weight gradient reads it)
    const int n = g.layer[0].K, K16 = (n + KI - 1) & ~(KI - 1);
    float* xa = panel; float* xb = panel + ROWS * MIX_MAX_N;
    for (int idx = threadIdx.x; idx < ROWS * n; idx += THREADS) {
      const int r = idx / n, k = idx - r * n;
      xa[idx] = (r < nrows) ? g.X[(row0 + r) * g.ldx + k] : 0.f;
    }
    __syncthreads();
    for (int l = 0; l < g.mixL; ++l) {
      const float* wl = g.mixW + l * n * n;
      for (int idx = threadIdx.x; idx < ROWS * n; idx += THREADS) {
        const int r = idx / n, j = idx - r * n;
        float acc = 0.f;
        for (int k = 0; k < n; ++k) acc = fmaf(xa[r * n + k], wl[j * n + k], acc);
        if (l < g.mixL - 1) acc = acc > 0.f ? acc : acc * g.mix_slope;
        xb[idx] = acc;
      }
      __syncthreads();
      float* t = xa; xa = xb; xb = t;
    }
    float v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int idx = threadIdx.x + u * THREADS; v[u] = idx < ROWS * n ? xa[idx] : 0.f; }
    __syncthreads();
    for (int idx = threadIdx.x; idx < ROWS * K16; idx += THREADS) { const int r = idx / K16; panel[r * LDP + (idx - r * K16)] = 0.f; }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = threadIdx.x + u * THREADS;
      if (idx < ROWS * n) {
        const int r = idx / n, k = idx - r * n;
        panel[r * LDP + k] = v[u];
        if (r < nrows) g.xout[(row0 + r) * g.ldxo + k] = v[u];
      }
    }
  } else {
    const int K0 = g.layer[0].K, K16 = (K0 + KI - 1) & ~(KI - 1);
    for (int idx = threadIdx.x; idx < ROWS * K16; idx += THREADS) {
      const int r = idx / K16, k = idx - r * K16;
      panel[r * LDP + k] = (r < nrows && k < K0) ? g.X[(row0 + r) * g.ldx + k] : 0.f;
    }
  }
  __syncthreads();

  FMLP_STAMP(MAXL, 0);
  for (int l = 0; l < g.L; ++l) {
    const Layer& ly = g.layer[l];
    FMLP_STAMP(l, 0);
    f32x4 acc[RB][CBW];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int c = 0; c < CBW; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool vec = ((reinterpret_cast<uintptr_t>(ly.W) & 15) == 0) && (ly.ldw % 4 == 0) && (ly.K % 4 == 0);
    const int ncb_real = (ly.N + 15) / 16;
    int nc = (ncb_real - wave + WAVES - 1) / WAVES;        // column blocks wave, wave+8, ... below ncb_real
    nc = nc < 0 ? 0 : (nc > CBW ? CBW : nc);
    const float* pk = PACKED ? g.packed + g.pack_off[l] : nullptr;
    const unsigned long long mbits = (unsigned long long)mraw.x | ((unsigned long long)mraw.y << 32);
    float bias_l[CBW];
#pragma unroll
    for (int c = 0; c < CBW; ++c) bias_l[c] = bias[c];
#define CLICA_FMLP_DISPATCH(VECV, PACKV)                                                           \
    switch (nc) {                                                                                \
      case 4: layer_gemm<VECV, PACKV, 4>(ly, pk, panel, wave, lane, acc, bpre); break;           \
      case 3: layer_gemm<VECV, PACKV, 3>(ly, pk, panel, wave, lane, acc, bpre); break;           \
      case 2: layer_gemm<VECV, PACKV, 2>(ly, pk, panel, wave, lane, acc, bpre); break;           \
      case 1: layer_gemm<VECV, PACKV, 1>(ly, pk, panel, wave, lane, acc, bpre); break;           \
      default: break;                                                                            \
    }
    if (PACKED) { CLICA_FMLP_DISPATCH(true, true) }
    else if (vec) { CLICA_FMLP_DISPATCH(true, false) }
    else { CLICA_FMLP_DISPATCH(false, false) }
#undef CLICA_FMLP_DISPATCH
    FMLP_STAMP(l, 1);
#if CLICA_FMLP_DEFER_STORE
    // The HBM copy of the PREVIOUS layer's output (this layer's input panel, still intact until the barrier below) is
    // issued here by the waves whose k-loop ends first: the two waves of a SIMD share its matrix pipe with age-priority
    // arbitration, waves 0..3 leave the loop ~20 % earlier than waves 4..7 and would only wait at the barrier.
    if (l > 0 && wave < WAVES / 2) store_panel(panel, g.layer[l - 1], row0, nrows, (wave * 64 + lane), 64 * (WAVES / 2));
#endif
    __syncthreads();                                   // every wave is done reading the input panel
    FMLP_STAMP(l, 2);

    // Request the NEXT layer's first weight fragments and sign bits now: they do not depend on the panel, are
    // older than the activation stores below, and land while this layer's epilogue runs.
    if (l + 1 < g.L) {
      const Layer& nx = g.layer[l + 1];
      if (PACKED) request_first_b(g.packed + g.pack_off[l + 1], nx.K, nx.N, wave, lane, bpre);
      mraw = __builtin_amdgcn_raw_buffer_load_b64(mask_rsrc(nx.dact ? nx.mask_in : nullptr), mslot, 0, 0);
      request_bias(nx, bias);
    }
    __builtin_amdgcn_sched_barrier(0);

    // epilogue into the panel (see epilogue_to_panel); the slow activation-re-reading variant of the backward
    // chain (no sign bits given) is the only one that touches global memory here
    unsigned long long obits = 0ull;
    {
      const int N = ly.N;
      const float slope = g.slope;
      const bool bits = ly.mask_out != nullptr;
      if (!ly.dact) {
        if (ly.leaky) obits = bits ? epilogue_to_panel<EPI_BIAS_LEAKY, true>(panel, acc, bias_l, 0ull, slope, N, wave, lane)
                                   : epilogue_to_panel<EPI_BIAS_LEAKY, false>(panel, acc, bias_l, 0ull, slope, N, wave, lane);
        else obits = bits ? epilogue_to_panel<EPI_BIAS, true>(panel, acc, bias_l, 0ull, slope, N, wave, lane)
                          : epilogue_to_panel<EPI_BIAS, false>(panel, acc, bias_l, 0ull, slope, N, wave, lane);
      } else if (ly.mask_in) {
        epilogue_to_panel<EPI_DACT_MASK, false>(panel, acc, bias_l, mbits, slope, N, wave, lane);
      } else if (!AUX || !ly.aux) {
        epilogue_to_panel<EPI_DACT_NONE, false>(panel, acc, bias_l, 0ull, slope, N, wave, lane);
      } else {
        const int ncb = ((N + KI - 1) & ~(KI - 1)) / 16;
#pragma unroll
        for (int c = 0; c < CBW; ++c) {
          const int cb = wave + c * WAVES;
          if (cb < ncb) {
            const int col = cb * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < RB; ++r) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int row = r * 16 + (lane >> 4) * 4 + e;
                float v = acc[r][c][e];
                if (row < nrows && col < N) v *= (ly.aux[(row0 + row) * ly.ldaux + col] > 0.f ? 1.f : slope);
                panel[row * LDP + col] = col < N ? v : 0.f;
              }
            }
          }
        }
      }
    }
    __builtin_amdgcn_raw_buffer_store_b64((u32x2){(unsigned)obits, (unsigned)(obits >> 32)}, mask_rsrc(ly.mask_out), mslot, 0, 0);
    FMLP_STAMP(l, 3);
    __syncthreads();
    FMLP_STAMP(l, 4);

#if CLICA_FMLP_DEFER_STORE
    if (l + 1 < g.L) { FMLP_STAMP(l, 5); continue; }    // stored by the early waves at the end of the next layer's k-loop
#endif
    // stream the new activations to HBM straight from the panel (coalesced rows)
    const bool ovec = ((reinterpret_cast<uintptr_t>(ly.out) & 15) == 0) && (ly.ldo % 4 == 0) && (ly.N % 4 == 0);
    if (ovec) {
      // A FIXED number of unconditional raw-buffer stores (pieces outside the panel get an out-of-range offset
      // and are dropped by the hardware): the wait in front of the next layer's first MFMAs can then be an
      // exact count that leaves these stores in flight instead of draining them.  Streaming (nt) stores: the
      // saved activations are not read again in this kernel, keep them from evicting the weights out of L2.
      const int n4 = ly.N / 4;
      const __amdgpu_buffer_rsrc_t orsrc =
          __builtin_amdgcn_make_buffer_rsrc(ly.out + row0 * ly.ldo, 0, (int)(((int64_t)(nrows - 1) * ly.ldo + ly.N) * 4), kRsrcWord3);
      constexpr int PIECES = (ROWS * (MAXW / 4) + THREADS - 1) / THREADS;
#pragma unroll
      for (int it = 0; it < PIECES; ++it) {
        const int idx = threadIdx.x + it * THREADS;
        const int r = idx / n4, c4 = idx - r * n4;
        const bool in = r < nrows;
        const f32x4 v = *reinterpret_cast<const f32x4*>(&panel[(in ? r : 0) * LDP + 4 * c4]);
        const unsigned off = in ? (unsigned)((r * (int)ly.ldo + 4 * c4) * 4) : kOobOffset;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orsrc, off, 0, FP32_STORE_AUX);
      }
    } else {
      for (int idx = threadIdx.x; idx < nrows * ly.N; idx += THREADS) {
        const int r = idx / ly.N, c = idx - r * ly.N;
        ly.out[(row0 + r) * ly.ldo + c] = panel[r * LDP + c];
      }
    }
    // no barrier needed here: the next layer only READS the panel until its own post-GEMM barrier
    FMLP_STAMP(l, 5);
  }
}
#ifdef CLICA_FMLP_TRACE
extern "C" int clica_debug_fmlp_trace(unsigned long long* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(clica::fmlp::g_trace), &buf, sizeof(buf));
}
#endif

// ---- weight packing: nn.Linear layout -> MFMA fragment order ----------------------------------------------
// packed[layer][cb][ki][h][lane] (float4) = W[cb*16 + (lane&15)][ki*32 + h*16 + 4*(lane>>4) .. +3], zero padded.
// A wave's B-fragment load for (cb, ki, h) is then ONE fully contiguous 1 KB request instead of sixteen 64-byte
// pieces of sixteen different weight rows (which made the fused forward texture-addresser bound).
static inline int64_t pack_float4s(int N, int K) { return (int64_t)((N + 15) / 16) * ((K + KI - 1) / KI) * 2 * 64; }

// One thread per DESTINATION float4: a wave writes one whole 1 KB fragment (fully coalesced); its reads are sixteen
// 64-byte row pieces (plain layout) or 64-byte column pieces of four rows (transposed layout) of an L2-resident
// matrix.  (The first version went source-parallel: coalesced reads but 213 K scattered 16-byte writes, 11 us.)
// Padding entries (rows >= N of the last column block, k >= K of the last k-iteration) are written as zeros.
constexpr int MAXSEG = 2 * MAXL;
struct PackSeg { const float* W; int64_t ldw; int rows, cols, transposed; };   // logical matrix [rows][cols] = W or W^T
struct PackArgs {
  int nseg;
  int64_t first[MAXSEG + 1];     // first destination float4 of each segment (global numbering over both buffers)
  float4* dst[MAXSEG];           // destination of the segment's first float4
  PackSeg seg[MAXSEG];
};

__global__ __launch_bounds__(256) void mlp_pack_k(PackArgs a) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.first[a.nseg]) return;
  int sidx = 0;
#pragma unroll
  for (int i = 1; i < MAXSEG; ++i) sidx += (i < a.nseg && idx >= a.first[i]) ? 1 : 0;
  const PackSeg& sg = a.seg[sidx];
  const unsigned d = (unsigned)(idx - a.first[sidx]);
  const unsigned kiters = (unsigned)(sg.cols + KI - 1) / KI;
  const unsigned lane = d & 63u, h = (d >> 6) & 1u, frag = d >> 7;          // frag = cb * kiters + ki
  const unsigned cb = frag / kiters, ki = frag - cb * kiters;
  const int n = (int)(cb * 16u + (lane & 15u)), k = (int)(ki * KI + h * 16u + 4u * (lane >> 4));
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (n < sg.rows) {
    if (!sg.transposed) {
      const float* row = sg.W + (int64_t)n * sg.ldw + k;
      if (k + 3 < sg.cols && ((reinterpret_cast<uintptr_t>(row) & 15) == 0)) {
        const float4 t = *reinterpret_cast<const float4*>(row);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) if (k + u < sg.cols) v[u] = row[u];
      }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) if (k + u < sg.cols) v[u] = sg.W[(int64_t)(k + u) * sg.ldw + n];
    }
  }
  a.dst[sidx][d] = make_float4(v[0], v[1], v[2], v[3]);
}


// =====================================================================================================================
// Split-bf16 variant of the same whole-stack kernel (opt-in, clica_mlp_*_split): fp32-grade results on the bf16
// matrix cores.  Every fp32 operand is split EXACTLY into three bf16 pieces (8 + 8 + 8 mantissa bits, by
// truncation: v = hi + mid + lo with no rounding), and the six piece products of order <= 2 -- lo.hi, hi.lo,
// mid.mid, mid.hi, hi.mid, hi.hi -- are accumulated in fp32 by v_mfma_f32_16x16x32_bf16.  Each bf16 x bf16 product
// is exact in fp32; the dropped products are below 2^-24 of |a||b|.  Measured against fp64 on a 500 x 500 layer over
// 12 288 rows: max error 8.6e-7 of max|y| (the fp32-MFMA kernel: 1.0e-6).  Six bf16 instructions of 16 cycles cover
// K = 32 where the fp32 path needs eight of 32 cycles: 0.375 of the matrix time (tools/proto/bf16x3_probe.py:
// 26 vs 50 us marginal per 500 x 500 layer; the weight stream, now 6 B per weight from L2, is what remains).
//   * panel: three bf16 planes [48][520] in LDS (150 KB);
//   * weights: fragment order per piece, one contiguous 1 KB request per (piece, column block, k-iteration);
//   * the product is taken transposed (matrix rows = output features, columns = batch rows): a lane then owns
//     FOUR CONSECUTIVE FEATURES of one batch row, so the epilogue emits one float4 to HBM and one 8-byte LDS
//     write per plane instead of element-wise traffic, and there is no panel -> HBM copy phase.
// =====================================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#ifndef CLICA_SPLIT_PINGPONG
#define CLICA_SPLIT_PINGPONG 1
#endif
#ifndef CLICA_SPLIT_LOAD_GAP
#define CLICA_SPLIT_LOAD_GAP 4
#endif
#ifndef CLICA_SPLIT_BALANCE
#define CLICA_SPLIT_BALANCE 0     // 1: keep the two waves of a SIMD in step (LDS progress words + s_setprio); measured, no gain -- see layer_gemm_split
#endif
#ifndef CLICA_SPLIT_FAST_EPI
#define CLICA_SPLIT_FAST_EPI 1
#endif
constexpr int LDPB = MAXW + 8;                  // bf16 elements per panel row: 1040 B, 16-byte aligned, bank-staggered
constexpr int PLANE = ROWS * LDPB;              // bf16 elements per plane

__device__ __forceinline__ void split3(float v, unsigned& hb, unsigned& mb, unsigned& lb) {     // piece bits in the HIGH half
  hb = __float_as_uint(v) & 0xFFFF0000u;
  const float r1 = v - __uint_as_float(hb);
  mb = __float_as_uint(r1) & 0xFFFF0000u;
  const float r2 = r1 - __uint_as_float(mb);
  lb = __float_as_uint(r2) & 0xFFFF0000u;       // <= 8 significant bits left: exact
}

static inline int64_t pack3_entries(int N, int K) { return (int64_t)((N + 15) / 16) * ((K + KI - 1) / KI) * 64; }   // 16-byte entries per piece

// ---- the two split arithmetics of mlp_split_k ------------------------------------------------------------------------------------
// AR = 0  "bf16x3": v = hi + mid + lo exactly (3 bf16 pieces by truncation), the six piece products of order <= 2; operands keep their
//         fp32 exponent range; 6 B per element; 6 MFMAs of K = 32 per fp32 product block.
// AR = 1  "f16x2" (round 5): v s = hi + lo with hi = RN_f16(v s), lo = RN_f16(v s - hi) (22 significand bits: the Ootomo-Yokota
//         construction with an UNSCALED second piece -- see below), the three products hi.hi, hi.lo, lo.hi (lo.lo < 2^-22 |a||b| dropped);
//         4 B per element; 3 MFMAs per block at the same MFMA rate: half the matrix work and two thirds of the operand bytes of AR = 0.
//         fp16 has a 5-bit exponent, so every TENSOR carries a power-of-two scale s that puts its largest magnitude into
//         [2^kF16Target, 2^(kF16Target+1)); elements keep 22 bits down to 2^-3 (where lo turns subnormal) and an ABSOLUTE error of
//         2^-25 below that, i.e. <= 2^-33 of the tensor's maximum: norm-wise the fp32 accumulation (2^-24 sum |a||b|) dominates, as in
//         AR = 0 (measured: 6.2e-7 of max|y| against fp64 on a 500 x 500 layer over 12 288 rows; native fp32 MFMA 7.2e-7, bf16x3 4.6e-7).
//         The scale in force during a launch is the one derived from the PREVIOUS launch's maximum of the same tensor (Split16 state,
//         clica_split16_*): every producer records max |v| as it writes, a one-workgroup update turns the maxima into the next scales.
//         A tensor whose maximum grows by more than 2^(15-kF16Target-1) = 64 x between two consecutive steps would overflow fp16: the
//         producers detect it (scaled magnitude > kF16Alarm) and raise a sticky device flag that the host checks at its sync points.
template <int AR> struct Arith;
template <> struct Arith<0> {
  static constexpr int NP = 3, NPROD = 6;
  static constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};      // (weight piece, activation piece), small terms first
  static constexpr int XORDER[3] = {0, 2, 1};                                         // order in which the products first need the activation pieces
};
template <> struct Arith<1> {
  static constexpr int NP = 2, NPROD = 3;
  static constexpr int PW[3] = {1, 0, 0}, PX[3] = {0, 1, 0};
  static constexpr int XORDER[2] = {0, 1};
};
using s16::kF16Target; using s16::kF16Alarm; using s16::Split16State; using s16::kS16CapWG; using s16::kS16CapPW;
using s16::s16_partA; using s16::s16_partD; using s16::s16_partW; using s16::split16_update_tensor;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <int AR>
__device__ __forceinline__ f32x4 mfma16(const u32x4 w, const u32x4 x, const f32x4 acc) {
  if constexpr (AR == 0) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), acc, 0, 0, 0);
}
typedef float f32x2s __attribute__((ext_vector_type(2)));
// two scaled fp32 values -> packed hi pieces and packed lo pieces (5 instructions: cvt_pk, two cvt back, packed subtract, cvt_pk)
__device__ __forceinline__ void split16_pair(const f32x2s t, unsigned& hi, unsigned& lo) {
  const f16x2 h = __builtin_convertvector(t, f16x2);
  const f32x2s r = t - __builtin_convertvector(h, f32x2s);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}
__device__ __forceinline__ void split16_one(const float t, unsigned short& hi, unsigned short& lo) {
  const _Float16 h = (_Float16)t;
  hi = __builtin_bit_cast(unsigned short, h);
  lo = __builtin_bit_cast(unsigned short, (_Float16)(t - (float)h));
}
// one call per wave and tensor: the wave's maximum (true units; NaN / inf as a huge value so that the update kernel sees them) into an LDS word
__device__ __forceinline__ void amax_wave_to_lds(unsigned* lds_word, float m_scaled, float inv_scale) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m_scaled = fmaxf(m_scaled, __shfl_xor(m_scaled, off, 64));
  if ((threadIdx.x & 63) == 0) {
    const float t = (m_scaled <= 3.0e38f) ? m_scaled * inv_scale : 3.4e38f;
    atomicMax(lds_word, __float_as_uint(t));
  }
}

// fragment-order pieces: dst[piece][cb][ki][lane] (16 B = 8 bf16) = piece(W[cb*16 + (lane&15)][ki*32 + (lane>>4)*8 .. +7])
struct Pack3Args {
  int nseg;
  int64_t first[MAXSEG + 1];
  u32x4* dst[MAXSEG];          // piece 0 of the segment; pieces are `entries` apart
  int64_t entries[MAXSEG];
  PackSeg seg[MAXSEG];
};
__global__ __launch_bounds__(256) void mlp_pack3_k(Pack3Args a) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.first[a.nseg]) return;
  int sidx = 0;
#pragma unroll
  for (int i = 1; i < MAXSEG; ++i) sidx += (i < a.nseg && idx >= a.first[i]) ? 1 : 0;
  const PackSeg& sg = a.seg[sidx];
  const unsigned d = (unsigned)(idx - a.first[sidx]);
  const unsigned kiters = (unsigned)(sg.cols + KI - 1) / KI;
  const unsigned lane = d & 63u, frag = d >> 6;
  const unsigned cb = frag / kiters, ki = frag - cb * kiters;
  const int n = (int)(cb * 16u + (lane & 15u)), k = (int)(ki * KI + 8u * (lane >> 4));
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    float v = 0.f;
    if (n < sg.rows && k + u < sg.cols) v = sg.transposed ? sg.W[(int64_t)(k + u) * sg.ldw + n] : sg.W[(int64_t)n * sg.ldw + k + u];
    split3(v, h[u], m[u], l[u]);
  }
  auto pk = [](const unsigned (&b)[8]) {
    return (u32x4){(b[0] >> 16) | b[1], (b[2] >> 16) | b[3], (b[4] >> 16) | b[5], (b[6] >> 16) | b[7]};
  };
  u32x4* dst = a.dst[sidx] + d;
  dst[0] = pk(h); dst[a.entries[sidx]] = pk(m); dst[2 * a.entries[sidx]] = pk(l);
}

// f16x2 weights: the same fragment order with two pieces, scaled by the layer's weight scale in force (Split16State::sW); the
// forward-orientation segments also record max |W| of their layer for the next scale update.
struct Pack2Args {
  int nseg;
  int64_t first[MAXSEG + 1];
  u32x4* dst[MAXSEG];
  int64_t entries[MAXSEG];
  PackSeg seg[MAXSEG];
  const float* scale[MAXSEG];      // device: the scale of this segment's tensor
  int layer[MAXSEG];               // forward-orientation segments: the layer whose max |W| this segment records; -1: none
  int nfwd;                        // number of forward-orientation segments = layers (they come first)
  Split16State* st;
};
__device__ __forceinline__ void pack2_block(const Pack2Args& a, const unsigned block) {
  const int64_t idx = (int64_t)block * 256 + threadIdx.x;
  const bool live = idx < a.first[a.nseg];
  // (no early return: whole waves take part in the maximum; a wave never straddles two segments' amax slots because segment
  //  sizes are multiples of 64 entries)
  int sidx = 0;
#pragma unroll
  for (int i = 1; i < MAXSEG; ++i) sidx += (i < a.nseg && idx >= a.first[i]) ? 1 : 0;
  const PackSeg& sg = a.seg[sidx];
  const unsigned d = live ? (unsigned)(idx - a.first[sidx]) : 0u;
  const unsigned kiters = (unsigned)(sg.cols + KI - 1) / KI;
  const unsigned lane = d & 63u, frag = d >> 6;
  const unsigned cb = frag / kiters, ki = frag - cb * kiters;
  const int n = (int)(cb * 16u + (lane & 15u)), k = (int)(ki * KI + 8u * (lane >> 4));
  const float sc = *a.scale[sidx];
  unsigned hw[4], lw[4];
  float m = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    f32x2s v = {0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int kk = k + 2 * u + e;
      if (live && n < sg.rows && kk < sg.cols) v[e] = sc * (sg.transposed ? sg.W[(int64_t)kk * sg.ldw + n] : sg.W[(int64_t)n * sg.ldw + kk]);
    }
    m = fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1])));
    split16_pair(v, hw[u], lw[u]);
  }
  if (live) {
    u32x4* dst = a.dst[sidx] + d;
    dst[0] = (u32x4){hw[0], hw[1], hw[2], hw[3]};
    dst[a.entries[sidx]] = (u32x4){lw[0], lw[1], lw[2], lw[3]};
  }
  // this wave's slot: max |W| (true units) and the layer it belongs to
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  const unsigned wid = (unsigned)(idx >> 6);
  if ((threadIdx.x & 63) == 0 && wid < kS16CapPW && a.layer[sidx] >= 0) s16_partW(a.st)[wid] = __float_as_uint((m <= 3.0e38f) ? m / sc : 3.4e38f);
  if ((threadIdx.x & 63) == 0 && !(m <= kF16Alarm)) s16::s16_raise_poison(a.st);      // the guard: a weight that does not fit its layer's scale (parameters set from outside)
  if (block == 0 && threadIdx.x == 0) s16::s16_refresh_gen(a.st);
  if (block == 0 && threadIdx.x <= Split16State::NT) {      // wave ranges of the layers (forward-orientation segments come first, one per layer)
    const int l = threadIdx.x;
    a.st->wfirst[l] = (unsigned)(a.first[l < a.nfwd ? l : a.nfwd] >> 6);
    if (l == 0) a.st->nPW = (unsigned)(a.first[a.nfwd] >> 6);
  }
}
__global__ __launch_bounds__(256) void mlp_pack2_k(Pack2Args a) { pack2_block(a, blockIdx.x); }
// The training step's two independent front launches in one -- the weight pack (blocks [0, pack_blocks))
// and the latent pair draw z, z~ (the blocks behind them; rng::sample_pair_elem, the body of sampler.hip's sample_pair_elem_k: same Philox
// counters, same numbers).  As two launches in line they cost 12 + 9 us in front of the forward; forked onto two streams the HIP graph's
// fork / join cost more than it hid (engine.py: _step_body).
static_assert(rng::THREADS == 256, "the merged launch uses one block size for both bodies");
__global__ __launch_bounds__(256) void mlp_pack2_sample_k(Pack2Args a, rng::PairArgs s, unsigned pack_blocks) {
  if (blockIdx.x >= pack_blocks) {
    rng::sample_pair_elem(s, (int64_t)(blockIdx.x - pack_blocks) * 256 + threadIdx.x);
    return;
  }
  pack2_block(a, blockIdx.x);
}

__global__ __launch_bounds__(256) void split16_update_k(Split16State* st, int L) { split16_update_tensor(st, L, (int)blockIdx.x); }
__global__ __launch_bounds__(64) void split16_init_k(Split16State* st, unsigned capWG, unsigned capPW) {
  const int t = threadIdx.x;
  if (t < Split16State::NT) st->sA[t] = st->sD[t] = st->sW[t] = st->sWC[t] = st->pA[t] = st->pD[t] = 1.f;
  if (t < Split16State::NT) st->cntA[t] = st->cntD[t] = st->cntW[t] = 0u;
  if (t == 0) {
    st->flags = 0u; st->updates = 0u; st->nPW = 0u; st->capWG = capWG; st->capPW = capPW;
    st->gen_copy = 0xFFFFFFFEu; st->poison = 0xFFFFFFFFu; st->skipped = 0u; st->dp_poison = nullptr;      // the guard (split16.h): no step poisoned
  }
}

// Debug build only (-DCLICA_SPLIT_TRACE, tools/split_trace.py): s_memtime stamps per (workgroup, wave, layer, phase), kept in
// registers and written once at the end of the kernel
#ifdef CLICA_SPLIT_TRACE
// (round 6: the stamps live in LDS behind the kernel's own words -- as a dynamically indexed private array they put the trace build on
//  scratch memory, 528 B per lane, which the product kernel does not use)
__device__ unsigned long long* g_strace = nullptr;
#define ST_NS 8
#define ST_LDS_BYTES (WAVES * MAXL * ST_NS * 8)
#define ST_DECL unsigned long long* st_lds = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(planes) + (size_t)Arith<AR>::NP * PLANE * 2 + (size_t)a.boff[a.g.L] * 4 + 256); \
  for (int i_ = lane; i_ < MAXL * ST_NS; i_ += 64) st_lds[wave * MAXL * ST_NS + i_] = 0ull
#define ST_STAMP(l, ph) do { if (lane_id == 0) st_lds[(wave * MAXL + (l)) * ST_NS + (ph)] = __builtin_readcyclecounter(); } while (0)
#define ST_FLUSH(L) do { if (g_strace) { for (int i_ = lane_id; i_ < MAXL * ST_NS; i_ += 64) \
      g_strace[((size_t)blockIdx.x * WAVES + wave) * MAXL * ST_NS + i_] = st_lds[wave * MAXL * ST_NS + i_]; } } while (0)
#else
#define ST_LDS_BYTES 0
#define ST_DECL do { } while (0)
#define ST_STAMP(l, ph) do { } while (0)
#define ST_FLUSH(L) do { } while (0)
#endif

struct SplitArgs {
  Args g;                        // same description of the stack as the fp32 kernel (g.packed unused)
  const u32x4* packed3;
  int64_t off3[MAXL];            // 16-byte entry offset of each layer's piece 0
  int64_t ent3[MAXL];            // entries per piece of each layer
  int boff[MAXL + 1];            // float offset of each layer's bias row inside the LDS bias table (rows padded to KI)
  int warm_next;                 // L2 warm-up of the layer that follows a wide one: at most this many KB per early wave (see mlp_split_k)
  // f16x2 arithmetic only (Arith<1>): scales in force and where to record the maxima, in THIS launch's order --
  // s_t[0]: the launch's input tensor, s_t[l + 1]: output of layer l; s_w[l]: weights of layer l; part_t[workgroup][NT]: this launch's
  // slot array for the maxima (same positions), count_t: where workgroup 0 leaves the number of slots written.
  // `last_unscaled`: the last layer's output is not re-split (fp32 consumer only): its scale is 1 whatever s_t[L] says.
  const float* s_t; const float* s_w; unsigned* part_t; unsigned* count_t; unsigned cap_wg; int last_unscaled;
  Split16State* st16;            // the guard (split16.h): where a workgroup announces a tensor that outgrew its scale
  // Backward chain of a training step (clica_mlp_dgrad_split_tail): behind its last link every workgroup also leaves the partial weight
  // gradients of the encoder's n-wide FIRST and LAST layer over its 48 rows (fp32 vector ALU, as wgrad_tiny_k computes them), one slab
  // per workgroup in the weight-gradient workspace -- the separate wgrad_tiny_k launch (12.8 us, latency-bound) is gone from the step.
  //   last layer:  dW = dzl^T al  [n_l][w_l], db = column sums of dzl;   first layer: dW = dz0^T x  [w_0][n_0], db = column sums of dz0
  struct Tail {
    const float* dzl; int64_t ld_dzl; const float* al; int64_t ld_al; const float* dz0; int64_t ld_dz0; const float* x; int64_t ld_x;
    int n_l, w_l, w_0, n_0;                  // n_l <= 16, n_0 <= 15, w_l <= 127, w_0 <= 128 (clica_mlp_chain_tail_supported)
    float* slab_l; float* dbslab_l; float* slab_0; float* dbslab_0;      // [workgroups][...]; slab_l == nullptr: no tail
  } tail;
  // The launch's input still lacks the loss's pair-sweep partials (clica_lp_dy_parts, include/clica.h): the prologue adds them -- the
  // sums bwd_reduce_k (lp_loss.hip) would have formed, in its order -- writes the finished rows back for the tail, and workgroup 0 leaves
  // the forward's means and ticks the step counter.  part == nullptr: plain input.
  struct DyParts {
    const float* part; int nsplit, nsplit_alt, np, n; long long rows; const float* words; float limit;
    const float* blocksums; int nblocks; float inv_count; float* means; int* tick; float* dy; long long ldy;
  } parts;
  // Round 4: what the training step's epilogues need of a layer, in ONE 64-byte record (one s_load_dwordx16 at the top of the layer).
  // Reading the same facts field by field from g.layer[l] behind the k-loop was a chain of ~10 DEPENDENT scalar loads, each with its
  // own s_waitcnt lgkmcnt(0) (flag -> branch -> next flag) in front of the epilogue, with the matrix pipe idle (tools/split_trace.py).
  struct alignas(64) LayerQ {
    unsigned short* planes; unsigned long long* mask_out;
    int N, pl_units, pl_ones, fast_kind;          // fast_kind: 0 generic epilogue, 1 forward hidden layer, 3 backward link (split_epilogue_fast)
    int boff, K; long long off3, ent3;
    int pad[2];
  } q[MAXL];
};
#ifndef CLICA_SPLIT_WARM_NEXT
#define CLICA_SPLIT_WARM_NEXT 12                 // 12: also a 500 x 500 layer that follows directly (1.5 MB per XCD); 3: short layers only
#endif
constexpr int WARM_LOADS = CLICA_SPLIT_WARM_NEXT;   // L2 warm-up: 1 KB slices of the NEXT layer's weights per participating wave (see mlp_split_k)
constexpr int WARM_LOADS2 = 3;                      // ... of the layer after it (short layers only)
constexpr int BIAS_LDS_MAX = (160 * 1024 - 3 * PLANE * 2) / 4 - 64;    // floats left beside the three planes (the f16x2 kernel has a plane to spare)

// Narrow layers (ONE column block per wave: the 100- and 10-wide layers).  The wide loop below keeps ONE k-iteration of
// weights in flight, enough when 72 MFMAs (1 152+ cycles) cover an L2 miss.  Here an iteration is 18 or 36 MFMAs, and in the
// training step the packed weights are COLD (mlp_pack3_k rewrote them a moment ago on another XCD: the first touch per XCD comes
// from HBM): the trace showed the 500 -> 100 layer at 29.9 k cycles for 16 iterations cold against 15.8 k warm, i.e. one exposed
// memory round trip per iteration.  So: a ring of FOUR weight sets (three iterations in flight, the registers the wide loop
// spends on its four column blocks) and two activation-fragment sets (the LDS reads of iteration i + 1 under the MFMAs of i).
template <int NC, int AR>
__device__ __forceinline__ void layer_gemm_split_narrow(const int K, const u32x4* __restrict__ w0, const int64_t ent, const unsigned short* planes,
                                                        int wave, int lane, f32x4 (&acc)[RB][CBW], const u32x4 (&wpre)[Arith<AR>::NP][CBW]) {
  constexpr int NP = Arith<AR>::NP, NPROD = Arith<AR>::NPROD;
  static_assert(NC == 1, "narrow variant: with two column blocks the ring (96) + both fragment sets (72) + accumulators spill");
  const int i15 = lane & 15, kg = lane >> 4;
  const int kiters = (K + KI - 1) / KI;
  const unsigned lane16 = (unsigned)lane * 16u;
  u32x4 w[4][NP][NC];
  u32x4 x[2][NP][RB];
  auto kof = [&](int ki) { return ki < kiters ? ki : kiters - 1; };      // past the end: a harmless re-read of the last iteration's operands
  auto fetch_w = [&](u32x4 (&d)[NP][NC], int ki) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const char* base = reinterpret_cast<const char*>(w0 + p * ent + ((int64_t)(wave + c * WAVES) * kiters + ki) * 64);
        d[p][c] = *reinterpret_cast<const u32x4*>(base + lane16);
      }
  };
  auto fetch_x = [&](u32x4 (&d)[NP][RB], int ki) {
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
      for (int r = 0; r < RB; ++r)
        d[Arith<AR>::XORDER[q]][r] = *reinterpret_cast<const u32x4*>(&planes[Arith<AR>::XORDER[q] * PLANE + (r * 16 + i15) * LDPB + ki * KI + kg * 8]);
  };
  auto mma = [&](const u32x4 (&ww)[NP][NC], const u32x4 (&xx)[NP][RB]) {
#pragma unroll
    for (int t = 0; t < NPROD; ++t)
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < RB; ++r)
          acc[r][c] = mfma16<AR>(ww[Arith<AR>::PW[t]][c], xx[Arith<AR>::PX[t]][r], acc[r][c]);
  };
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int c = 0; c < NC; ++c) w[0][p][c] = wpre[p][c];
  fetch_w(w[1], kof(1));
  fetch_w(w[2], kof(2));
  fetch_x(x[0], 0);
  for (int ki = 0; ki < kiters; ki += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (ki + u < kiters) {                           // wave-uniform
        fetch_w(w[(u + 3) & 3], kof(ki + u + 3));
        fetch_x(x[(u + 1) & 1], kof(ki + u + 1));
        __builtin_amdgcn_sched_barrier(0);
        mma(w[u], x[u & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): the dead look-ahead requests (see layer_gemm_split)
}

template <int NC, int AR>
__device__ __forceinline__ void layer_gemm_split(const int K, const u32x4* __restrict__ w0, const int64_t ent, const unsigned short* planes,
                                                 int wave, int lane, f32x4 (&acc)[RB][CBW], const u32x4 (&wpre)[Arith<AR>::NP][CBW], volatile int* prog) {
  constexpr int NP = Arith<AR>::NP, NPROD = Arith<AR>::NPROD;
  const int i15 = lane & 15, kg = lane >> 4;
  const int kiters = (K + KI - 1) / KI;
  u32x4 wcur[NP][CBW], wnxt[NP][CBW];
  const unsigned lane16 = (unsigned)lane * 16u;
  auto fetch_w = [&](u32x4 (&w)[NP][CBW], int ki) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int c = 0; c < NC; ++c) {     // wave-uniform 64-bit base (scalar registers) + 32-bit lane offset: no per-load VGPR address pair
        const char* base = reinterpret_cast<const char*>(w0 + p * ent + ((int64_t)(wave + c * WAVES) * kiters + ki) * 64);
        w[p][c] = *reinterpret_cast<const u32x4*>(base + lane16);
      }
  };
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int c = 0; c < NC; ++c) wcur[p][c] = wpre[p][c];
  // Ping-pong form (round 3; the fp32 kernel's k-loop got the same treatment in round 2): two weight sets AND two activation
  // fragment sets with swapped roles in a loop unrolled by two -- no rotation copies (the 12 x 4 v_mov per iteration above),
  // the activation fragments of iteration ki + 1 are read from the panel during the MFMAs of ki instead of in front of them,
  // and sched_group_barrier pins one weight request behind every third MFMA of the first half of the block and one panel read
  // behind every NC-th MFMA of the second half (requests bunched at the head of the block keep the wave off the matrix pipe).
  // (Two activation-fragment sets as well -- reading iteration ki + 1's fragments during the MFMAs of ki -- does not fit: 96 + 72 +
  //  48 registers plus the epilogue's live values spill 71 VGPRs.  The fragments stay single-buffered, read at the head of the
  //  block in the order hi, lo, mid = the order the six products first need them.)
  auto fetch_x = [&](u32x4 (&x)[NP][RB], int ki) {
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
      for (int r = 0; r < RB; ++r)
        x[Arith<AR>::XORDER[q]][r] = *reinterpret_cast<const u32x4*>(&planes[Arith<AR>::XORDER[q] * PLANE + (r * 16 + i15) * LDPB + ki * KI + kg * 8]);
  };
  auto mma = [&](const u32x4 (&w)[NP][CBW], const u32x4 (&x)[NP][RB]) {
#pragma unroll
    for (int t = 0; t < NPROD; ++t)
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < RB; ++r)
          acc[r][c] = mfma16<AR>(w[Arith<AR>::PW[t]][c], x[Arith<AR>::PX[t]][r], acc[r][c]);
  };
  auto pin = [&]() {        // one weight request behind every CLICA_SPLIT_LOAD_GAP-th MFMA, the rest of the MFMAs behind the last one
#pragma unroll
    for (int i = 0; i < NP * NC; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, CLICA_SPLIT_LOAD_GAP, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read
    }
  };
  auto kof = [&](int ki) { return ki < kiters ? ki : kiters - 1; };      // past the end: a harmless re-read of the last iteration's operands
  u32x4 x[NP][RB];
  int ki = 0;
  for (; ki + 1 < kiters; ki += 2) {
#if CLICA_SPLIT_BALANCE
    // Round 4: keep the two waves of a SIMD in step.  The issue arbiter prefers the OLDER wave whenever both want the matrix pipe;
    // the older wave of each pair left a 500 x 500 layer's k-loop after ~25.8 k cycles, the younger ran the remaining ~17 k cycles
    // alone at 68 % of the pipe's rate (one wave cannot cover its own weight-stream latency): 42-43 k cycles for 36.9 k of matrix
    // work.  Each wave publishes its progress in LDS and the one that is BEHIND its partner (wave ^ 4: same SIMD) raises its priority.
    // MEASURED (tools/split_trace.py): the pair then runs 39.7 k / 41.4 k cycles -- in step, but no faster than the unbalanced
    // younger wave (43.3 k): two waves together reach ~0.89 of the pipe's rate, which is what the CU's 64 B/clk vector-memory path
    // leaves when the weight stream needs 43 of the ~51 B/clk it delivers (tools/l2_stream_bench.hip); and the early waves' idle
    // window (L2 warm-up, next layer's requests) is gone.  Launch time unchanged (127.8 vs 127.4 us): off by default.
    const int partner = __builtin_amdgcn_readfirstlane(prog[wave ^ 4]);
    prog[wave] = ki + 2;
    if (partner > ki) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
#endif
    fetch_x(x, ki);
    fetch_w(wnxt, kof(ki + 1));
    mma(wcur, x);
    pin();
    __builtin_amdgcn_sched_barrier(0);
    fetch_x(x, ki + 1);
    fetch_w(wcur, kof(ki + 2));
    mma(wnxt, x);
    pin();
    __builtin_amdgcn_sched_barrier(0);
  }
#if CLICA_SPLIT_BALANCE
  __builtin_amdgcn_s_setprio(0);
#endif
  if (ki < kiters) { fetch_x(x, ki); mma(wcur, x); }
  // The last pass's look-ahead request (kof) is dead but still counted: without this the compiler protects its target
  // registers with an s_waitcnt vmcnt(0) wherever the epilogue first reuses one of them -- in the middle of the epilogue's
  // stores, behind the next layer's weight requests.  Here it only waits for loads issued a whole MFMA block ago.
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt / lgkmcnt untouched
}

template <int NP>
__device__ __forceinline__ void request_first_w3(const u32x4* __restrict__ w0, int64_t ent, int K, int N, int wave, int lane, u32x4 (&w)[NP][CBW]) {
  const int kiters = (K + KI - 1) / KI, ncb_real = (N + 15) / 16;
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int c = 0; c < CBW; ++c) {
      const int cb = wave + c * WAVES;
      w[p][c] = w0[p * ent + ((int64_t)(cb < ncb_real ? cb : 0) * kiters) * 64 + lane];
    }
}

// ---- epilogue of one layer of mlp_split_k, specialised at compile time ------------------------------------------------------
// The generic epilogue inside the kernel decides per element between bias / LeakyReLU / sign-bit derivative, sign-bit
// collection, plane copy, fp32 copy (vector or scalar) ... at run time: 3 900 instructions for twelve accumulator blocks, a
// third of them scalar branches and kernarg re-loads, 15.8 k cycles per 500 x 500 layer next to a 43 k-cycle k-loop
// (tools/split_trace.py).  The combinations the training step actually runs are instantiated here without a single
// data-dependent branch; everything else (odd alignments, no sign bits, slope outside (0, 1), debug copies) keeps the generic code.
//   ACT: 1 = bias + LeakyReLU (forward, hidden layer)   2 = x LeakyReLU'(sign bits)  (backward chain)
//   BITS: collect the (y > 0) bits of this layer's output;  PL: write the bf16-plane copy;  OUTV: write the fp32 copy (16-byte stores)
__device__ __forceinline__ unsigned pack_hi(unsigned a, unsigned b) {     // {a.hi16, b.hi16} -> one dword (a in the low half)
  return __builtin_amdgcn_perm(b, a, 0x07060302u);
}
// Round 4: rewritten around what one instruction can do for TWO values (ISA count: ~63 -> ~33 vector instructions per accumulator
// block of four values): bias add, slope product and the first residual subtraction of the exact 3-way split as packed fp32
// (v_pk_add_f32 / v_pk_mul_f32, the subtraction through the neg modifier); the bf16 pieces are cut out of the UNMASKED words by
// the packing v_perm_b32 (only the subtrahends need the mask); the sign bit of a value enters its lane register as the carry of
// v_addc_co_u32 (compare + add-with-carry instead of compare + select + shift-or); the backward gate is v_bfe_i32 (bit -> 0 / ~0)
// + v_bfi_b32 (bitwise select of t and slope t) instead of and + compare + select; the constant-1 feature of the plane copy is
// patched by a 2-byte store behind the block's own 8-byte store (same lane, program order) instead of four compare + select
// pairs per block in the common path.  (Worth 1 % of the launch: the epilogue is not bound by its vector instructions, see DESIGN.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef CLICA_EPI_ABLATE
#define CLICA_EPI_ABLATE 0      // measurement builds only (tools/ab.sh): 1 = no plane stores to HBM in the fast epilogues (WRONG results)
#endif
__device__ __forceinline__ void shift_in_positive(unsigned& bits, float t) {      // bits = 2 * bits + (t > 0)
  asm("v_cmp_lt_f32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(t) : "vcc");
}
template <int ACT, bool BITS, bool PL, bool OUTV>
__device__ __forceinline__ void split_epilogue_fast(const f32x4 (&acc)[RB][CBW], unsigned short* planes, const float* bias_row, const float slope,
                                                    const unsigned long long mbits, const int N, const int ncb, const int wave, const int lane,
                                                    const __amdgpu_buffer_rsrc_t orsrc, const int ldo, const int nrows,
                                                    const __amdgpu_buffer_rsrc_t prsrc, const int pl_group_bytes, const int pl_ones,
                                                    unsigned& lo_bits, unsigned& hi_bits) {
  const int i15 = lane & 15, kg = lane >> 4;
  const unsigned pl_lane = (unsigned)((i15 >> 2) * 256 + (i15 & 3) * 32 + kg * 8);
  const f32x2 slope2 = {slope, slope};
  const int mlo = (int)(unsigned)mbits, mhi = (int)(unsigned)(mbits >> 32);
  unsigned lo = 0u, hi = 0u;
#pragma unroll
  for (int c = 0; c < CBW; ++c) {
    const int cb = wave + c * WAVES;
    if (cb < ncb) {                                    // wave-uniform
      const int n0 = cb * 16 + kg * 4;
      f32x2 b01 = {0.f, 0.f}, b23 = {0.f, 0.f};
      if (ACT == 1) { const f32x4 b4 = *reinterpret_cast<const f32x4*>(&bias_row[n0]); b01 = (f32x2){b4[0], b4[1]}; b23 = (f32x2){b4[2], b4[3]}; }
      const unsigned pl_cb = (unsigned)((cb >> 1) * 3 * 1024 + (cb & 1) * 128) + pl_lane;
      unsigned short* const dst0 = planes + i15 * LDPB + n0;
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int row = r * 16 + i15;
        const f32x4 v = acc[r][c];
        f32x2 t[2] = {(f32x2){v[0], v[1]}, (f32x2){v[2], v[3]}};
        if (ACT == 1) {
          t[0] += b01; t[1] += b23;
          const f32x2 s0 = t[0] * slope2, s1 = t[1] * slope2;
          t[0] = (f32x2){fmaxf(t[0][0], s0[0]), fmaxf(t[0][1], s0[1])};
          t[1] = (f32x2){fmaxf(t[1][0], s1[0]), fmaxf(t[1][1], s1[1])};
        }
        if (ACT == 2) {
          const f32x2 s[2] = {t[0] * slope2, t[1] * slope2};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int bit = (c * RB + r) * 4 + e;      // compile time
            // (inline asm: written as C the optimiser canonicalises the pair back into and + compare + select)
            int sel; float tv = t[e >> 1][e & 1];
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(sel) : "v"(bit < 32 ? mlo : mhi), "n"(bit & 31));      // 0 or ~0
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(tv) : "v"(sel), "v"(tv), "v"(s[e >> 1][e & 1]));       // bit ? t : slope t
            t[e >> 1][e & 1] = tv;
          }
        }
        // (no column mask here: features >= N have all-zero fragment-order weights and a zero-padded bias row, so their
        //  accumulators and activations ARE zero -- the generic epilogue's explicit mask cost two selects per value)
        if (BITS) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int bit = (c * RB + r) * 4 + e;
            if (bit < 32) shift_in_positive(lo, t[e >> 1][e & 1]); else shift_in_positive(hi, t[e >> 1][e & 1]);
          }
        }
        // exact 3-way split by truncation, two values per subtraction; each piece is the high half of t, r1, r2
        f32x2 r1[2], r2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x2 hm = {__uint_as_float(__float_as_uint(t[h][0]) & 0xFFFF0000u), __uint_as_float(__float_as_uint(t[h][1]) & 0xFFFF0000u)};
          r1[h] = t[h] - hm;
          const f32x2 mm = {__uint_as_float(__float_as_uint(r1[h][0]) & 0xFFFF0000u), __uint_as_float(__float_as_uint(r1[h][1]) & 0xFFFF0000u)};
          r2[h] = r1[h] - mm;                          // <= 8 significant bits left: its top 16 bits are exact
        }
        const u32x2 ph = (u32x2){pack_hi(__float_as_uint(t[0][0]), __float_as_uint(t[0][1])), pack_hi(__float_as_uint(t[1][0]), __float_as_uint(t[1][1]))};
        const u32x2 pm = (u32x2){pack_hi(__float_as_uint(r1[0][0]), __float_as_uint(r1[0][1])), pack_hi(__float_as_uint(r1[1][0]), __float_as_uint(r1[1][1]))};
        const u32x2 pl = (u32x2){pack_hi(__float_as_uint(r2[0][0]), __float_as_uint(r2[0][1])), pack_hi(__float_as_uint(r2[1][0]), __float_as_uint(r2[1][1]))};
        unsigned short* dst = dst0 + r * 16 * LDPB;
        *reinterpret_cast<u32x2*>(dst) = ph;
        *reinterpret_cast<u32x2*>(dst + PLANE) = pm;
        *reinterpret_cast<u32x2*>(dst + 2 * PLANE) = pl;
        if (PL && !(CLICA_EPI_ABLATE & 1)) {
          const unsigned po = (unsigned)(r * pl_group_bytes) + pl_cb;
          __builtin_amdgcn_raw_buffer_store_b64(ph, prsrc, po, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(pm, prsrc, po + 1024u, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(pl, prsrc, po + 2048u, 0, 0);
        }
        if (OUTV) {
          const unsigned off = (row < nrows && n0 < N) ? (unsigned)((row * ldo + n0) * 4) : kOobOffset;
          __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(t[0][0]), __float_as_uint(t[0][1]), __float_as_uint(t[1][0]), __float_as_uint(t[1][1])},
                                                 orsrc, off, 0, 0);
        }
      }
      if (PL && pl_ones && cb == (N >> 4) && (N & 31) != 0) {     // wave-uniform: feature N of the HBM copy is the constant 1 (see the generic epilogue)
        // the owning lane overwrites the (zero) hi piece of feature N behind its own 8-byte store: same lane, same address, program order
        const bool owner = N >= n0 && N < n0 + 4;
#pragma unroll
        for (int r = 0; r < RB; ++r)
          __builtin_amdgcn_raw_buffer_store_b16((unsigned short)0x3F80u, prsrc,
                                                owner ? (unsigned)(r * pl_group_bytes) + pl_cb + (unsigned)((N - n0) * 2) : kOobOffset, 0, 0);
      }
    } else if (BITS) {                                 // keep the bit positions of the blocks that follow
#pragma unroll
      for (int r = 0; r < RB; ++r) { if ((c * RB + r) * 4 < 32) lo <<= 4; else hi <<= 4; }
    }
  }
  if (BITS) {        // first value shifted in = bit 0: reverse.  lo took 32 values, hi the remaining CBW * RB * 4 - 32
    lo = __builtin_bitreverse32(lo);
    hi = __builtin_bitreverse32(hi) >> (64 - CBW * RB * 4);
  }
  lo_bits = lo; hi_bits = hi;
}

// ---- epilogue of one layer in the f16x2 arithmetic ---------------------------------------------------------------------------
//   ACT 1: t = acc * cmul + bias * s_out, LeakyReLU when `leaky` (slope in (0, 1): max(t, slope t))
//   ACT 2: t = acc * cmul, then t or slope * t by the forward's sign bit (backward chain link)
// t is the layer output IN THE OUTPUT TENSOR'S SCALED UNITS (cmul = s_out / (s_in s_w) folds all three scales; LeakyReLU and the gate
// are positively homogeneous).  It is split into hi / lo for the next layer's panel (LDS) and, `has_pl`, for the plane copy the
// weight-gradient kernel reads (planes.h, two pieces per unit); `has_out`: the fp32 copy gets t / s_out (exact: powers of two).
// Features >= N come out as exact zeros (zero weight fragments, zero-padded bias row), as in the bf16x3 fast path.
// MODE (round 6): 0 = which copies to write is decided at run time (wave-uniform flags: any combination);  1 = the plane copy only;
// 2 = the fp32 copy only, 16-byte stores -- the four combinations a training step runs (hidden layer -> planes, the 100-wide layer in
// front of an n-wide one -> fp32, both directions), compiled without the other paths: in the generic form every accumulator block
// carries the branches and the exec-mask code of the copies it does not write (phase trace: epilogue body 8.3 k cycles per 500-wide
// layer against 3.0 k for the same arithmetic in isolation, tools/proto/epi_probe.hip).  ACT 1 in MODE 1 / 2 is a LeakyReLU layer.
#ifndef CLICA_SPLIT_PLANE_AUX
#define CLICA_SPLIT_PLANE_AUX 2      // cache policy of the f16x2 plane-copy stores: 2 = nt (streaming).  Round 6, four interleaved A/B runs on one box: launch 86.3 ->
                                     // 83.2 us, and the weight-gradient phase that reads the planes 97.7 -> 94.6 us (step +2 %): written back (0), 30 MB of a launch's
                                     // last layers sit dirty in the L2s until the end-of-kernel write-back.  (sc1 = 16: slower than 0; sc0 | nt = 3, nt | sc1 = 18: as nt or worse.)
#endif
template <int ACT, bool BITS, int MODE>
__device__ __forceinline__ void split16_epilogue(const f32x4 (&acc)[RB][CBW], unsigned short* planes, const float* bias_row, const float slope, const bool leaky_,
                                                 const unsigned long long mbits, const int N, const int ncb, const int wave, const int lane,
                                                 const bool has_out_, const bool ovec_, float* out_rows, const __amdgpu_buffer_rsrc_t orsrc, const int ldo, const int nrows,
                                                 const bool has_pl_, const __amdgpu_buffer_rsrc_t prsrc, const int pl_group_bytes, const int pl_ones,
                                                 const float cmul, const float s_out, const float inv_s_out,
                                                 unsigned& lo_bits, unsigned& hi_bits, float& amax_scaled) {
  const bool leaky = MODE == 0 ? leaky_ : true, has_pl = MODE == 0 ? has_pl_ : (MODE == 1), has_out = MODE == 0 ? has_out_ : (MODE == 2),
             ovec = MODE == 0 ? ovec_ : true;
  const int i15 = lane & 15, kg = lane >> 4;
  const unsigned pl_lane = (unsigned)((i15 >> 2) * 256 + (i15 & 3) * 32 + kg * 8);
  const f32x2 slope2 = {slope, slope}, cmul2 = {cmul, cmul}, inv2 = {inv_s_out, inv_s_out}, sout2 = {s_out, s_out};
  const int mlo = (int)(unsigned)mbits, mhi = (int)(unsigned)(mbits >> 32);
  unsigned lo = 0u, hi = 0u;
  float am = 0.f;
#pragma unroll
  for (int c = 0; c < CBW; ++c) {
    const int cb = wave + c * WAVES;
    if (cb < ncb) {                                    // wave-uniform
      const int n0 = cb * 16 + kg * 4;
      f32x2 b01 = {0.f, 0.f}, b23 = {0.f, 0.f};
      if (ACT == 1) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(&bias_row[n0]);
        b01 = (f32x2){b4[0], b4[1]} * sout2; b23 = (f32x2){b4[2], b4[3]} * sout2;      // bias in the output's scaled units (once per column block)
      }
      const unsigned pl_cb = (unsigned)((cb >> 1) * 2 * 1024 + (cb & 1) * 128) + pl_lane;
      unsigned short* const dst0 = planes + i15 * LDPB + n0;
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int row = r * 16 + i15;
        const f32x4 v = acc[r][c];
        f32x2 t[2] = {(f32x2){v[0], v[1]}, (f32x2){v[2], v[3]}};
        if (ACT == 1) {
          t[0] = t[0] * cmul2 + b01; t[1] = t[1] * cmul2 + b23;      // (v_pk_fma_f32 under -ffp-contract=fast)
          if (leaky) {                                 // wave-uniform
            const f32x2 s0 = t[0] * slope2, s1 = t[1] * slope2;
            t[0] = (f32x2){fmaxf(t[0][0], s0[0]), fmaxf(t[0][1], s0[1])};
            t[1] = (f32x2){fmaxf(t[1][0], s1[0]), fmaxf(t[1][1], s1[1])};
          }
        }
        if (ACT == 2) {
          t[0] *= cmul2; t[1] *= cmul2;
          const f32x2 sl[2] = {t[0] * slope2, t[1] * slope2};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int bit = (c * RB + r) * 4 + e;      // compile time
            int sel; float tv = t[e >> 1][e & 1];
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(sel) : "v"(bit < 32 ? mlo : mhi), "n"(bit & 31));      // 0 or ~0
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(tv) : "v"(sel), "v"(tv), "v"(sl[e >> 1][e & 1]));      // bit ? t : slope t
            t[e >> 1][e & 1] = tv;
          }
        }
        if (BITS) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int bit = (c * RB + r) * 4 + e;
            if (bit < 32) shift_in_positive(lo, t[e >> 1][e & 1]); else shift_in_positive(hi, t[e >> 1][e & 1]);
          }
        }
        am = __builtin_fmaxf(am, __builtin_fmaxf(__builtin_fabsf(t[0][0]), __builtin_fabsf(t[0][1])));      // v_max3_f32 with |.| modifiers
        am = __builtin_fmaxf(am, __builtin_fmaxf(__builtin_fabsf(t[1][0]), __builtin_fabsf(t[1][1])));
        unsigned h0, l0, h1, l1;
        split16_pair(t[0], h0, l0);
        split16_pair(t[1], h1, l1);
        const u32x2 ph = {h0, h1}, pl = {l0, l1};
        unsigned short* dst = dst0 + r * 16 * LDPB;
        *reinterpret_cast<u32x2*>(dst) = ph;
        *reinterpret_cast<u32x2*>(dst + PLANE) = pl;
        if (has_pl) {                                  // wave-uniform
          const unsigned po = (unsigned)(r * pl_group_bytes) + pl_cb;
          __builtin_amdgcn_raw_buffer_store_b64(ph, prsrc, po, 0, CLICA_SPLIT_PLANE_AUX);
          __builtin_amdgcn_raw_buffer_store_b64(pl, prsrc, po + 1024u, 0, CLICA_SPLIT_PLANE_AUX);
        }
        if (has_out) {                                 // wave-uniform
          const f32x2 o0 = t[0] * inv2, o1 = t[1] * inv2;
          if (ovec) {
            const unsigned off = (row < nrows && n0 < N) ? (unsigned)((row * ldo + n0) * 4) : kOobOffset;
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(o0[0]), __float_as_uint(o0[1]), __float_as_uint(o1[0]), __float_as_uint(o1[1])},
                                                   orsrc, off, 0, 0);
          } else {
            const float ov[4] = {o0[0], o0[1], o1[0], o1[1]};
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (row < nrows && n0 + e < N) out_rows[(int64_t)row * ldo + n0 + e] = ov[e];
          }
        }
      }
      if (has_pl && pl_ones && cb == (N >> 4) && (N & 31) != 0) {     // wave-uniform: feature N of the HBM copy is the constant 1
        const bool owner = N >= n0 && N < n0 + 4;
        const unsigned short one = 0x3C00u;      // fp16 1.0 -- NOT the activation's scale: the weight-gradient kernel takes the scale of X out of dW only
#pragma unroll
        for (int r = 0; r < RB; ++r)
          __builtin_amdgcn_raw_buffer_store_b16(one, prsrc, owner ? (unsigned)(r * pl_group_bytes) + pl_cb + (unsigned)((N - n0) * 2) : kOobOffset, 0, 0);
      }
    } else if (BITS) {                                 // keep the bit positions of the blocks that follow
#pragma unroll
      for (int r = 0; r < RB; ++r) { if ((c * RB + r) * 4 < 32) lo <<= 4; else hi <<= 4; }
    }
  }
  if (BITS) {
    lo = __builtin_bitreverse32(lo);
    hi = __builtin_bitreverse32(hi) >> (64 - CBW * RB * 4);
  }
  lo_bits = lo; hi_bits = hi;
  amax_scaled = am;
}

// One wave's share of the L2 warm-up of layer `lt` (only if it fits: at most NL KB per participating wave of the XCD): plain
// 16-byte loads of this wave's 1 KB slices of the packed weights, values discarded by the caller once they have landed.
template <int NL, int NP>
__device__ __forceinline__ void warm_up_l2(const SplitArgs& a, const int L, const int lt, const int limit, const int wave, const int lane, u32x4 (&warm)[NL]) {
  constexpr int WW = WAVES / 2;                                               // participating waves per workgroup (the early half)
  const int nwx = (((int)gridDim.x + 7) >> 3) * WW;                           // ... per XCD (workgroups go round-robin over the eight)
  const int xw = ((int)blockIdx.x >> 3) * WW + wave;
  if (lt >= L) return;
  const int64_t loads = (NP * a.ent3[lt] + 63) >> 6;                           // 64 entries of 16 B per wave instruction
  if (loads > (int64_t)(limit < NL ? limit : NL) * nwx) return;
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    const int64_t idx = xw + (int64_t)u * nwx;
    if (idx < loads) {
      const int64_t e = idx * 64 + lane;
      warm[u] = a.packed3[a.off3[lt] + (e < NP * a.ent3[lt] ? e : 0)];
    }
  }
}

// The tail of a training step's backward chain (SplitArgs::Tail): two small fp32 products over the workgroup's 48 rows on
// v_mfma_f32_16x16x4_f32 (exact fp32 products, the arithmetic of the native GEMMs) --
//   last layer   slab[i][j] = sum_r dzl[r][i] al[r][j]      A = dzl^T (n_l <= 16 rows of the product), B = al  (w_l columns)
//   first layer  slab[j][k] = sum_r dz0[r][j] x[r][k]       A = x^T   (n_0 <= 15),                     B = dz0 (w_0 columns), stored transposed
// Waves 0..3 take the last layer, 4..7 the first; a wave owns two 16-column blocks of the wide operand and walks the 48 rows in twelve
// k-steps of four: 24 MFMAs, 36 loads, all in flight.  Bias gradients ride as one more column / row: B column w_l = 1 (column sums of
// dzl), A row n_0 = 1 (column sums of dz0).  Operands come straight from global memory in the MFMA's own lane order -- no LDS.
// (First version: vector-ALU products with the narrow operand handed out by v_readlane -- 768 dependent readlane + FMA pairs per wave
//  on two waves per job: +20 us on the chain launch, more than the 12.8 us launch it replaced.)
__device__ __forceinline__ void tail_wgrad(const SplitArgs::Tail& T, int64_t row0, int nrows, int wave, int lane) {
  static_assert(WAVES == 8 && ROWS == 48, "tail_wgrad: eight waves, twelve k-steps");
  const bool first = wave >= 4;
  const int q = wave & 3, l15 = lane & 15, kr = lane >> 4;
  const int n = first ? T.n_0 : T.n_l, width = first ? T.w_0 : T.w_l;
  const float* __restrict__ narrow = first ? T.x : T.dzl;
  const float* __restrict__ wide = first ? T.dz0 : T.al;
  const int64_t ldn = first ? T.ld_x : T.ld_dzl, ldw = first ? T.ld_dz0 : T.ld_al;
  float av[ROWS / 4], bv[2][ROWS / 4];
  auto load_wide = [&]() {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int col = (2 * q + c) * 16 + l15;
#pragma unroll
      for (int v = 0; v < ROWS / 4; ++v) {
        const int r = 4 * v + kr;
        bv[c][v] = (r < nrows && col < width) ? wide[(row0 + r) * ldw + col] : ((!first && r < nrows && col == width) ? 1.f : 0.f);
      }
    }
  };
  // dz0 was written by THIS workgroup's last epilogue: its stores are complete (workgroup-scope release) before anybody reads them back.
  // (The last layer's operands do not depend on this launch, but requested in front of the fence they would hold it up for an HBM
  //  round trip; the kernel's prologue has touched them into L2 instead.)
  __threadfence_block();
  __syncthreads();
#pragma unroll
  for (int v = 0; v < ROWS / 4; ++v) {
    const int r = 4 * v + kr;
    av[v] = (r < nrows && l15 < n) ? narrow[(row0 + r) * ldn + l15] : ((first && r < nrows && l15 == n) ? 1.f : 0.f);
  }
  load_wide();
  f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int v = 0; v < ROWS / 4; ++v)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[v], bv[c][v], acc[c], 0, 0, 0);
  // D[i = 4 (lane / 16) + e][j = column]
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int col = (2 * q + c) * 16 + l15;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = 4 * kr + e;
      if (!first) {
        if (i < n && col < width) T.slab_l[((size_t)blockIdx.x * n + i) * width + col] = acc[c][e];
        if (i < n && col == width) T.dbslab_l[(size_t)blockIdx.x * n + i] = acc[c][e];
      } else {
        if (i < n && col < width) T.slab_0[((size_t)blockIdx.x * width + col) * n + i] = acc[c][e];
        if (i == n && col < width) T.dbslab_0[(size_t)blockIdx.x * width + col] = acc[c][e];
      }
    }
  }
}

template <int AR>
__global__ __launch_bounds__(THREADS) void mlp_split_k(SplitArgs a) {
  constexpr int NP = Arith<AR>::NP;
  extern __shared__ __attribute__((aligned(16))) unsigned short planes[];     // [NP][ROWS][LDPB] bf16 / fp16 bit patterns
  const Args& g = a.g;
  const int lane = threadIdx.x & 63;
  const int lane_id = lane;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  ST_DECL;
  ST_STAMP(MAXL - 1, 0);                               // kernel entry (trace build; the stacks traced have < MAXL layers)
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int nrows = (int)min((int64_t)ROWS, g.M - row0);

  const int mslot = (wave * 64 + lane) * 8;
  auto mask_rsrc = [&](const unsigned long long* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(p) + (p ? (int64_t)blockIdx.x * WAVES * 64 : 0), 0,
                                             p ? WAVES * 64 * 8 : 0, kRsrcWord3);
  };
  // Prologue.  Everything it needs from memory is REQUESTED first -- the first layer's weights, the sign bits, every layer's bias,
  // the input rows, the mixing weights -- and only then consumed: the trace showed 27.9 k cycles (12 % of the forward launch)
  // between kernel entry and the first layer when each of these was a dependent round trip of its own (seven bias loops, the
  // input, three mixing layers reading their weights from global memory), all of them cold.
  u32x4 wpre[NP][CBW];
  request_first_w3<NP>(a.packed3 + a.off3[0], a.ent3[0], g.layer[0].K, g.layer[0].N, wave, lane, wpre);
  // the tail's wide operand of the last layer (tail_wgrad) comes from HBM, cold: its 48 rows are touched HERE (one load per 128 bytes,
  // value discarded), where every wave waits for memory anyway, so that the tail finds them in this XCD's L2
  float tail_touch = 0.f;
  if (a.tail.slab_l && wave < 3) {
    const int idx = wave * 64 + lane, r = idx >> 2, c = (idx & 3) * 32;
    if (r < nrows && c < a.tail.w_l) tail_touch = a.tail.al[(row0 + r) * a.tail.ld_al + c];
  }
  if (a.tail.slab_l && wave == 3 && lane < nrows) tail_touch = a.tail.x[(row0 + lane) * a.tail.ld_x];      // ... and the first layer's narrow one
  u32x2 mraw = __builtin_amdgcn_raw_buffer_load_b64(mask_rsrc(g.layer[0].dact ? g.layer[0].mask_in : nullptr), mslot, 0, 0);
  // all layers' biases in the LDS left beside the planes (rows zero-padded to KI): the epilogue reads four consecutive
  // features with one ds_read_b128 instead of holding them in registers across the k-loop
  float* bias_lds = reinterpret_cast<float*>(planes + NP * PLANE);
  volatile int* prog = reinterpret_cast<volatile int*>(bias_lds + a.boff[g.L]);      // k-loop progress of the eight waves (layer_gemm_split)
#if CLICA_SPLIT_BALANCE
  if (lane == 0) prog[wave] = 0;
#endif
  // (prog is VOLATILE and reaches the stores as a generic pointer: every `prog[wave] = 0` was a flat_store followed by s_waitcnt vmcnt(0)
  //  -- in the prologue a wait for every weight / bias / input request issued above it, behind each layer's epilogue a drain of the
  //  wave's plane stores in front of the barrier.  The balance experiment is off; so are its stores.)
  unsigned* amax_lds = const_cast<unsigned*>(reinterpret_cast<volatile unsigned*>(prog)) + WAVES;      // f16x2: the workgroup's maxima, one word per tensor of the launch
  // ... and the layers' scale factors, derived ONCE here by one thread per layer (read through the scalar cache in every epilogue they
  // were two cold misses in front of the first layers' epilogues: 8.5 k cycles for a 100-wide layer against 3.3 k for the same layer later)
  float* lscale = reinterpret_cast<float*>(amax_lds + 16);          // [cmul | s_out | 1 / s_out][MAXL]
  float sc_out = 1.f, sc_in = 1.f, sc_w = 1.f;        // requested here, turned into the layer's factors below, behind every other request
  if constexpr (AR == 1) {
    if (threadIdx.x < Split16State::NT) amax_lds[threadIdx.x] = 0u;
    if ((int)threadIdx.x < g.L) {
      const int l = threadIdx.x;
      sc_out = (l == g.L - 1 && a.last_unscaled) ? 1.f : a.s_t[l + 1];
      sc_in = a.s_t[l]; sc_w = a.s_w[l];
    }
  }
  constexpr int BIAS_IT = (BIAS_LDS_MAX + THREADS - 1) / THREADS;
  float bv[BIAS_IT];
  const int btotal = a.boff[g.L];
#pragma unroll
  for (int u = 0; u < BIAS_IT; ++u) {
    const int idx = threadIdx.x + u * THREADS;
    bv[u] = 0.f;
    if (idx < btotal) {
      int l = 0;
#pragma unroll
      for (int q = 1; q < MAXL; ++q) l += (q < g.L && idx >= a.boff[q]) ? 1 : 0;
      const Layer& ly = g.layer[l];
      const int i = idx - a.boff[l];
      if (ly.bias && i < ly.N) bv[u] = ly.bias[i];
    }
  }
  float s_in0 = 1.f, in_max = 0.f;                     // f16x2: scale of the launch's input tensor, running max of its scaled magnitudes
  if constexpr (AR == 1) s_in0 = a.s_t[0];
  auto store_split = [&](int r, int k, float v) {
    if constexpr (AR == 0) {
      unsigned hb, mb, lb; split3(v, hb, mb, lb);
      planes[r * LDPB + k] = (unsigned short)(hb >> 16);
      planes[PLANE + r * LDPB + k] = (unsigned short)(mb >> 16);
      planes[2 * PLANE + r * LDPB + k] = (unsigned short)(lb >> 16);
    } else {
      const float t = v * s_in0;
      in_max = fmaxf(in_max, fabsf(t));
      split16_one(t, planes[r * LDPB + k], planes[PLANE + r * LDPB + k]);
    }
  };
  {
    const int K0 = g.layer[0].K, K16 = (K0 + KI - 1) & ~(KI - 1);
    if (g.mixW) {      // x = g(z) in a corner of the still empty plane storage (see mlp_fwd_k): same arithmetic order
      float* xa = reinterpret_cast<float*>(planes); float* xb = xa + ROWS * MIX_MAX_N;
      float* wm = xb + ROWS * MIX_MAX_N;                 // the mixing weights, staged once (mixL * n * n <= 3 * 256 floats)
      const int n = K0;
      float xv[2], wv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {                      // ROWS * MIX_MAX_N <= 2 * THREADS
        const int idx = threadIdx.x + u * THREADS;
        const int r = idx / n, k = idx - r * n;
        xv[u] = (idx < ROWS * n && r < nrows) ? g.X[(row0 + r) * g.ldx + k] : 0.f;
        wv[u] = idx < g.mixL * n * n ? g.mixW[idx] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < BIAS_IT; ++u) { const int idx = threadIdx.x + u * THREADS; if (idx < btotal) bias_lds[idx] = bv[u]; }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = threadIdx.x + u * THREADS;
        if (idx < ROWS * n) xa[idx] = xv[u];
        if (idx < g.mixL * n * n) wm[idx] = wv[u];
      }
      for (int idx = threadIdx.x + 2 * THREADS; idx < g.mixL * n * n; idx += THREADS) wm[idx] = g.mixW[idx];     // (more than two per thread: wide nets)
      ST_STAMP(MAXL - 1, 1);
      __syncthreads();
      ST_STAMP(MAXL - 1, 2);
      for (int l = 0; l < g.mixL; ++l) {
        const float* wl = wm + l * n * n;
        for (int idx = threadIdx.x; idx < ROWS * n; idx += THREADS) {
          const int r = idx / n, j = idx - r * n;
          float s = 0.f;
          for (int k = 0; k < n; ++k) s = fmaf(xa[r * n + k], wl[j * n + k], s);
          if (l < g.mixL - 1) s = s > 0.f ? s : s * g.mix_slope;
          xb[idx] = s;
        }
        __syncthreads();
        float* t = xa; xa = xb; xb = t;
      }
      float v[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) { const int idx = threadIdx.x + u * THREADS; v[u] = idx < ROWS * n ? xa[idx] : 0.f; }
      ST_STAMP(MAXL - 1, 3);
      __syncthreads();
      for (int idx = threadIdx.x; idx < ROWS * K16; idx += THREADS) { const int r = idx / K16; store_split(r, idx - r * K16, 0.f); }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = threadIdx.x + u * THREADS;
        if (idx < ROWS * n) {
          const int r = idx / n, k = idx - r * n;
          store_split(r, k, v[u]);
          if (r < nrows) g.xout[(row0 + r) * g.ldxo + k] = v[u];
        }
      }
    } else {
      // the input rows: requested for up to four values per thread before the first is used
      constexpr int XIT = 4;
      const SplitArgs::DyParts& P = a.parts;
      int nsp = 0;
      if (P.part) {
        nsp = P.nsplit;
        if (P.words && lp2::guard_falls_back(P.words, P.limit)) nsp = P.nsplit_alt;      // the sweep that wrote the partials (as bwd_reduce_k reads it)
        if (blockIdx.x == 0 && threadIdx.x < 64) {      // the forward's three means + the counter tick: bwd_reduce_k's extra block, verbatim
          float v[3] = {0.f, 0.f, 0.f};
          for (int b = threadIdx.x; b < P.nblocks; b += 64) {
            v[0] += P.blocksums[b * 3 + 0]; v[1] += P.blocksums[b * 3 + 1]; v[2] += P.blocksums[b * 3 + 2];
          }
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v[c] += __shfl_down(v[c], off, 64);
          if (threadIdx.x == 0) {
            P.means[0] = v[0] * P.inv_count; P.means[1] = v[1] * P.inv_count; P.means[2] = v[2] * P.inv_count;
            if (P.tick) P.tick[0] += 1;
          }
        }
      }
      for (int idx0 = threadIdx.x; idx0 < ROWS * K16; idx0 += XIT * THREADS) {
        float xv[XIT];
#pragma unroll
        for (int u = 0; u < XIT; ++u) {
          const int idx = idx0 + u * THREADS;
          const int r = idx / K16, k = idx - r * K16;
          const bool ok = idx < ROWS * K16 && r < nrows && k < K0;
          xv[u] = ok ? g.X[(row0 + r) * g.ldx + k] : 0.f;
          if (P.part && ok) {
            if (row0 + r < P.rows) {
              // bwd_reduce_k's order: four interleaved partial sums over the splits, then (t0 + t1) + (t2 + t3), then + what is there
              float t4[4] = {0.f, 0.f, 0.f, 0.f};
              const float* src = P.part + (row0 + r) * P.np + k;
              const long long stride = P.rows * (long long)P.np;
              for (int sp = 0; sp < nsp; sp += 4) {
#pragma unroll
                for (int sub = 0; sub < 4; ++sub)
                  if (sp + sub < nsp) t4[sub] += src[(long long)(sp + sub) * stride];
              }
              xv[u] = xv[u] + ((t4[0] + t4[1]) + (t4[2] + t4[3]));
            }
            P.dy[(row0 + r) * P.ldy + k] = xv[u];
          }
        }
#pragma unroll
        for (int u = 0; u < XIT; ++u) {
          const int idx = idx0 + u * THREADS;
          if (idx < ROWS * K16) { const int r = idx / K16; store_split(r, idx - r * K16, xv[u]); }
        }
      }
#pragma unroll
      for (int u = 0; u < BIAS_IT; ++u) { const int idx = threadIdx.x + u * THREADS; if (idx < btotal) bias_lds[idx] = bv[u]; }
    }
  }
  if constexpr (AR == 1) {
    if ((int)threadIdx.x < g.L) {
      const int l = threadIdx.x;
      lscale[l] = sc_out / (sc_in * sc_w); lscale[MAXL + l] = sc_out; lscale[2 * MAXL + l] = 1.f / sc_out;
    }
  }
  __syncthreads();
  asm volatile("" ::"v"(tail_touch));
  if constexpr (AR == 1) amax_wave_to_lds(&amax_lds[0], in_max, 1.f / s_in0);

#pragma unroll 1
  for (int l = 0; l < g.L; ++l) {
    const Layer& ly = g.layer[l];
    ST_STAMP(l, 0);
    f32x4 acc[RB][CBW];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int c = 0; c < CBW; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    SplitArgs::LayerQ q = a.q[l];                      // this layer's quick record: one wide scalar load, complete long before the k-loop ends
    const bool has_next = l + 1 < g.L;
    const int ncb_real = (q.N + 15) / 16;
    int nc = (ncb_real - wave + WAVES - 1) / WAVES;
    nc = nc < 0 ? 0 : (nc > CBW ? CBW : nc);
    const unsigned long long mbits = (unsigned long long)mraw.x | ((unsigned long long)mraw.y << 32);
    const u32x4* w0 = a.packed3 + q.off3;
    switch (nc) {
      case 4: layer_gemm_split<4, AR>(q.K, w0, q.ent3, planes, wave, lane, acc, wpre, prog); break;
      case 3: layer_gemm_split<3, AR>(q.K, w0, q.ent3, planes, wave, lane, acc, wpre, prog); break;
      case 2: layer_gemm_split<2, AR>(q.K, w0, q.ent3, planes, wave, lane, acc, wpre, prog); break;
      case 1: layer_gemm_split_narrow<1, AR>(q.K, w0, q.ent3, planes, wave, lane, acc, wpre); break;
      default: break;
    }
    // pin the record in scalar registers HERE (one wait, behind the k-loop's own work): read lazily it was a chain of dependent
    // scalar loads in front of the epilogue
    asm volatile("" : "+s"(q.planes), "+s"(q.mask_out), "+s"(q.N), "+s"(q.pl_units), "+s"(q.pl_ones), "+s"(q.fast_kind), "+s"(q.boff));
    ST_STAMP(l, 1);
    // L2 warm-up for the SHORT layers ahead.  In the training step the packed weights are cold in this XCD's L2 (mlp_pack3_k rewrote
    // them a moment ago), and a layer with one column block per wave has nothing to hide the first touch behind: the trace shows the
    // 500 -> 100 layer at 29.9 k cycles cold against 15.8 k with the weights resident, and the misses queue behind this kernel's own
    // plane stores (> 10 k cycles: look-ahead inside the k-loop does not cover them, and a wave that issues such a load anywhere
    // else stalls its NEXT k-loop behind it in the in-order vmcnt).  The one place where they cost nothing: the older wave of
    // each SIMD leaves a wide layer's k-loop ~16 k cycles before the younger one and only waits at the barrier -- there it
    // touches its 1 KB slices of the next two layers' weights if those are short; the values are discarded below.  In the backward
    // chain the WIDE layers are cold as well (packed a forward and a loss ago: layers 3 / 4 of the chain ran their k-loops at
    // 40 k / 51 k cycles per wave pair instead of 25 k / 42 k), so there the next layer is touched whatever its size (12 KB per early
    // wave for 500 x 500): chain launch 138 -> 125 us in the step; in the forward the same costs 3 us (a.warm_next).
    u32x4 warm[WARM_LOADS], warm2[WARM_LOADS2];
#pragma unroll
    for (int u = 0; u < WARM_LOADS; ++u) warm[u] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
    for (int u = 0; u < WARM_LOADS2; ++u) warm2[u] = (u32x4){0u, 0u, 0u, 0u};
    if (nc >= 3 && wave < WAVES / 2) {
      warm_up_l2<WARM_LOADS, NP>(a, g.L, l + 1, a.warm_next, wave, lane, warm);
      warm_up_l2<WARM_LOADS2, NP>(a, g.L, l + 2, WARM_LOADS2, wave, lane, warm2);
    }
    // Round 4: the next layer's first weights (read-only: no need to wait for the barrier) and sign bits are requested HERE.  Behind
    // the barrier all eight waves' 104 requests of 1 KB arrived at the CU's 64 B/clk address path at once and took ~1.7 k cycles to
    // get through with the matrix pipe idle (tools/split_trace.py, "request"); the four early waves of a wide layer reach this point
    // ~17 k cycles before the late ones, so their half is through long before the barrier opens.
    if (has_next) {
      const Layer& nx = g.layer[l + 1];
      request_first_w3<NP>(a.packed3 + a.off3[l + 1], a.ent3[l + 1], nx.K, nx.N, wave, lane, wpre);
      mraw = __builtin_amdgcn_raw_buffer_load_b64(mask_rsrc(nx.dact ? nx.mask_in : nullptr), mslot, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                   // every wave is done reading the planes
    ST_STAMP(l, 2);
    // the warm-up values are dead; consuming them costs a vmcnt(0) (the compiler cannot order the conditional loads against the weight
    // requests), so only the waves that issued them do it -- they were here ~17 k cycles early and everything of theirs has landed;
    // the late waves must NOT wait for the weight requests they issued a moment ago
    if (nc >= 3 && wave < WAVES / 2) {
#pragma unroll
      for (int u = 0; u < WARM_LOADS; ++u) asm volatile("" ::"v"(warm[u]));
#pragma unroll
      for (int u = 0; u < WARM_LOADS2; ++u) asm volatile("" ::"v"(warm2[u]));
    }
    ST_STAMP(l, 4);
    __builtin_amdgcn_sched_barrier(0);
    ST_STAMP(l, 5);
    if constexpr (AR == 1) {
      int lane = lane_id;
      asm volatile("" : "+v"(lane));
      const int N = q.N;
      const int ncb = ((N + KI - 1) & ~(KI - 1)) / 16;
      const float cmul = lscale[l], s_out = lscale[MAXL + l], inv_s_out = lscale[2 * MAXL + l];
      const bool has_out = ly.out != nullptr, has_pl = q.planes != nullptr;
      const bool ovec = has_out && ((reinterpret_cast<uintptr_t>(ly.out) & 15) == 0) && (ly.ldo % 4 == 0) && (N % 4 == 0);
      const __amdgpu_buffer_rsrc_t orsrc =
          __builtin_amdgcn_make_buffer_rsrc(has_out ? ly.out + row0 * ly.ldo : nullptr, 0,
                                            has_out ? (int)(((int64_t)(nrows - 1) * ly.ldo + N) * 4) : 0, kRsrcWord3);
      const int pl_group_bytes = q.pl_units * NP * 1024;
      const __amdgpu_buffer_rsrc_t prsrc =
          __builtin_amdgcn_make_buffer_rsrc(has_pl ? reinterpret_cast<char*>(q.planes) + (int64_t)blockIdx.x * RB * pl_group_bytes : nullptr, 0,
                                            has_pl ? RB * pl_group_bytes : 0, kRsrcWord3);
      float* out_rows = has_out ? ly.out + row0 * ly.ldo : nullptr;
      unsigned lo = 0u, hi = 0u;
      float am = 0.f;
      const float* brow = &bias_lds[q.boff];
#define CLICA_EPI16_ARGS(MB) acc, planes, brow, g.slope, ly.leaky != 0, MB, N, ncb, wave, lane, has_out, ovec, out_rows, orsrc, (int)ly.ldo, nrows, \
                             has_pl, prsrc, pl_group_bytes, q.pl_ones, cmul, s_out, inv_s_out, lo, hi, am
      const unsigned long long mb = ly.mask_in ? mbits : ~0ull;      // backward link without sign bits: no gate
      switch (q.fast_kind) {             // (launch_split: the training step's combinations; 0 = anything else)
        case 11: split16_epilogue<1, true, 1>(CLICA_EPI16_ARGS(mbits)); break;
        case 12: split16_epilogue<1, true, 2>(CLICA_EPI16_ARGS(mbits)); break;
        case 21: split16_epilogue<2, false, 1>(CLICA_EPI16_ARGS(mb)); break;
        case 22: split16_epilogue<2, false, 2>(CLICA_EPI16_ARGS(mb)); break;
        default:
          if (!ly.dact) {
            if (q.mask_out) split16_epilogue<1, true, 0>(CLICA_EPI16_ARGS(mbits));
            else split16_epilogue<1, false, 0>(CLICA_EPI16_ARGS(mbits));
          } else {
            split16_epilogue<2, false, 0>(CLICA_EPI16_ARGS(mb));
          }
      }
#undef CLICA_EPI16_ARGS
      ST_STAMP(l, 6);
      if (has_pl && q.pl_ones && (N & 31) == 0 && wave == 0) {   // the ones column in an extra unit
        const bool first = (lane & 1) == 0 && ((lane >> 3) & 1) == 0;
        const unsigned one = 0x3C00u;
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
          for (int pp = 0; pp < NP; ++pp) {
            const u32x4 v4 = (u32x4){(pp == 0 && first) ? one : 0u, 0u, 0u, 0u};
            __builtin_amdgcn_raw_buffer_store_b128(v4, prsrc, (unsigned)(r * pl_group_bytes + ((N >> 5) * NP + pp) * 1024 + lane * 16), 0, 0);
          }
      }
      if (q.mask_out) __builtin_amdgcn_raw_buffer_store_b64((u32x2){lo, hi}, mask_rsrc(q.mask_out), (wave * 64 + lane) * 8, 0, 0);
      amax_wave_to_lds(&amax_lds[l + 1], am, inv_s_out);
#if CLICA_SPLIT_BALANCE
      if (lane == 0) prog[wave] = 0;
#endif
      ST_STAMP(l, 3);
      __syncthreads();
      continue;
    }
    if (CLICA_SPLIT_FAST_EPI && q.fast_kind != 0) {
      // the training step's epilogues, from the quick record alone
      int lane = lane_id;
      asm volatile("" : "+v"(lane));
      const int N = q.N;
      const int ncb = ((N + KI - 1) & ~(KI - 1)) / 16;
      const int pl_group_bytes = q.pl_units * 3 * 1024;
      const __amdgpu_buffer_rsrc_t prsrc =
          __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(q.planes) + (int64_t)blockIdx.x * RB * pl_group_bytes, 0, RB * pl_group_bytes, kRsrcWord3);
      const __amdgpu_buffer_rsrc_t no_out = __builtin_amdgcn_make_buffer_rsrc(static_cast<float*>(nullptr), 0, 0, kRsrcWord3);
      unsigned lo = 0u, hi = 0u;
      if (q.fast_kind == 1) split_epilogue_fast<1, true, true, false>(acc, planes, &bias_lds[q.boff], g.slope, mbits, N, ncb, wave, lane, no_out, 0, nrows, prsrc, pl_group_bytes, q.pl_ones, lo, hi);
      else split_epilogue_fast<2, false, true, false>(acc, planes, &bias_lds[q.boff], g.slope, mbits, N, ncb, wave, lane, no_out, 0, nrows, prsrc, pl_group_bytes, q.pl_ones, lo, hi);
      ST_STAMP(l, 6);
      if (q.pl_ones && (N & 31) == 0 && wave == 0) {   // the ones column in an extra unit (see the generic path below)
        const bool first = (lane & 1) == 0 && ((lane >> 3) & 1) == 0;
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
          for (int pp = 0; pp < 3; ++pp) {
            const u32x4 v4 = (u32x4){(pp == 0 && first) ? 0x00003F80u : 0u, 0u, 0u, 0u};
            __builtin_amdgcn_raw_buffer_store_b128(v4, prsrc, (unsigned)(r * pl_group_bytes + ((N >> 5) * 3 + pp) * 1024 + lane * 16), 0, 0);
          }
      }
      __builtin_amdgcn_raw_buffer_store_b64((u32x2){lo, hi}, mask_rsrc(q.mask_out), (wave * 64 + lane) * 8, 0, 0);
#if CLICA_SPLIT_BALANCE
      if (lane == 0) prog[wave] = 0;
#endif
      ST_STAMP(l, 3);
      __syncthreads();
      continue;
    }

    // epilogue: lane = batch row r*16 + (lane & 15), features cb*16 + (lane >> 4)*4 + e
    // Everything the epilogue derives from the lane id is recomputed here from an opaque copy: left to itself the compiler
    // hoists those layer-invariant addresses out of the layer loop, they stay live across the k-loop (which needs every
    // register it can get), spill, and their scratch re-loads then force vmcnt waits in the middle of the epilogue's stores.
    int lane = lane_id;
    asm volatile("" : "+v"(lane));
    const int i15 = lane & 15, kg = lane >> 4;
    const int mslot = (wave * 64 + lane) * 8;
    const int N = ly.N;
    const int ncb = ((N + KI - 1) & ~(KI - 1)) / 16;            // incl. the next layer's k-padding (written as zeros)
    const bool use_mask = ly.dact && ly.mask_in;
    const bool slope01 = g.slope > 0.f && g.slope < 1.f;
    const bool has_out = ly.out != nullptr;
    const bool ovec = has_out && ((reinterpret_cast<uintptr_t>(ly.out) & 15) == 0) && (ly.ldo % 4 == 0) && (N % 4 == 0);
    const __amdgpu_buffer_rsrc_t orsrc =
        __builtin_amdgcn_make_buffer_rsrc(has_out ? ly.out + row0 * ly.ldo : nullptr, 0,
                                          has_out ? (int)(((int64_t)(nrows - 1) * ly.ldo + N) * 4) : 0, kRsrcWord3);
    // bf16-plane copy for the weight-gradient kernel: this workgroup's 48 rows are three 16-row groups of `pl_units` 32-feature
    // units, 3 KB (three 1 KB plane pieces) each; a lane's four features of one row are 8 bytes of a piece.  ALL 48 rows are
    // written (rows beyond the batch hold finite values that meet zero rows of the other operand).
    const bool has_pl = ly.planes != nullptr;
    const int pl_group_bytes = ly.pl_units * 3 * 1024;
    const __amdgpu_buffer_rsrc_t prsrc =
        __builtin_amdgcn_make_buffer_rsrc(has_pl ? reinterpret_cast<char*>(ly.planes) + (int64_t)blockIdx.x * RB * pl_group_bytes : nullptr, 0,
                                          has_pl ? RB * pl_group_bytes : 0, kRsrcWord3);
    const unsigned pl_lane = (unsigned)((i15 >> 2) * 256 + (i15 & 3) * 32 + kg * 8);
    unsigned lo = 0u, hi = 0u;
#pragma unroll
    for (int c = 0; c < CBW; ++c) {
      const int cb = wave + c * WAVES;
      if (cb < ncb) {                                    // wave-uniform
        const int n0 = cb * 16 + kg * 4;
        const bool ragged = cb * 16 + 16 > N;          // wave-uniform: only the last real block and the k-padding blocks need the column mask
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(&bias_lds[a.boff[l] + n0]);
        const unsigned pl_cb = (unsigned)((cb >> 1) * 3 * 1024 + (cb & 1) * 128) + pl_lane;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const int row = r * 16 + i15;
          f32x4 v = acc[r][c];
          unsigned hb[4], mb[4], lb[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int bit = (c * RB + r) * 4 + e;
            float t = v[e];
            if (!ly.dact) {
              t += b4[e];
              if (ly.leaky) t = slope01 ? fmaxf(t, t * g.slope) : (t > 0.f ? t : t * g.slope);   // 0 < slope < 1: leaky = max(t, slope t)
            } else if (use_mask) {
              const unsigned mk = bit < 32 ? ((unsigned)mbits >> bit) : ((unsigned)(mbits >> 32) >> (bit - 32));
              t = (mk & 1u) ? t : t * g.slope;
            }
            if (ragged && n0 + e >= N) t = 0.f;
            // sign bits are shifted in from the right and reversed once at the end: `(t > 0) ? 1u << bit : 0` costs a VGPR per
            // constant (no literal operand in v_cndmask), and the compiler keeps all 48 of them live across the k-loop
            if (bit < 32) lo = (lo << 1) | (unsigned)(t > 0.f); else hi = (hi << 1) | (unsigned)(t > 0.f);
            v[e] = t;
            split3(t, hb[e], mb[e], lb[e]);
          }
          unsigned short* dst = planes + row * LDPB + n0;
          const u32x2 ph = (u32x2){(hb[0] >> 16) | hb[1], (hb[2] >> 16) | hb[3]};
          const u32x2 pm = (u32x2){(mb[0] >> 16) | mb[1], (mb[2] >> 16) | mb[3]};
          const u32x2 pl = (u32x2){(lb[0] >> 16) | lb[1], (lb[2] >> 16) | lb[3]};
          *reinterpret_cast<u32x2*>(dst) = ph;
          *reinterpret_cast<u32x2*>(dst + PLANE) = pm;
          *reinterpret_cast<u32x2*>(dst + 2 * PLANE) = pl;
          if (has_pl) {                                  // wave-uniform
            u32x2 phg = ph;
            // feature N of the HBM copy is the constant 1 (the weight-gradient GEMM returns db = dZ^T 1 in that column);
            // the on-chip panel keeps its zero there
            if (ragged && ly.pl_ones) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (n0 + e == N) phg[e >> 1] |= (e & 1) ? 0x3F800000u : 0x00003F80u;
            }
            const unsigned po = (unsigned)(r * pl_group_bytes) + pl_cb;
            __builtin_amdgcn_raw_buffer_store_b64(phg, prsrc, po, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(pm, prsrc, po + 1024u, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(pl, prsrc, po + 2048u, 0, 0);
          }
          if (ovec) {       // unconditional raw-buffer store, rows / features outside the tensor get an out-of-range offset
            const unsigned off = (row < nrows && n0 < N) ? (unsigned)((row * (int)ly.ldo + n0) * 4) : kOobOffset;
            // plain write-back stores (not nt): the next layer's weight loads queue behind these in the wave's in-order
            // vmcnt, and an L2 write acknowledges an order of magnitude sooner than a streaming write to HBM does
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orsrc, off, 0, 0);
          } else if (has_out) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (row < nrows && n0 + e < N) ly.out[(row0 + row) * ly.ldo + n0 + e] = v[e];
          }
        }
      } else {                                           // keep the bit positions of the blocks that follow
#pragma unroll
        for (int r = 0; r < RB; ++r) { if ((c * RB + r) * 4 < 32) lo <<= 4; else hi <<= 4; }
      }
    }
    lo = __builtin_bitreverse32(lo); hi = __builtin_bitreverse32(hi) >> (64 - CBW * RB * 4);
    if (has_pl && ly.pl_ones && (N & 31) == 0 && wave == 0) {
      // N is a whole number of units: the ones column lives in an extra unit (feature 0 of unit N / 32) that no accumulator
      // block covers -- wave 0 writes it (hi plane: 1.0 at feature 0 of every row, everything else zero)
      const bool first = (lane & 1) == 0 && ((lane >> 3) & 1) == 0;      // this lane's 16 bytes start at feature 0 of the unit
#pragma unroll
      for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) {
          const u32x4 v4 = (u32x4){(pp == 0 && first) ? 0x00003F80u : 0u, 0u, 0u, 0u};
          __builtin_amdgcn_raw_buffer_store_b128(v4, prsrc, (unsigned)(r * pl_group_bytes + ((N >> 5) * 3 + pp) * 1024 + lane * 16), 0, 0);
        }
    }
    __builtin_amdgcn_raw_buffer_store_b64((u32x2){lo, hi}, mask_rsrc(ly.mask_out), mslot, 0, 0);
#if CLICA_SPLIT_BALANCE
    if (lane == 0) prog[wave] = 0;
#endif
    ST_STAMP(l, 3);
    __syncthreads();
  }
  if constexpr (AR == 1) {        // (behind the last layer's closing barrier) this workgroup's slot of the maxima
    if (blockIdx.x < a.cap_wg && threadIdx.x < Split16State::NT) a.part_t[(size_t)threadIdx.x * kS16CapWG + blockIdx.x] = amax_lds[threadIdx.x];
    if (blockIdx.x == 0 && (int)threadIdx.x <= g.L) a.count_t[threadIdx.x] = gridDim.x < a.cap_wg ? gridDim.x : a.cap_wg;
    // the guard: tensor t of this launch (0: its input, l + 1: output of layer l) was cut to fp16 on scale s_t[t]; a maximum beyond the
    // alarm (or a non-finite one) in this workgroup's rows poisons the step (the last layer's output stays fp32 when nobody re-splits it)
    if ((int)threadIdx.x <= g.L && !((int)threadIdx.x == g.L && a.last_unscaled)) {
      const float m = __uint_as_float(amax_lds[threadIdx.x]);
      if (!(m * a.s_t[threadIdx.x] <= kF16Alarm)) s16::s16_raise_poison(a.st16);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) s16::s16_refresh_gen(a.st16);
  }
  if (a.tail.slab_l) tail_wgrad(a.tail, row0, nrows, wave, lane_id);
  ST_FLUSH(g.L);
}
#ifdef CLICA_SPLIT_TRACE
extern "C" int clica_debug_split_trace(unsigned long long* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(clica::fmlp::g_strace), &buf, sizeof(buf));
}
#endif

}  // namespace fmlp
}  // namespace clica

using namespace clica;

template <bool PACKED, bool AUX>
static int launch_mlp_inst(const fmlp::Args& g, hipStream_t st, const char* who) {
  using namespace fmlp;
  constexpr size_t lds = (size_t)ROWS * LDP * sizeof(float);
  auto k = mlp_fwd_k<PACKED, AUX>;
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
  (void)once;
  hipLaunchKernelGGL(k, dim3((unsigned)ceil_div(g.M, ROWS)), dim3(THREADS), lds, st, g);
  return launch_status(who);
}
static int launch_mlp(const fmlp::Args& g, bool packed, bool aux, hipStream_t st, const char* who) {
  if (packed) return aux ? launch_mlp_inst<true, true>(g, st, who) : launch_mlp_inst<true, false>(g, st, who);
  return launch_mlp_inst<false, false>(g, st, who);
}

extern "C" int clica_mlp_pack_bytes(int32_t n_layers, const int32_t* N, const int32_t* K, int32_t transpose, size_t* bytes) {
  using namespace fmlp;
  CLICA_CHECK_ARG(N && K && bytes && n_layers >= 1 && n_layers <= MAXL, "clica_mlp_pack_bytes: bad argument");
  int64_t f4 = 0;
  for (int l = 0; l < n_layers; ++l) {
    CLICA_CHECK_ARG(N[l] >= 1 && K[l] >= 1 && N[l] <= MAXW && K[l] <= MAXW, "clica_mlp_pack_bytes: layer %d is %d x %d (max %d)", l, N[l], K[l], MAXW);
    f4 += transpose ? pack_float4s(K[l], N[l]) : pack_float4s(N[l], K[l]);
  }
  *bytes = (size_t)f4 * 16;
  return CLICA_OK;
}

extern "C" int clica_mlp_signmask_bytes(int64_t M, size_t* bytes) {
  using namespace fmlp;
  CLICA_CHECK_ARG(bytes && M > 0, "clica_mlp_signmask_bytes: bad argument");
  *bytes = (size_t)ceil_div(M, ROWS) * WAVES * 64 * sizeof(uint64_t);
  return CLICA_OK;
}

static int launch_pack(fmlp::PackArgs& a, clica_stream_t stream, const char* who) {
  using namespace fmlp;
  hipLaunchKernelGGL(mlp_pack_k, dim3((unsigned)ceil_div(a.first[a.nseg], 256)), dim3(256), 0, as_stream(stream), a);
  return launch_status(who);
}

extern "C" int clica_mlp_pack(int32_t n_layers, const float* const* W, const int64_t* ldw, const int32_t* N, const int32_t* K,
                              int32_t transpose, float* packed, clica_stream_t stream) {
  using namespace fmlp;
  CLICA_CHECK_ARG(W && ldw && N && K && packed && n_layers >= 1 && n_layers <= MAXL, "clica_mlp_pack: bad argument");
  CLICA_CHECK_ARG((reinterpret_cast<uintptr_t>(packed) & 15) == 0, "clica_mlp_pack: packed buffer must be 16-byte aligned");
  PackArgs a{};
  a.nseg = n_layers; a.first[0] = 0;
  int64_t off = 0;
  for (int l = 0; l < n_layers; ++l) {
    CLICA_CHECK_ARG(W[l] && N[l] >= 1 && K[l] >= 1 && N[l] <= MAXW && K[l] <= MAXW && ldw[l] >= K[l], "clica_mlp_pack: layer %d: bad argument", l);
    const int rows = transpose ? K[l] : N[l], cols = transpose ? N[l] : K[l];
    a.seg[l] = PackSeg{W[l], ldw[l], rows, cols, transpose ? 1 : 0};
    a.dst[l] = reinterpret_cast<float4*>(packed) + off;
    const int64_t cnt = pack_float4s(rows, cols);
    off += cnt; a.first[l + 1] = a.first[l] + cnt;
  }
  return launch_pack(a, stream, "clica_mlp_pack");
}

extern "C" int clica_mlp_pack_both(int32_t n_layers, const float* const* W, const int64_t* ldw, const int32_t* N, const int32_t* K,
                                   float* packed_fwd, float* packed_bwd, clica_stream_t stream) {
  using namespace fmlp;
  CLICA_CHECK_ARG(W && ldw && N && K && packed_fwd && packed_bwd && n_layers >= 2 && n_layers <= MAXL, "clica_mlp_pack_both: bad argument");
  CLICA_CHECK_ARG((reinterpret_cast<uintptr_t>(packed_fwd) & 15) == 0 && (reinterpret_cast<uintptr_t>(packed_bwd) & 15) == 0,
                  "clica_mlp_pack_both: packed buffers must be 16-byte aligned");
  PackArgs a{};
  a.first[0] = 0;
  int sgi = 0;
  int64_t off = 0;
  for (int l = 0; l < n_layers; ++l) {
    CLICA_CHECK_ARG(W[l] && N[l] >= 1 && K[l] >= 1 && N[l] <= MAXW && K[l] <= MAXW && ldw[l] >= K[l], "clica_mlp_pack_both: layer %d: bad argument", l);
    a.seg[sgi] = PackSeg{W[l], ldw[l], N[l], K[l], 0};
    a.dst[sgi] = reinterpret_cast<float4*>(packed_fwd) + off;
    const int64_t cnt = pack_float4s(N[l], K[l]);
    off += cnt; a.first[sgi + 1] = a.first[sgi] + cnt; ++sgi;
  }
  off = 0;                              // chain order: layer L-1 first, down to layer 1; layer 0 has no data gradient
  for (int l = n_layers - 1; l >= 1; --l) {
    a.seg[sgi] = PackSeg{W[l], ldw[l], K[l], N[l], 1};
    a.dst[sgi] = reinterpret_cast<float4*>(packed_bwd) + off;
    const int64_t cnt = pack_float4s(K[l], N[l]);
    off += cnt; a.first[sgi + 1] = a.first[sgi] + cnt; ++sgi;
  }
  a.nseg = sgi;
  return launch_pack(a, stream, "clica_mlp_pack_both");
}

static int mlp_fwd_impl(const float* X, int64_t ldx, int64_t M, int32_t n_layers,
                        const float* const* W, const int64_t* ldw, const float* const* bias,
                        float* const* out, const int64_t* ldo, const int32_t* N, const int32_t* K,
                        const float* packed, uint64_t* const* signmask, float slope,
                        const float* mixW, int32_t mixL, float mix_slope, float* xout, int64_t ldxo, clica_stream_t stream) {
  using namespace fmlp;
  CLICA_CHECK_ARG(X && W && ldw && bias && out && ldo && N && K && M > 0, "clica_mlp_fwd: NULL pointer / empty batch");
  CLICA_CHECK_ARG(n_layers >= 1 && n_layers <= MAXL, "clica_mlp_fwd: %d layers (1..%d supported)", n_layers, MAXL);
  Args g{};
  g.X = X; g.ldx = ldx; g.M = M; g.L = n_layers; g.slope = slope; g.packed = packed;
  g.mixW = mixW; g.mixL = mixL; g.mix_slope = mix_slope; g.xout = xout; g.ldxo = ldxo;
  CLICA_CHECK_ARG(!packed || (reinterpret_cast<uintptr_t>(packed) & 15) == 0, "clica_mlp_fwd: packed weights must be 16-byte aligned");
  int64_t poff = 0;
  for (int l = 0; l < n_layers; ++l) {
    CLICA_CHECK_ARG(W[l] && out[l] && N[l] >= 1 && K[l] >= 1, "clica_mlp_fwd: layer %d: bad argument", l);
    CLICA_CHECK_ARG(N[l] <= MAXW && K[l] <= MAXW, "clica_mlp_fwd: layer %d is %d x %d; the on-chip panel holds widths <= %d "
                    "(use the per-layer clica_linear_fwd for wider encoders)", l, N[l], K[l], MAXW);
    CLICA_CHECK_ARG(ldw[l] >= K[l] && ldo[l] >= N[l], "clica_mlp_fwd: layer %d: leading dimension too small", l);
    CLICA_CHECK_ARG(l == 0 || K[l] == N[l - 1], "clica_mlp_fwd: layer %d input width %d != previous output width %d", l, K[l], N[l - 1]);
    unsigned long long* mo = (signmask && signmask[l]) ? reinterpret_cast<unsigned long long*>(signmask[l]) : nullptr;
    g.layer[l] = Layer{W[l], ldw[l], bias[l], out[l], ldo[l], nullptr, 0, mo, nullptr, N[l], K[l], l + 1 < n_layers ? 1 : 0, 0};
    g.pack_off[l] = poff; poff += pack_float4s(N[l], K[l]) * 4;
  }
  CLICA_CHECK_ARG(ldx >= K[0], "clica_mlp_fwd: ldx < K[0]");
  if (mixW) {
    CLICA_CHECK_ARG(mixL >= 1 && K[0] <= MIX_MAX_N && xout && ldxo >= K[0],
                    "clica_mlp_fwd_mixed: mixing net of width %d (max %d), %d layers; x_out required", K[0], MIX_MAX_N, mixL);
  }
  return launch_mlp(g, packed != nullptr, false, as_stream(stream), mixW ? "clica_mlp_fwd_mixed" : "clica_mlp_fwd");
}

extern "C" int clica_mlp_fwd(const float* X, int64_t ldx, int64_t M, int32_t n_layers,
                             const float* const* W, const int64_t* ldw, const float* const* bias,
                             float* const* out, const int64_t* ldo, const int32_t* N, const int32_t* K,
                             const float* packed, uint64_t* const* signmask, float slope, clica_stream_t stream) {
  return mlp_fwd_impl(X, ldx, M, n_layers, W, ldw, bias, out, ldo, N, K, packed, signmask, slope, nullptr, 0, 0.f, nullptr, 0, stream);
}

extern "C" int clica_mlp_fwd_mixed(const float* Z, int64_t ldz, int64_t M, const float* mix_W, int32_t mix_layers, float mix_slope,
                                   float* x_out, int64_t ldxo, int32_t n_layers,
                                   const float* const* W, const int64_t* ldw, const float* const* bias,
                                   float* const* out, const int64_t* ldo, const int32_t* N, const int32_t* K,
                                   const float* packed, uint64_t* const* signmask, float slope, clica_stream_t stream) {
  CLICA_CHECK_ARG(mix_W != nullptr, "clica_mlp_fwd_mixed: mix_W is NULL");
  return mlp_fwd_impl(Z, ldz, M, n_layers, W, ldw, bias, out, ldo, N, K, packed, signmask, slope, mix_W, mix_layers, mix_slope, x_out, ldxo, stream);
}


// Backward data chain in one launch: for j = 0..n-1   out[j] = (in_j B_j) * LeakyReLU'(act[j]),  in_0 = dY, in_j = out[j-1]
// where B_j is given ONLY in packed fragment order (clica_mlp_pack(..., transpose = 1) of the layers in chain
// order, i.e. the encoder's layers L-1 .. 1) and N[j] / K[j] are the logical output / contraction widths of
// link j (= K_l / N_l of the encoder layer it differentiates).
extern "C" int clica_mlp_dgrad(const float* dY, int64_t lddy, int64_t M, int32_t n_links,
                               const int32_t* N, const int32_t* K, const float* packed,
                               const float* const* act, const int64_t* ldact, const uint64_t* const* signmask,
                               float* const* out, const int64_t* ldo, float slope, clica_stream_t stream) {
  using namespace fmlp;
  CLICA_CHECK_ARG(dY && N && K && packed && act && ldact && out && ldo && M > 0, "clica_mlp_dgrad: NULL pointer / empty batch");
  CLICA_CHECK_ARG(n_links >= 1 && n_links <= MAXL, "clica_mlp_dgrad: %d links (1..%d supported)", n_links, MAXL);
  CLICA_CHECK_ARG((reinterpret_cast<uintptr_t>(packed) & 15) == 0, "clica_mlp_dgrad: packed weights must be 16-byte aligned");
  Args g{};
  g.X = dY; g.ldx = lddy; g.M = M; g.L = n_links; g.slope = slope; g.packed = packed;
  int64_t poff = 0;
  for (int j = 0; j < n_links; ++j) {
    CLICA_CHECK_ARG(out[j] && N[j] >= 1 && K[j] >= 1 && N[j] <= MAXW && K[j] <= MAXW, "clica_mlp_dgrad: link %d is %d x %d (max %d)", j, N[j], K[j], MAXW);
    CLICA_CHECK_ARG(ldo[j] >= N[j] && (!act[j] || ldact[j] >= N[j]), "clica_mlp_dgrad: link %d: leading dimension too small", j);
    CLICA_CHECK_ARG(j == 0 || K[j] == N[j - 1], "clica_mlp_dgrad: link %d contraction %d != previous width %d", j, K[j], N[j - 1]);
    const unsigned long long* mi = (signmask && signmask[j]) ? reinterpret_cast<const unsigned long long*>(signmask[j]) : nullptr;
    g.layer[j] = Layer{nullptr, 0, nullptr, out[j], ldo[j], act[j], ldact[j], nullptr, mi, N[j], K[j], 0, 1};
    g.pack_off[j] = poff; poff += pack_float4s(N[j], K[j]) * 4;
  }
  CLICA_CHECK_ARG(lddy >= K[0], "clica_mlp_dgrad: lddy < K[0]");
  bool aux = false;
  for (int j = 0; j < n_links; ++j) aux = aux || (g.layer[j].aux && !g.layer[j].mask_in);
  return launch_mlp(g, true, aux, as_stream(stream), "clica_mlp_dgrad");
}

// ---- split-bf16 entry points (see mlp_split_k) ------------------------------------------------------------------------
static int pack3_fill(fmlp::Pack3Args& a, int& sgi, int64_t& off, const float* W, int64_t ldw, int rows, int cols, int transposed, void* base) {
  using namespace fmlp;
  a.seg[sgi] = PackSeg{W, ldw, rows, cols, transposed};
  const int64_t ent = pack3_entries(rows, cols);
  a.dst[sgi] = reinterpret_cast<u32x4*>(base) + off;
  a.entries[sgi] = ent;
  a.first[sgi + 1] = a.first[sgi] + ent;
  off += 3 * ent; ++sgi;
  return 0;
}
static int launch_pack3(fmlp::Pack3Args& a, clica_stream_t stream, const char* who) {
  using namespace fmlp;
  hipLaunchKernelGGL(mlp_pack3_k, dim3((unsigned)ceil_div(a.first[a.nseg], 256)), dim3(256), 0, as_stream(stream), a);
  return launch_status(who);
}

extern "C" int clica_mlp_pack_split_bytes(int32_t n_layers, const int32_t* N, const int32_t* K, int32_t transpose, size_t* bytes) {
  using namespace fmlp;
  CLICA_CHECK_ARG(N && K && bytes && n_layers >= 1 && n_layers <= MAXL, "clica_mlp_pack_split_bytes: bad argument");
  int64_t e = 0;
  for (int l = 0; l < n_layers; ++l) {
    CLICA_CHECK_ARG(N[l] >= 1 && K[l] >= 1 && N[l] <= MAXW && K[l] <= MAXW, "clica_mlp_pack_split_bytes: layer %d is %d x %d (max %d)", l, N[l], K[l], MAXW);
    e += 3 * (transpose ? pack3_entries(K[l], N[l]) : pack3_entries(N[l], K[l]));
  }
  *bytes = (size_t)e * 16;
  return CLICA_OK;
}

extern "C" int clica_mlp_pack_split(int32_t n_layers, const float* const* W, const int64_t* ldw, const int32_t* N, const int32_t* K,
                                    int32_t transpose, void* packed, clica_stream_t stream) {
  using namespace fmlp;
  CLICA_CHECK_ARG(W && ldw && N && K && packed && n_layers >= 1 && n_layers <= MAXL, "clica_mlp_pack_split: bad argument");
  CLICA_CHECK_ARG((reinterpret_cast<uintptr_t>(packed) & 15) == 0, "clica_mlp_pack_split: packed buffer must be 16-byte aligned");
  Pack3Args a{};
  int sgi = 0; int64_t off = 0;
  for (int l = 0; l < n_layers; ++l) {
    CLICA_CHECK_ARG(W[l] && N[l] >= 1 && K[l] >= 1 && N[l] <= MAXW && K[l] <= MAXW && ldw[l] >= K[l], "clica_mlp_pack_split: layer %d: bad argument", l);
    pack3_fill(a, sgi, off, W[l], ldw[l], transpose ? K[l] : N[l], transpose ? N[l] : K[l], transpose ? 1 : 0, packed);
  }
  a.nseg = sgi;
  return launch_pack3(a, stream, "clica_mlp_pack_split");
}

extern "C" int clica_mlp_pack_split_both(int32_t n_layers, const float* const* W, const int64_t* ldw, const int32_t* N, const int32_t* K,
                                         void* packed_fwd, void* packed_bwd, clica_stream_t stream) {
  using namespace fmlp;
  CLICA_CHECK_ARG(W && ldw && N && K && packed_fwd && packed_bwd && n_layers >= 2 && n_layers <= MAXL, "clica_mlp_pack_split_both: bad argument");
  CLICA_CHECK_ARG((reinterpret_cast<uintptr_t>(packed_fwd) & 15) == 0 && (reinterpret_cast<uintptr_t>(packed_bwd) & 15) == 0,
                  "clica_mlp_pack_split_both: packed buffers must be 16-byte aligned");
  Pack3Args a{};
  int sgi = 0; int64_t off = 0;
  for (int l = 0; l < n_layers; ++l) {
    CLICA_CHECK_ARG(W[l] && N[l] >= 1 && K[l] >= 1 && N[l] <= MAXW && K[l] <= MAXW && ldw[l] >= K[l], "clica_mlp_pack_split_both: layer %d: bad argument", l);
    pack3_fill(a, sgi, off, W[l], ldw[l], N[l], K[l], 0, packed_fwd);
  }
  off = 0;
  for (int l = n_layers - 1; l >= 1; --l) pack3_fill(a, sgi, off, W[l], ldw[l], K[l], N[l], 1, packed_bwd);
  a.nseg = sgi;
  return launch_pack3(a, stream, "clica_mlp_pack_split_both");
}

extern "C" int clica_mlp_planes_bytes(int64_t M, int32_t width, int32_t ones_column, size_t* bytes) {
  CLICA_CHECK_ARG(bytes && M > 0 && width >= 1, "clica_mlp_planes_bytes: bad argument");
  *bytes = planes::bytes(M, width, ones_column);
  return CLICA_OK;
}

static int launch_split(fmlp::SplitArgs& a, int arith, clica_stream_t stream, const char* who) {
  using namespace fmlp;
  a.boff[0] = 0;
  for (int l = 0; l < a.g.L; ++l) a.boff[l + 1] = a.boff[l] + ((a.g.layer[l].N + KI - 1) & ~(KI - 1));
  if (a.boff[a.g.L] > BIAS_LDS_MAX) {
    set_error("%s: the layer widths sum to %d (> %d): the on-chip bias table does not fit beside the bf16 planes", who, a.boff[a.g.L], BIAS_LDS_MAX);
    return CLICA_E_INVALID;
  }
  const bool slope01 = a.g.slope > 0.f && a.g.slope < 1.f;
  for (int l = 0; l < a.g.L; ++l) {
    const Layer& ly = a.g.layer[l];
    SplitArgs::LayerQ& q = a.q[l];
    q.planes = ly.planes; q.mask_out = ly.mask_out; q.N = ly.N; q.K = ly.K; q.pl_units = ly.pl_units; q.pl_ones = ly.pl_ones;
    q.boff = a.boff[l]; q.off3 = a.off3[l]; q.ent3 = a.ent3[l];
    // the combinations the training step runs (see split_epilogue_fast): plane copy only, slope in (0, 1), and either
    // bias + LeakyReLU + sign bits (forward hidden layer) or the sign-bit gate (backward link)
    q.fast_kind = 0;
    if (slope01 && ly.planes && !ly.out) {
      if (!ly.dact && ly.leaky && ly.mask_out) q.fast_kind = 1;
      else if (ly.dact && ly.mask_in && !ly.mask_out) q.fast_kind = 3;
    }
  }
  if (arith == 1) {
    if (!slope01) { set_error("%s: the f16x2 arithmetic needs a LeakyReLU slope in (0, 1), got %g", who, (double)a.g.slope); return CLICA_E_INVALID; }
    for (int l = 0; l < a.g.L; ++l) {      // the specialised f16x2 epilogues (split16_epilogue, MODE 1 / 2)
      const Layer& ly = a.g.layer[l];
      const bool pl_only = ly.planes && !ly.out;
      const bool out_vec = ly.out && !ly.planes && ((reinterpret_cast<uintptr_t>(ly.out) & 15) == 0) && (ly.ldo % 4 == 0) && (ly.N % 4 == 0);
      int k = 0;
      if (!ly.dact && ly.leaky && ly.mask_out) k = pl_only ? 11 : (out_vec ? 12 : 0);
      else if (ly.dact && !ly.mask_out) k = pl_only ? 21 : (out_vec ? 22 : 0);
      a.q[l].fast_kind = k;
    }
    const size_t lds = 2 * (size_t)PLANE * sizeof(unsigned short) + (size_t)a.boff[a.g.L] * sizeof(float) + 256 + ST_LDS_BYTES;
    constexpr size_t lds_max = 2 * (size_t)PLANE * sizeof(unsigned short) + (size_t)BIAS_LDS_MAX * sizeof(float) + 256 + ST_LDS_BYTES;
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_split_k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max), true);
    (void)once;
    hipLaunchKernelGGL(mlp_split_k<1>, dim3((unsigned)ceil_div(a.g.M, ROWS)), dim3(THREADS), lds, as_stream(stream), a);
    return launch_status(who);
  }
  const size_t lds = 3 * (size_t)PLANE * sizeof(unsigned short) + (size_t)a.boff[a.g.L] * sizeof(float) + 256 + ST_LDS_BYTES;      // + the waves' progress words
  constexpr size_t lds_max = 3 * (size_t)PLANE * sizeof(unsigned short) + (size_t)BIAS_LDS_MAX * sizeof(float) + 256 + ST_LDS_BYTES;
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_split_k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max), true);
  (void)once;
  hipLaunchKernelGGL(mlp_split_k<0>, dim3((unsigned)ceil_div(a.g.M, ROWS)), dim3(THREADS), lds, as_stream(stream), a);
  return launch_status(who);
}

static int mlp_fwd_split_impl(const float* X, int64_t ldx, int64_t M, const float* mix_W, int32_t mix_layers, float mix_slope,
                              float* x_out, int64_t ldxo, int32_t n_layers, const float* const* bias,
                              float* const* out, const int64_t* ldo, const int32_t* N, const int32_t* K,
                              const void* packed_split, uint64_t* const* signmask, void* const* planes, float slope,
                              void* state16, clica_stream_t stream) {
  using namespace fmlp;
  const int NPc = state16 ? 2 : 3;
  CLICA_CHECK_ARG(X && bias && out && ldo && N && K && packed_split && M > 0, "clica_mlp_fwd_split: NULL pointer / empty batch");
  CLICA_CHECK_ARG(n_layers >= 1 && n_layers <= MAXL, "clica_mlp_fwd_split: %d layers (1..%d supported)", n_layers, MAXL);
  CLICA_CHECK_ARG((reinterpret_cast<uintptr_t>(packed_split) & 15) == 0, "clica_mlp_fwd_split: packed weights must be 16-byte aligned");
  SplitArgs a{};
  a.warm_next = fmlp::WARM_LOADS2;      // forward: short layers only (its wide layers' weights are fresh out of mlp_pack3_k: not cold-sensitive)
  Args& g = a.g;
  g.X = X; g.ldx = ldx; g.M = M; g.L = n_layers; g.slope = slope;
  g.mixW = mix_W; g.mixL = mix_layers; g.mix_slope = mix_slope; g.xout = x_out; g.ldxo = ldxo;
  a.packed3 = reinterpret_cast<const u32x4*>(packed_split);
  int64_t off = 0;
  for (int l = 0; l < n_layers; ++l) {
    void* pl = planes ? planes[l] : nullptr;
    CLICA_CHECK_ARG((out[l] || pl) && N[l] >= 1 && K[l] >= 1 && N[l] <= MAXW && K[l] <= MAXW, "clica_mlp_fwd_split: layer %d is %d x %d (max %d)", l, N[l], K[l], MAXW);
    CLICA_CHECK_ARG(!out[l] || ldo[l] >= N[l], "clica_mlp_fwd_split: layer %d: leading dimension too small", l);
    CLICA_CHECK_ARG(l == 0 || K[l] == N[l - 1], "clica_mlp_fwd_split: layer %d input width %d != previous output width %d", l, K[l], N[l - 1]);
    CLICA_CHECK_ARG((reinterpret_cast<uintptr_t>(pl) & 15) == 0, "clica_mlp_fwd_split: layer %d: plane buffer must be 16-byte aligned", l);
    unsigned long long* mo = (signmask && signmask[l]) ? reinterpret_cast<unsigned long long*>(signmask[l]) : nullptr;
    g.layer[l] = Layer{nullptr, 0, bias[l], out[l], ldo[l], nullptr, 0, mo, nullptr, N[l], K[l], l + 1 < n_layers ? 1 : 0, 0,
                       reinterpret_cast<unsigned short*>(pl), planes::units(N[l], 1), 1};
    a.off3[l] = off; a.ent3[l] = pack3_entries(N[l], K[l]); off += NPc * a.ent3[l];
  }
  CLICA_CHECK_ARG(ldx >= K[0], "clica_mlp_fwd_split: ldx < K[0]");
  if (mix_W) CLICA_CHECK_ARG(mix_layers >= 1 && K[0] <= MIX_MAX_N && x_out && ldxo >= K[0], "clica_mlp_fwd_split: bad mixing-net arguments");
  if (state16) {
    Split16State* st = reinterpret_cast<Split16State*>(state16);
    CLICA_CHECK_ARG(ceil_div(M, (int64_t)ROWS) <= (int64_t)kS16CapWG, "clica_mlp_fwd_split16: M = %lld rows exceeds the state's %u workgroup slots", (long long)M, kS16CapWG);
    a.s_t = st->sA; a.s_w = st->sW; a.part_t = s16_partA(st); a.count_t = st->cntA; a.cap_wg = kS16CapWG; a.st16 = st;
    a.last_unscaled = (planes && planes[n_layers - 1]) ? 0 : 1;
  }
  return launch_split(a, state16 ? 1 : 0, stream, state16 ? "clica_mlp_fwd_split16" : "clica_mlp_fwd_split");
}

extern "C" int clica_mlp_fwd_split(const float* X, int64_t ldx, int64_t M, const float* mix_W, int32_t mix_layers, float mix_slope,
                                   float* x_out, int64_t ldxo, int32_t n_layers, const float* const* bias,
                                   float* const* out, const int64_t* ldo, const int32_t* N, const int32_t* K,
                                   const void* packed_split, uint64_t* const* signmask, void* const* planes, float slope,
                                   clica_stream_t stream) {
  return mlp_fwd_split_impl(X, ldx, M, mix_W, mix_layers, mix_slope, x_out, ldxo, n_layers, bias, out, ldo, N, K, packed_split, signmask, planes,
                            slope, nullptr, stream);
}
extern "C" int clica_mlp_fwd_split16(const float* X, int64_t ldx, int64_t M, const float* mix_W, int32_t mix_layers, float mix_slope,
                                     float* x_out, int64_t ldxo, int32_t n_layers, const float* const* bias,
                                     float* const* out, const int64_t* ldo, const int32_t* N, const int32_t* K,
                                     const void* packed_split16, uint64_t* const* signmask, void* const* planes, float slope,
                                     void* state, clica_stream_t stream) {
  CLICA_CHECK_ARG(state != nullptr, "clica_mlp_fwd_split16: state is NULL");
  return mlp_fwd_split_impl(X, ldx, M, mix_W, mix_layers, mix_slope, x_out, ldxo, n_layers, bias, out, ldo, N, K, packed_split16, signmask, planes,
                            slope, state, stream);
}

static int mlp_dgrad_split_impl(const float* dY, int64_t lddy, int64_t M, int32_t n_links, const int32_t* N, const int32_t* K,
                                const void* packed_split, const uint64_t* const* signmask,
                                float* const* out, const int64_t* ldo, void* const* planes, float slope, void* state16, clica_stream_t stream,
                                const clica_chain_tail* tail = nullptr) {
  using namespace fmlp;
  const int NPc = state16 ? 2 : 3;
  CLICA_CHECK_ARG(dY && N && K && packed_split && out && ldo && M > 0, "clica_mlp_dgrad_split: NULL pointer / empty batch");
  CLICA_CHECK_ARG(n_links >= 1 && n_links <= MAXL, "clica_mlp_dgrad_split: %d links (1..%d supported)", n_links, MAXL);
  CLICA_CHECK_ARG((reinterpret_cast<uintptr_t>(packed_split) & 15) == 0, "clica_mlp_dgrad_split: packed weights must be 16-byte aligned");
  SplitArgs a{};
  a.warm_next = fmlp::WARM_LOADS;       // backward chain: every layer (its weights were packed a forward + a loss ago: HBM-cold by now)
  Args& g = a.g;
  g.X = dY; g.ldx = lddy; g.M = M; g.L = n_links; g.slope = slope;
  a.packed3 = reinterpret_cast<const u32x4*>(packed_split);
  int64_t off = 0;
  for (int j = 0; j < n_links; ++j) {
    void* pl = planes ? planes[j] : nullptr;
    CLICA_CHECK_ARG((out[j] || pl) && N[j] >= 1 && K[j] >= 1 && N[j] <= MAXW && K[j] <= MAXW, "clica_mlp_dgrad_split: link %d is %d x %d (max %d)", j, N[j], K[j], MAXW);
    CLICA_CHECK_ARG(!out[j] || ldo[j] >= N[j], "clica_mlp_dgrad_split: link %d: leading dimension too small", j);
    CLICA_CHECK_ARG(j == 0 || K[j] == N[j - 1], "clica_mlp_dgrad_split: link %d contraction %d != previous width %d", j, K[j], N[j - 1]);
    CLICA_CHECK_ARG((reinterpret_cast<uintptr_t>(pl) & 15) == 0, "clica_mlp_dgrad_split: link %d: plane buffer must be 16-byte aligned", j);
    const unsigned long long* mi = (signmask && signmask[j]) ? reinterpret_cast<const unsigned long long*>(signmask[j]) : nullptr;
    g.layer[j] = Layer{nullptr, 0, nullptr, out[j], ldo[j], nullptr, 0, nullptr, mi, N[j], K[j], 0, 1,
                       reinterpret_cast<unsigned short*>(pl), planes::units(N[j], 0), 0};
    a.off3[j] = off; a.ent3[j] = pack3_entries(N[j], K[j]); off += NPc * a.ent3[j];
  }
  CLICA_CHECK_ARG(lddy >= K[0], "clica_mlp_dgrad_split: lddy < K[0]");
  if (state16) {
    Split16State* st = reinterpret_cast<Split16State*>(state16);
    CLICA_CHECK_ARG(ceil_div(M, (int64_t)ROWS) <= (int64_t)kS16CapWG, "clica_mlp_dgrad_split16: M = %lld rows exceeds the state's %u workgroup slots", (long long)M, kS16CapWG);
    a.s_t = st->sD; a.s_w = st->sWC; a.cap_wg = kS16CapWG; a.part_t = s16_partD(st); a.count_t = st->cntD; a.st16 = st;
    a.last_unscaled = (planes && planes[n_links - 1]) ? 0 : 1;
  }
  if (tail) {      // the n-wide first / last layer's weight-gradient slabs behind the last link (SplitArgs::Tail)
    const int Le = tail->n_layers;
    CLICA_CHECK_ARG(tail->a_last && tail->x && tail->N && tail->K && tail->wgrad_workspace, "clica_mlp_dgrad_split_tail: NULL pointer in the tail descriptor");
    CLICA_CHECK_ARG(Le == n_links + 1, "clica_mlp_dgrad_split_tail: the chain has %d links, the encoder %d layers (links + 1 expected)", n_links, Le);
    CLICA_CHECK_ARG(wsplit::chain_tail_supported(Le, tail->N, tail->K), "clica_mlp_dgrad_split_tail: layer shapes not covered (clica_mlp_chain_tail_supported)");
    // chain link 0 contracts over the last layer's outputs, the last link produces dZ_0 (width of the first layer's output)
    CLICA_CHECK_ARG(K[0] == tail->N[Le - 1] && N[0] == tail->K[Le - 1] && N[n_links - 1] == tail->N[0],
                    "clica_mlp_dgrad_split_tail: the chain's widths do not match the encoder's first / last layer");
    CLICA_CHECK_ARG(out[n_links - 1] && tail->lda >= tail->K[Le - 1] && tail->ldx >= tail->K[0],
                    "clica_mlp_dgrad_split_tail: the last link needs its fp32 output (dZ_0); leading dimensions too small");
    SplitArgs::Tail& T = a.tail;
    int rc = wsplit::chain_tail_slabs(M, Le, tail->N, tail->K, tail->wgrad_workspace, tail->wgrad_workspace_bytes,
                                      &T.slab_0, &T.dbslab_0, &T.slab_l, &T.dbslab_l);
    if (rc) return rc;
    T.dzl = dY; T.ld_dzl = lddy; T.al = tail->a_last; T.ld_al = tail->lda; T.dz0 = out[n_links - 1]; T.ld_dz0 = ldo[n_links - 1];
    T.x = tail->x; T.ld_x = tail->ldx; T.n_l = tail->N[Le - 1]; T.w_l = tail->K[Le - 1]; T.w_0 = tail->N[0]; T.n_0 = tail->K[0];
    if (tail->dy_parts) {
      const clica_lp_dy_parts& dp = *tail->dy_parts;
      CLICA_CHECK_ARG(dp.part && dp.means && dp.blocksums && dp.rows >= 1 && dp.rows <= M && dp.n == K[0] && dp.np >= dp.n && dp.nsplit >= 1,
                      "clica_mlp_dgrad_split_tail: dy_parts does not match the chain's input (%lld rows of %d, chain input %lld x %d)",
                      (long long)dp.rows, dp.n, (long long)M, K[0]);
      a.parts = SplitArgs::DyParts{dp.part, dp.nsplit, dp.nsplit_alt > 0 ? dp.nsplit_alt : dp.nsplit, dp.np, dp.n, dp.rows, dp.guard_words, dp.guard_limit,
                                   dp.blocksums, dp.nblocks, dp.inv_count, dp.means, dp.tick, const_cast<float*>(dY), lddy};
    }
  }
  return launch_split(a, state16 ? 1 : 0, stream, state16 ? "clica_mlp_dgrad_split16" : "clica_mlp_dgrad_split");
}
// The backward chain of a training step: clica_mlp_dgrad_split (state == NULL) / clica_mlp_dgrad_split16 plus, behind the last link,
// the weight-gradient slabs of the encoder's n-wide first and last layer (clica_chain_tail, include/clica.h).
extern "C" int clica_mlp_dgrad_split_tail(const float* dY, int64_t lddy, int64_t M, int32_t n_links, const int32_t* N, const int32_t* K,
                                          const void* packed_split, const uint64_t* const* signmask,
                                          float* const* out, const int64_t* ldo, void* const* planes, float slope, void* state,
                                          const clica_chain_tail* tail, clica_stream_t stream) {
  CLICA_CHECK_ARG(tail != nullptr, "clica_mlp_dgrad_split_tail: tail is NULL");
  return mlp_dgrad_split_impl(dY, lddy, M, n_links, N, K, packed_split, signmask, out, ldo, planes, slope, state, stream, tail);
}

extern "C" int clica_mlp_dgrad_split(const float* dY, int64_t lddy, int64_t M, int32_t n_links, const int32_t* N, const int32_t* K,
                                     const void* packed_split, const uint64_t* const* signmask,
                                     float* const* out, const int64_t* ldo, void* const* planes, float slope, clica_stream_t stream) {
  return mlp_dgrad_split_impl(dY, lddy, M, n_links, N, K, packed_split, signmask, out, ldo, planes, slope, nullptr, stream);
}
extern "C" int clica_mlp_dgrad_split16(const float* dY, int64_t lddy, int64_t M, int32_t n_links, const int32_t* N, const int32_t* K,
                                       const void* packed_split16, const uint64_t* const* signmask,
                                       float* const* out, const int64_t* ldo, void* const* planes, float slope, void* state, clica_stream_t stream) {
  CLICA_CHECK_ARG(state != nullptr, "clica_mlp_dgrad_split16: state is NULL");
  return mlp_dgrad_split_impl(dY, lddy, M, n_links, N, K, packed_split16, signmask, out, ldo, planes, slope, state, stream);
}

// ---- f16x2 arithmetic: state, weights ---------------------------------------------------------------------------------------------
extern "C" int clica_split16_state_bytes(size_t* bytes) {
  CLICA_CHECK_ARG(bytes != nullptr, "clica_split16_state_bytes: bytes is NULL");
  *bytes = s16::kS16StateBytes;
  return CLICA_OK;
}
extern "C" int clica_split16_state_init(void* state, clica_stream_t stream) {
  CLICA_CHECK_ARG(state && (reinterpret_cast<uintptr_t>(state) & 15) == 0, "clica_split16_state_init: state must be a 16-byte aligned device buffer");
  hipLaunchKernelGGL(fmlp::split16_init_k, dim3(1), dim3(64), 0, as_stream(stream), reinterpret_cast<fmlp::Split16State*>(state), fmlp::kS16CapWG, fmlp::kS16CapPW);
  return launch_status("clica_split16_state_init");
}
extern "C" int clica_split16_update(void* state, int32_t n_layers, clica_stream_t stream) {
  CLICA_CHECK_ARG(state && n_layers >= 1 && n_layers <= fmlp::MAXL, "clica_split16_update: bad argument");
  hipLaunchKernelGGL(fmlp::split16_update_k, dim3(s16::kS16UpdateBlocks), dim3(256), 0, as_stream(stream), reinterpret_cast<fmlp::Split16State*>(state), (int)n_layers);
  return launch_status("clica_split16_update");
}
extern "C" int clica_split16_read(const void* state, int32_t* flags, int32_t* updates, float* scales_a, float* scales_d, float* scales_w,
                                  float* last_scales_a, float* last_scales_d, clica_stream_t stream) {
  CLICA_CHECK_ARG(state != nullptr, "clica_split16_read: state is NULL");
  fmlp::Split16State h;
  hipStream_t st = as_stream(stream);
  if (hipMemcpyAsync(&h, state, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return launch_status("clica_split16_read");
  if (flags) *flags = (int32_t)h.flags;
  if (updates) *updates = (int32_t)h.updates;
  for (int i = 0; i < fmlp::Split16State::NT; ++i) {
    if (scales_a) scales_a[i] = h.sA[i];
    if (scales_d) scales_d[i] = h.sD[i];
    if (scales_w) scales_w[i] = h.sW[i];
    if (last_scales_a) last_scales_a[i] = h.pA[i];
    if (last_scales_d) last_scales_d[i] = h.pD[i];
  }
  return CLICA_OK;
}
extern "C" int clica_split16_guard(const void* state, int32_t* flags, int32_t* skipped, int32_t* updates, int32_t* poisoned, clica_stream_t stream) {
  CLICA_CHECK_ARG(state != nullptr, "clica_split16_guard: state is NULL");
  fmlp::Split16State h;
  hipStream_t st = as_stream(stream);
  if (hipMemcpyAsync(&h, state, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return launch_status("clica_split16_guard");
  if (flags) *flags = (int32_t)h.flags;
  if (skipped) *skipped = (int32_t)h.skipped;
  if (updates) *updates = (int32_t)h.updates;
  if (poisoned) *poisoned = (h.poison == h.gen_copy) ? 1 : 0;      // the step whose producers ran last (data parallel: this rank's own verdict)
  return CLICA_OK;
}
namespace clica { namespace fmlp {
__global__ void split16_poison_export_k(const Split16State* st, float* slot) { slot[0] = (st->poison == st->gen_copy) ? 1.f : 0.f; }
__global__ void split16_set_dp_k(Split16State* st, const float* slot) { st->dp_poison = slot; }
} }
extern "C" int clica_split16_set_dp_poison(void* state, const float* slot, clica_stream_t stream) {
  CLICA_CHECK_ARG(state != nullptr, "clica_split16_set_dp_poison: state is NULL");
  hipLaunchKernelGGL(fmlp::split16_set_dp_k, dim3(1), dim3(1), 0, as_stream(stream), reinterpret_cast<fmlp::Split16State*>(state), slot);
  return launch_status("clica_split16_set_dp_poison");
}
extern "C" int clica_split16_poison_export(const void* state, float* slot, clica_stream_t stream) {
  CLICA_CHECK_ARG(state != nullptr && slot != nullptr, "clica_split16_poison_export: NULL argument");
  hipLaunchKernelGGL(fmlp::split16_poison_export_k, dim3(1), dim3(1), 0, as_stream(stream), reinterpret_cast<const fmlp::Split16State*>(state), slot);
  return launch_status("clica_split16_poison_export");
}
extern "C" int clica_split16_clear_flags(void* state, clica_stream_t stream) {
  CLICA_CHECK_ARG(state != nullptr, "clica_split16_clear_flags: state is NULL");
  if (hipMemsetAsync(&reinterpret_cast<fmlp::Split16State*>(state)->flags, 0, sizeof(unsigned), as_stream(stream)) != hipSuccess)
    return launch_status("clica_split16_clear_flags");
  return CLICA_OK;
}
extern "C" int clica_mlp_pack_split16_bytes(int32_t n_layers, const int32_t* N, const int32_t* K, int32_t transpose, size_t* bytes) {
  using namespace fmlp;
  CLICA_CHECK_ARG(N && K && bytes && n_layers >= 1 && n_layers <= MAXL, "clica_mlp_pack_split16_bytes: bad argument");
  int64_t e = 0;
  for (int l = 0; l < n_layers; ++l) {
    CLICA_CHECK_ARG(N[l] >= 1 && K[l] >= 1 && N[l] <= MAXW && K[l] <= MAXW, "clica_mlp_pack_split16_bytes: layer %d is %d x %d (max %d)", l, N[l], K[l], MAXW);
    e += 2 * (transpose ? pack3_entries(K[l], N[l]) : pack3_entries(N[l], K[l]));
  }
  *bytes = (size_t)e * 16;
  return CLICA_OK;
}
static int pack2_fill(fmlp::Pack2Args& a, int32_t n_layers, const float* const* W, const int64_t* ldw, const int32_t* N, const int32_t* K,
                      void* packed_fwd, void* packed_bwd, void* state) {
  using namespace fmlp;
  CLICA_CHECK_ARG(W && ldw && N && K && packed_fwd && packed_bwd && state && n_layers >= 2 && n_layers <= MAXL, "clica_mlp_pack_split16_both: bad argument");
  CLICA_CHECK_ARG((reinterpret_cast<uintptr_t>(packed_fwd) & 15) == 0 && (reinterpret_cast<uintptr_t>(packed_bwd) & 15) == 0,
                  "clica_mlp_pack_split16_both: packed buffers must be 16-byte aligned");
  Split16State* st = reinterpret_cast<Split16State*>(state);
  a = Pack2Args{};
  a.st = st;
  int sgi = 0; int64_t off = 0;
  auto fill = [&](const float* Wl, int64_t ld, int rows, int cols, int transposed, void* base, const float* sc, int layer) {
    a.seg[sgi] = PackSeg{Wl, ld, rows, cols, transposed};
    const int64_t ent = pack3_entries(rows, cols);
    a.dst[sgi] = reinterpret_cast<u32x4*>(base) + off;
    a.entries[sgi] = ent; a.first[sgi + 1] = a.first[sgi] + ent;
    a.scale[sgi] = sc; a.layer[sgi] = layer;
    off += 2 * ent; ++sgi;
  };
  for (int l = 0; l < n_layers; ++l) {
    CLICA_CHECK_ARG(W[l] && N[l] >= 1 && K[l] >= 1 && N[l] <= MAXW && K[l] <= MAXW && ldw[l] >= K[l], "clica_mlp_pack_split16_both: layer %d: bad argument", l);
    fill(W[l], ldw[l], N[l], K[l], 0, packed_fwd, &st->sW[l], l);
  }
  a.nfwd = sgi;
  off = 0;
  for (int l = n_layers - 1; l >= 1; --l) fill(W[l], ldw[l], K[l], N[l], 1, packed_bwd, &st->sW[l], -1);
  a.nseg = sgi;
  CLICA_CHECK_ARG(((a.first[a.nseg] + 63) >> 6) <= (int64_t)kS16CapPW, "clica_mlp_pack_split16_both: the weights exceed the state's %u pack-wave slots", kS16CapPW);
  return CLICA_OK;
}
extern "C" int clica_mlp_pack_split16_both(int32_t n_layers, const float* const* W, const int64_t* ldw, const int32_t* N, const int32_t* K,
                                           void* packed_fwd, void* packed_bwd, void* state, clica_stream_t stream) {
  fmlp::Pack2Args a;
  int rc = pack2_fill(a, n_layers, W, ldw, N, K, packed_fwd, packed_bwd, state);
  if (rc) return rc;
  hipLaunchKernelGGL(fmlp::mlp_pack2_k, dim3((unsigned)ceil_div(a.first[a.nseg], 256)), dim3(256), 0, as_stream(stream), a);
  return launch_status("clica_mlp_pack_split16_both");
}
extern "C" int clica_mlp_pack_split16_both_sample(int32_t n_layers, const float* const* W, const int64_t* ldw, const int32_t* N, const int32_t* K,
                                                  void* packed_fwd, void* packed_bwd, void* state,
                                                  const clica_sampler_desc* marginal, const clica_sampler_desc* conditional,
                                                  const float* marginal_mean, int64_t ldmm, float* z, int64_t ldz, float* zt, int64_t ldzt,
                                                  int64_t M, const int32_t* step_dev, clica_stream_t stream) {
  fmlp::Pack2Args a;
  int rc = pack2_fill(a, n_layers, W, ldw, N, K, packed_fwd, packed_bwd, state);
  if (rc) return rc;
  clica::rng::PairArgs pa;
  int one = 0;
  rc = clica_sample_pair_args(marginal, conditional, marginal_mean, ldmm, z, ldz, zt, ldzt, M, step_dev, &pa, &one);
  if (rc) return rc;
  const unsigned pack_blocks = (unsigned)ceil_div(a.first[a.nseg], 256);
  if (!one) {      // row-wise sampler kinds (sphere, vMF): the two calls one after the other
    hipLaunchKernelGGL(fmlp::mlp_pack2_k, dim3(pack_blocks), dim3(256), 0, as_stream(stream), a);
    rc = launch_status("clica_mlp_pack_split16_both_sample");
    if (rc) return rc;
    return clica_sample_pair(marginal, conditional, marginal_mean, ldmm, z, ldz, zt, ldzt, M, step_dev, stream);
  }
  const unsigned sample_blocks = (unsigned)ceil_div(M * marginal->n, 256);
  hipLaunchKernelGGL(fmlp::mlp_pack2_sample_k, dim3(pack_blocks + sample_blocks), dim3(256), 0, as_stream(stream), a, pa, pack_blocks);
  return launch_status("clica_mlp_pack_split16_both_sample");
}

