// Device state of the f16x2 split arithmetic (fused_mlp.hip: Arith<1>; wgrad_split.hip: WArith<1>) and the kernel body that turns the
// maxima of one training step into the scales of the next.  Shared by fused_mlp.hip (producers, stand-alone update launch) and adam.hip
// (the update rides in the optimizer launch as one extra workgroup).
#pragma once
#include "common.h"

namespace clica {
namespace s16 {
constexpr int kF16Target = 8;              // scaled maximum of a tensor in [256, 512)
constexpr float kF16Alarm = 32768.f;       // a scaled magnitude beyond this raises the overflow flag (fp16 max 65504)

// Device state of the f16x2 arithmetic of ONE encoder (clica_split16_state_bytes floats): per tensor family and position the running
// maximum of the current step (true units, as uint bits: non-negative floats order like ints) and the scale in force.
//   family A: activations in forward order   A[0] = encoder input x, A[l + 1] = output of layer l
//   family D: gradients in CHAIN order        D[0] = d loss / d (last pre-activation), D[j + 1] = output of chain link j
//   family W: weights, W[l] forward order;  WC[j] = the same scales in chain order (link j uses layer L - 1 - j)
struct Split16State {
  static constexpr int NT = 8 + 1;      // fused_mlp.hip: MAXL + 1
  // (the maxima are NOT gathered by same-address global atomics: 2 048 of them per layer cost 60 us per launch, measured; every
  //  producer leaves its maxima in slots behind this header and the update reduces them)
  unsigned cntA[NT], cntD[NT], cntW[NT];   // live slots per tensor in the tensor-major slot arrays (0: not produced since the last update)
  unsigned nPW, capWG, capPW;              // pack waves written by the LAST whole-stack weight pack (their maxima stay valid until the next pack); capacities (informational)
  unsigned gen_copy;                       // THE GUARD (below): `updates` as the producers of the current step saw it
  unsigned wfirst[NT + 1];                 // pack waves [wfirst[l], wfirst[l + 1]) hold the maxima of layer l's weights
  unsigned pad0;
  const float* dp_poison;  // data parallel: where the all-reduced verdict of the step lives (a float behind the gradient arena: > 0 = some rank's step is
                           // poisoned), or nullptr: this rank decides from its own `poison` word (clica_split16_set_dp_poison)
  float sA[NT], sD[NT], sW[NT], sWC[NT];
  unsigned flags;          // bit 0: a scaled magnitude passed kF16Alarm in some step (sticky);  bit 1: a step was withheld by the guard (sticky);
                           // bit 2: the update saw an overflow no producer had announced (a producer without the guard: a bug)
  unsigned updates;        // number of scale updates so far = the GENERATION of the step whose producers run next
  unsigned poison;         // THE GUARD: generation of the last step in which a producer saw a scaled magnitude beyond kF16Alarm (or a non-finite one)
  unsigned skipped;        // number of steps the guard has withheld
  float pA[NT], pD[NT];    // the scales the LAST step ran with (kept by the update: what its plane copies are scaled by; inspection)
};
// THE GUARD (round 6).  A launch runs on the scales of the PREVIOUS step, so a tensor that grew > 64 x since then overflows fp16 -- and the
// step's results must not reach the parameters.  Every producer knows the scale in force and the fp32 magnitudes it cuts, so the one
// that sees a scaled magnitude beyond kF16Alarm (or a non-finite value) writes the current generation into `poison` (s16_raise_poison;
// plain stores of one value, only in the rare case).  The launch that applies the optimizer (adam_k, slab_reduce_group_k) compares
// `poison` with `gen_copy` -- the generation as the step's producers saw it, refreshed by one thread of every producer launch; neither
// word changes during the optimizer launch -- and, on a match, leaves parameters and moments untouched.  The scale update of the same
// launch (split16_update_tensor, tensor 0) then counts the step as skipped, raises flag bit 1 and takes the step / RNG counter back, so
// the NEXT replay of the step graph redoes the same batch on the scales this step measured: the arithmetic heals itself inside graph
// replay, one link of the chain per redo at worst (as calibrate_scales does from the host), and the host learns about it at its next
// log point (flags, skipped).  Data parallel: the ranks must agree -- the verdict travels as one float behind the gradient arena through
// the gradient all-reduce (clica_split16_poison_export writes it, dp_poison points at it).
__device__ __forceinline__ void s16_raise_poison(Split16State* st) { st->poison = st->updates; }
__device__ __forceinline__ void s16_refresh_gen(Split16State* st) { st->gen_copy = st->updates; }
__device__ __forceinline__ bool s16_step_poisoned(const Split16State* st) {
  const float* dp = st->dp_poison;
  return dp ? (*dp > 0.f) : (st->poison == st->gen_copy);
}
constexpr unsigned kS16CapWG = 4096;       // slots per tensor (whole-stack kernels: one per workgroup of 48 rows, batches up to 196 608 rows;
                                           // per-layer producers: workgroup id modulo kS16SlotsPerLayerKernel, combined by atomicMax)
constexpr unsigned kS16SlotsPerLayerKernel = 1024;
constexpr unsigned kS16CapPW = 16384;      // pack waves (512 weights each)
// slot arrays behind the header, TENSOR-major (the update's threads read consecutive slots of one tensor: coalesced, all loads in
// flight at once): partA[NT][capWG], partD[NT][capWG], partW2[NT][capWG] (true-unit maxima as float bits), partW[capPW] (one per pack wave)
__device__ __host__ inline unsigned* s16_partA(Split16State* st) { return reinterpret_cast<unsigned*>(st + 1); }
__device__ __host__ inline unsigned* s16_partD(Split16State* st) { return s16_partA(st) + (size_t)kS16CapWG * Split16State::NT; }
__device__ __host__ inline unsigned* s16_partW2(Split16State* st) { return s16_partD(st) + (size_t)kS16CapWG * Split16State::NT; }
__device__ __host__ inline unsigned* s16_partW(Split16State* st) { return s16_partW2(st) + (size_t)kS16CapWG * Split16State::NT; }
constexpr size_t kS16StateBytes = sizeof(Split16State) + (size_t)kS16CapWG * Split16State::NT * 4 * 3 + (size_t)kS16CapPW * 4;

// What a per-layer producer (wgrad_split.hip: plane conversions, fused GEMM epilogues) needs of ONE tensor of the state
struct S16Tensor { const float* scale; unsigned* slots; unsigned* count; Split16State* st; };
__host__ inline S16Tensor s16_tensor(Split16State* st, int family /* 0 A, 1 D, 2 W */, int index) {
  S16Tensor t;
  t.scale = family == 0 ? &st->sA[index] : (family == 1 ? &st->sD[index] : &st->sW[index]);
  t.slots = (family == 0 ? s16_partA(st) : (family == 1 ? s16_partD(st) : s16_partW2(st))) + (size_t)index * kS16CapWG;
  t.count = family == 0 ? &st->cntA[index] : (family == 1 ? &st->cntD[index] : &st->cntW[index]);
  t.st = st;
  return t;
}
// one call per WORKGROUP of a per-layer producer (all threads; `red` = 8 floats of shared memory): the workgroup's maximum of |true value|
__device__ __forceinline__ void s16_commit_block_max(const S16Tensor& T, float m_true, float* red, unsigned block_id, unsigned nblocks) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m_true = fmaxf(m_true, __shfl_xor(m_true, off, 64));
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) red[wave] = m_true;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = red[0];
    for (int w = 1; w < nw; ++w) m = fmaxf(m, red[w]);
    m = (m <= 3.0e38f) ? m : 3.4e38f;
    if (m > 0.f) atomicMax(T.slots + (block_id % kS16SlotsPerLayerKernel), __float_as_uint(m));      // (<= nblocks / 1024 workgroups per address)
    if (!(m * *T.scale <= kF16Alarm)) s16_raise_poison(T.st);      // THE GUARD: this workgroup cut a value that does not fit its scale
    if (block_id == 0) { *T.count = nblocks < kS16SlotsPerLayerKernel ? nblocks : kS16SlotsPerLayerKernel; s16_refresh_gen(T.st); }
  }
}

// Maxima of the step that has just run -> scales of the next one.  s = 2^(kF16Target - floor(log2 max)); a tensor nobody wrote
// (maximum 0) keeps its scale.  kS16UpdateBlocks workgroups of 256 threads, ONE TENSOR EACH (family = idx / NT, position = idx % NT):
// a workgroup reads its tensor's count and scale, then all of its slots at once, zeroes the slots it read, and raises the overflow flag
// when the step carried a scaled magnitude beyond kF16Alarm.  (As ONE workgroup walking all 27 tensors this body took 16 us -- every
// step a chain of dependent round trips to memory another XCD wrote -- and the 7 us optimizer launch it rides in took 18.)
constexpr int kS16UpdateBlocks = 3 * Split16State::NT;
// `step_dev`: the step / RNG counter to take back when the guard withheld the step (nullptr: stand-alone update, nothing to take back).
__device__ __forceinline__ void split16_update_tensor(Split16State* st, int L, int idx, int* step_dev = nullptr) {
  constexpr int NT = Split16State::NT;
  __shared__ unsigned red[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fam = idx / NT, k = idx - fam * NT;
  unsigned* cntp = fam == 0 ? &st->cntA[k] : (fam == 1 ? &st->cntD[k] : &st->cntW[k]);
  float* sp = fam == 0 ? &st->sA[k] : (fam == 1 ? &st->sD[k] : &st->sW[k]);
  unsigned* slots = (fam == 0 ? s16_partA(st) : (fam == 1 ? s16_partD(st) : s16_partW2(st))) + (size_t)k * kS16CapWG;
  // header fields first, all requested together
  const unsigned cnt = min(*cntp, kS16CapWG);
  const float cur = *sp;
  unsigned w0 = 0u, w1 = 0u;
  if (fam == 2) { const unsigned nPW = min(st->nPW, kS16CapPW); w0 = min(st->wfirst[k], nPW); w1 = min(st->wfirst[k + 1], nPW); }
  unsigned m = 0u;
  constexpr int U = kS16CapWG / 256;                   // every slot of the tensor in flight at once
  unsigned v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { const unsigned w = (unsigned)t + 256u * u; v[u] = w < cnt ? slots[w] : 0u; }
#pragma unroll
  for (int u = 0; u < U; ++u) { const unsigned w = (unsigned)t + 256u * u; m = max(m, v[u]); if (w < cnt) slots[w] = 0u; }
  if (fam == 2) {                                      // weights packed by the whole-stack path: one maximum per pack wave, layer k's range
    const unsigned* pW = s16_partW(st);
    for (unsigned i0 = w0; i0 < w1; i0 += 256 * 8) {
      unsigned q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const unsigned i = i0 + t + 256u * u; q[u] = i < w1 ? pW[i] : 0u; }
#pragma unroll
      for (int u = 0; u < 8; ++u) m = max(m, q[u]);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  if (t != 0) return;
  const unsigned bits = max(max(red[0], red[1]), max(red[2], red[3]));
  const float a = __uint_as_float(bits);
  float nxt = cur;
  const bool poisoned = s16_step_poisoned(st);         // (neither word changes during this launch)
  if (a > 0.f) {                                       // (nobody wrote the tensor: keep its scale)
    if (!(a < 3.0e38f) || a * cur > kF16Alarm) atomicOr(&st->flags, poisoned ? 1u : 5u);      // non-finite, or the step that just ran overflowed its scale
    if (a < 3.0e38f) {
      int e = (int)((bits >> 23) & 0xffu) - 127;      // floor(log2 a) for normal a
      if (((bits >> 23) & 0xffu) == 0u) e = -127;
      int se = kF16Target - e;
      // (+-60: the weight-gradient kernel divides its slab by the PRODUCT of two scales, which must stay a finite, non-zero fp32 number
      //  -- 2^+-120 -- for tensors of any size; a tensor whose maximum lies beyond 2^68 then overflows its scale and is caught
      //  by the guard, one below 2^-52 keeps fewer than 22 bits)
      se = se < -60 ? -60 : (se > 60 ? 60 : se);
      nxt = __uint_as_float((unsigned)(se + 127) << 23);
    }
  }
  if (fam == 0) st->pA[k] = cur;
  if (fam == 1) st->pD[k] = cur;
  *sp = nxt; *cntp = 0u;
  if (fam == 2 && k < L) st->sWC[L - 1 - k] = nxt;     // chain link j uses layer L - 1 - j
  if (idx == 0) {
    if (poisoned) {                                    // THE GUARD: the optimizer launch this update rides in has left the parameters alone
      atomicOr(&st->flags, 2u);
      st->skipped += 1u;
      if (step_dev) step_dev[0] -= 1;                  // the next replay draws the same batch, Adam's t stays the number of APPLIED steps
    }
    st->updates += 1u;
  }
}
}  // namespace s16
}  // namespace clica
