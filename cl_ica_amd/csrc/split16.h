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
  unsigned nPW, capWG, capPW, pad1;        // pack waves written by the whole-stack weight pack; capacities (informational)
  unsigned wfirst[NT + 1];                 // pack waves [wfirst[l], wfirst[l + 1]) hold the maxima of layer l's weights
  unsigned pad0[3];
  float sA[NT], sD[NT], sW[NT], sWC[NT];
  unsigned flags;          // bit 0: a scaled magnitude passed kF16Alarm (results of that launch are not to be trusted)
  unsigned updates;        // number of scale updates so far
  unsigned pad[2];
  float pA[NT], pD[NT];    // the scales the LAST step ran with (kept by the update: what its plane copies are scaled by; inspection)
};
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
struct S16Tensor { const float* scale; unsigned* slots; unsigned* count; };
__host__ inline S16Tensor s16_tensor(Split16State* st, int family /* 0 A, 1 D, 2 W */, int index) {
  S16Tensor t;
  t.scale = family == 0 ? &st->sA[index] : (family == 1 ? &st->sD[index] : &st->sW[index]);
  t.slots = (family == 0 ? s16_partA(st) : (family == 1 ? s16_partD(st) : s16_partW2(st))) + (size_t)index * kS16CapWG;
  t.count = family == 0 ? &st->cntA[index] : (family == 1 ? &st->cntD[index] : &st->cntW[index]);
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
    if (block_id == 0) *T.count = nblocks < kS16SlotsPerLayerKernel ? nblocks : kS16SlotsPerLayerKernel;
  }
}

// Maxima of the step that has just run -> scales of the next one.  s = 2^(kF16Target - floor(log2 max)); a tensor nobody wrote
// (maximum 0) keeps its scale.  One workgroup of 256 threads: it reduces the producers' slot arrays (every thread has all its loads in
// flight before the first use -- as a chain of dependent loads + LDS atomics this body took 17 us, longer than the optimizer launch it
// rides in), zeroes the slots it read and raises the overflow flag when the step carried a scaled magnitude beyond kF16Alarm.
__device__ __forceinline__ void split16_update_body(Split16State* st, int L) {
  constexpr int NT = Split16State::NT;
  __shared__ unsigned mx[3][NT];
  const int t = threadIdx.x, lane = t & 63;
  if (t < 3 * NT) mx[t / NT][t % NT] = 0u;
  // everything the header holds is requested FIRST (counts, wave ranges, this thread's scales): the state sits in memory another XCD
  // wrote, every dependent read is a ~2 us round trip, and this body rides in a 7 us launch
  const int tc = t < NT ? t : 0;
  const float curA = st->sA[tc], curD = st->sD[tc], curW = st->sW[tc];
  unsigned cA[NT], cD[NT], cW[NT], wf[NT + 1];
#pragma unroll
  for (int k = 0; k < NT; ++k) { cA[k] = min(st->cntA[k], kS16CapWG); cD[k] = min(st->cntD[k], kS16CapWG); cW[k] = min(st->cntW[k], kS16CapWG); }
  const unsigned nPW = min(st->nPW, kS16CapPW);
#pragma unroll
  for (int l = 0; l <= NT; ++l) wf[l] = st->wfirst[l];
  unsigned* pA = s16_partA(st); unsigned* pD = s16_partD(st); unsigned* pW2 = s16_partW2(st); const unsigned* pW = s16_partW(st);
  unsigned ma[NT], md[NT], mw[NT];
  unsigned nmax = 0u, nmaxw = 0u;
#pragma unroll
  for (int k = 0; k < NT; ++k) { ma[k] = 0u; md[k] = 0u; mw[k] = 0u; nmax = max(nmax, max(cA[k], cD[k])); nmaxw = max(nmaxw, cW[k]); }
  for (unsigned w0 = 0; w0 < nmax; w0 += 256) {                  // one pass for the whole-stack kernels up to 12 288 rows
    const unsigned w = w0 + t;
    unsigned va[NT], vd[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) { va[k] = w < cA[k] ? pA[(size_t)k * kS16CapWG + w] : 0u; vd[k] = w < cD[k] ? pD[(size_t)k * kS16CapWG + w] : 0u; }
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      ma[k] = max(ma[k], va[k]); md[k] = max(md[k], vd[k]);
      if (w < cA[k]) pA[(size_t)k * kS16CapWG + w] = 0u;
      if (w < cD[k]) pD[(size_t)k * kS16CapWG + w] = 0u;
    }
  }
  for (unsigned w0 = 0; w0 < nmaxw; w0 += 256) {                 // weights converted by the per-layer path
    const unsigned w = w0 + t;
    unsigned vw[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) vw[k] = w < cW[k] ? pW2[(size_t)k * kS16CapWG + w] : 0u;
#pragma unroll
    for (int k = 0; k < NT; ++k) { mw[k] = max(mw[k], vw[k]); if (w < cW[k]) pW2[(size_t)k * kS16CapWG + w] = 0u; }
  }
  for (unsigned i0 = 0; i0 < nPW; i0 += 256 * 8) {               // weights packed by the whole-stack path
    unsigned v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const unsigned i = i0 + t + 256u * u; v[u] = i < nPW ? pW[i] : 0u; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned i = i0 + t + 256u * u;
#pragma unroll
      for (int l = 0; l < NT; ++l) mw[l] = (i >= wf[l] && i < wf[l + 1]) ? max(mw[l], v[u]) : mw[l];
    }
  }
  __syncthreads();                                                 // mx zeroed
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    unsigned a = ma[k], d = md[k], w = mw[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      a = max(a, (unsigned)__shfl_xor((int)a, off, 64)); d = max(d, (unsigned)__shfl_xor((int)d, off, 64)); w = max(w, (unsigned)__shfl_xor((int)w, off, 64));
    }
    if (lane == 0) { atomicMax(&mx[0][k], a); atomicMax(&mx[1][k], d); atomicMax(&mx[2][k], w); }
  }
  __syncthreads();
  auto next = [&](unsigned bits, float cur, int emin, int emax) {
    const float a = __uint_as_float(bits);
    if (!(a > 0.f)) return cur;                                 // nobody wrote the tensor: keep its scale
    if (!(a < 3.0e38f) || a * cur > kF16Alarm) atomicOr(&st->flags, 1u);      // non-finite, or the step that just ran overflowed its scale
    if (!(a < 3.0e38f)) return cur;
    int e = (int)((bits >> 23) & 0xffu) - 127;                  // floor(log2 a) for normal a
    if (((bits >> 23) & 0xffu) == 0u) e = -127;
    int se = kF16Target - e;
    se = se < emin ? emin : (se > emax ? emax : se);
    return __uint_as_float((unsigned)(se + 127) << 23);
  };
  if (t < NT) { st->pA[t] = curA; st->sA[t] = next(mx[0][t], curA, -100, 100); st->cntA[t] = 0u; }
  if (t < NT) { st->pD[t] = curD; st->sD[t] = next(mx[1][t], curD, -100, 100); st->cntD[t] = 0u; }
  float w = 1.f;
  if (t < NT) { w = next(mx[2][t], curW, -100, 100); st->sW[t] = w; st->cntW[t] = 0u; }
  if (t < L && L - 1 - t >= 0 && L - 1 - t < NT) st->sWC[L - 1 - t] = w;      // chain link j uses layer L - 1 - j
  if (t == 0) { st->updates += 1u; st->nPW = 0u; }
}
}  // namespace s16
}  // namespace clica
