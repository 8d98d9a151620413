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
  // (the maxima are NOT gathered by global atomics: 2 048 same-address device-scope atomics per layer cost 60 us per launch,
  //  measured; every producer workgroup / pack wave leaves its maxima in slots of its own behind this header and the update
  //  reduces them)
  unsigned nA, nD, nPW;    // producer slots written since the last update: forward workgroups, chain workgroups, pack waves (0: not produced)
  unsigned capWG, capPW;   // capacities of the slot arrays (= kS16CapWG / kS16CapPW; informational)
  unsigned wfirst[NT + 1]; // pack waves [wfirst[l], wfirst[l + 1]) hold the maxima of layer l's weights (written by the pack launch)
  unsigned pad0[NT * 3 - 5 - (NT + 1)];
  float sA[NT], sD[NT], sW[NT], sWC[NT];
  unsigned flags;          // bit 0: a scaled magnitude passed kF16Alarm (results of that launch are not to be trusted)
  unsigned updates;        // number of scale updates so far
  unsigned pad[2];
  float pA[NT], pD[NT];    // the scales the LAST step ran with (kept by the update: what its plane copies are scaled by; inspection)
};
constexpr unsigned kS16CapWG = 4096;       // producer workgroups of a launch (48 rows each: batches up to 196 608 rows)
constexpr unsigned kS16CapPW = 16384;      // pack waves (512 weights each)
// slot arrays behind the header, TENSOR-major (the update's threads read consecutive workgroups of one tensor: coalesced, all loads in
// flight at once): partA[NT][capWG], partD[NT][capWG] (true-unit maxima as float bits), partW[capPW] (one per pack wave)
__device__ __host__ inline unsigned* s16_partA(Split16State* st) { return reinterpret_cast<unsigned*>(st + 1); }
__device__ __host__ inline unsigned* s16_partD(Split16State* st) { return s16_partA(st) + (size_t)kS16CapWG * Split16State::NT; }
__device__ __host__ inline unsigned* s16_partW(Split16State* st) { return s16_partD(st) + (size_t)kS16CapWG * Split16State::NT; }
constexpr size_t kS16StateBytes = sizeof(Split16State) + (size_t)kS16CapWG * Split16State::NT * 4 * 2 + (size_t)kS16CapPW * 4;

// Maxima of the step that has just run -> scales of the next one.  s = 2^(kF16Target - floor(log2 max)); a tensor nobody wrote
// (maximum 0) keeps its scale.  Activation scales stay inside fp16's normal range (the constant-1 feature of a plane copy is stored
// as the value s).  One workgroup of 256 threads: it reduces the producers' slot arrays (no global atomics anywhere; every thread has
// all its loads in flight before the first use -- as a chain of dependent loads + LDS atomics this body took 17 us, longer than the
// optimizer launch it rides in) and raises the overflow flag when the step carried a scaled magnitude beyond kF16Alarm.
__device__ __forceinline__ void split16_update_body(Split16State* st, int L) {
  constexpr int NT = Split16State::NT;
  __shared__ unsigned mx[3][NT];
  const int t = threadIdx.x, lane = t & 63;
  if (t < 3 * NT) mx[t / NT][t % NT] = 0u;
  // everything the header holds is requested FIRST (counts, wave ranges, this thread's scales): the state sits in memory another XCD
  // wrote, every dependent read is a ~2 us round trip, and this body rides in a 7 us launch
  const int tc = t < NT ? t : 0;
  const float curA = st->sA[tc], curD = st->sD[tc], curW = st->sW[tc];
  const unsigned nA = min(st->nA, kS16CapWG), nD = min(st->nD, kS16CapWG), nPW = min(st->nPW, kS16CapPW);
  unsigned wf[NT + 1];
#pragma unroll
  for (int l = 0; l <= NT; ++l) wf[l] = st->wfirst[l];
  const unsigned* pA = s16_partA(st); const unsigned* pD = s16_partD(st); const unsigned* pW = s16_partW(st);
  unsigned ma[NT], md[NT], mw[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) { ma[k] = 0u; md[k] = 0u; mw[k] = 0u; }
  for (unsigned w0 = 0; w0 < max(nA, nD); w0 += 256) {          // one pass for batches up to 12 288 rows
    const unsigned w = w0 + t;
    unsigned va[NT], vd[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) { va[k] = w < nA ? pA[(size_t)k * kS16CapWG + w] : 0u; vd[k] = w < nD ? pD[(size_t)k * kS16CapWG + w] : 0u; }
#pragma unroll
    for (int k = 0; k < NT; ++k) { ma[k] = max(ma[k], va[k]); md[k] = max(md[k], vd[k]); }
  }
  for (unsigned i0 = 0; i0 < nPW; i0 += 256 * 8) {
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
  if (t <= L) { st->pA[t] = curA; st->sA[t] = next(mx[0][t], curA, -14, 15); }
  if (t < L) { st->pD[t] = curD; st->sD[t] = next(mx[1][t], curD, -100, 100); }
  float w = 1.f;
  if (t < L) { w = next(mx[2][t], curW, -100, 100); st->sW[t] = w; }
  if (t < L && L - 1 - t >= 0 && L - 1 - t < NT) st->sWC[L - 1 - t] = w;      // chain link j uses layer L - 1 - j
  if (t == 0) { st->updates += 1u; st->nA = st->nD = st->nPW = 0u; }
}
}  // namespace s16
}  // namespace clica
