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
  //  measured; every producer workgroup / pack wave leaves its maxima in a slot of its own behind this header and the update
  //  kernel reduces them)
  unsigned nA, nD, nPW;    // producer slots written since the last update: forward workgroups, chain workgroups, pack waves (0: not produced)
  unsigned capWG, capPW;   // capacities of the slot arrays (= kS16CapWG / kS16CapPW; informational)
  unsigned pad0[NT * 3 - 5];
  float sA[NT], sD[NT], sW[NT], sWC[NT];
  unsigned flags;          // bit 0: a scaled magnitude passed kF16Alarm (results of that launch are not to be trusted)
  unsigned updates;        // number of scale updates so far
  unsigned pad[2];
  float pA[NT], pD[NT];    // the scales the LAST step ran with (kept by the update: what its plane copies are scaled by; inspection)
};
constexpr unsigned kS16CapWG = 4096;       // producer workgroups of a launch (48 rows each: batches up to 196 608 rows)
constexpr unsigned kS16CapPW = 16384;      // pack waves (512 weights each)
// slot arrays behind the header: partA[capWG][NT], partD[capWG][NT] (true-unit maxima as float bits), partW[capPW] (value), partWl[capPW] (layer, -1: none)
__device__ __host__ inline unsigned* s16_partA(Split16State* st) { return reinterpret_cast<unsigned*>(st + 1); }
__device__ __host__ inline unsigned* s16_partD(Split16State* st, unsigned capWG = kS16CapWG) { return s16_partA(st) + (size_t)capWG * Split16State::NT; }
__device__ __host__ inline unsigned* s16_partW(Split16State* st, unsigned capWG = kS16CapWG) { return s16_partD(st, capWG) + (size_t)capWG * Split16State::NT; }
__device__ __host__ inline int* s16_partWl(Split16State* st, unsigned capWG = kS16CapWG, unsigned capPW = kS16CapPW) { return reinterpret_cast<int*>(s16_partW(st, capWG) + capPW); }

// Maxima of the step that has just run -> scales of the next one.  s = 2^(kF16Target - floor(log2 max)); a tensor nobody wrote
// (maximum 0) keeps its scale.  Activation scales stay inside fp16's normal range (the constant-1 feature of a plane copy is stored
// as the value s).  One workgroup: it also reduces the producers' slot arrays (no global atomics anywhere) and raises the overflow flag
// when the step that has just run carried a scaled magnitude beyond kF16Alarm.
__device__ __forceinline__ void split16_update_body(Split16State* st, int L) {
  constexpr int NT = Split16State::NT;
  __shared__ unsigned mx[3][NT];
  const int t = threadIdx.x;
  if (t < 3 * NT) mx[t / NT][t % NT] = 0u;
  __syncthreads();
  const unsigned nA = min(st->nA, kS16CapWG), nD = min(st->nD, kS16CapWG), nPW = min(st->nPW, kS16CapPW);
  const unsigned* pA = s16_partA(st); const unsigned* pD = s16_partD(st);
  const unsigned* pW = s16_partW(st); const int* pWl = s16_partWl(st);
  for (unsigned i = t; i < nA * NT; i += 256) atomicMax(&mx[0][i % NT], pA[i]);
  for (unsigned i = t; i < nD * NT; i += 256) atomicMax(&mx[1][i % NT], pD[i]);
  for (unsigned i = t; i < nPW; i += 256) { const int l = pWl[i]; if (l >= 0 && l < NT) atomicMax(&mx[2][l], pW[i]); }
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
  if (t <= L) { st->pA[t] = st->sA[t]; st->sA[t] = next(mx[0][t], st->sA[t], -14, 15); }
  if (t < L) { st->pD[t] = st->sD[t]; st->sD[t] = next(mx[1][t], st->sD[t], -100, 100); }
  float w = 1.f;
  if (t < L) { w = next(mx[2][t], st->sW[t], -100, 100); st->sW[t] = w; }
  if (t < L && L - 1 - t >= 0 && L - 1 - t < NT) st->sWC[L - 1 - t] = w;      // chain link j uses layer L - 1 - j
  if (t == 0) { st->updates += 1u; st->nA = st->nD = st->nPW = 0u; }
}
}  // namespace s16
}  // namespace clica
