// Device side of the latent samplers that other translation units launch inside kernels of their own (sampler.hip holds the
// kernels and the C ABI; fused_mlp.hip merges the pair draw into the weight-pack launch of the training step: mlp_pack2_sample_k).
// Philox4x32-10, counter-based: (seed; element index, draw block, step, stream id) -> the same numbers whichever kernel hosts the draw.
#pragma once
#include "common.h"

namespace clica {
namespace rng {
constexpr int THREADS = 256;

struct Philox {
  uint32_t key0, key1;
  uint32_t c0, c1, c2, c3;   // c0 = element index, c1 = draw block, c2 = step, c3 = stream id
  uint32_t o0, o1, o2, o3;   // named registers: a runtime-indexed array would live in scratch memory
  int have;
  __device__ Philox(uint64_t seed, uint32_t idx, uint32_t step, uint32_t stream)
      : key0((uint32_t)seed), key1((uint32_t)(seed >> 32)), c0(idx), c1(0), c2(step), c3(stream), have(0) {}
  __device__ void refill() {
    uint32_t a0 = c0, a1 = c1, a2 = c2, a3 = c3, k0 = key0, k1 = key1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const uint64_t p0 = (uint64_t)0xD2511F53u * a0;
      const uint64_t p1 = (uint64_t)0xCD9E8D57u * a2;
      const uint32_t n0 = (uint32_t)(p1 >> 32) ^ a1 ^ k0;
      const uint32_t n1 = (uint32_t)p1;
      const uint32_t n2 = (uint32_t)(p0 >> 32) ^ a3 ^ k1;
      const uint32_t n3 = (uint32_t)p0;
      a0 = n0; a1 = n1; a2 = n2; a3 = n3;
      k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = a0; o1 = a1; o2 = a2; o3 = a3;
    ++c1; have = 4;
  }
  __device__ uint32_t next() {
    if (have == 0) refill();
    const uint32_t r = o0;
    o0 = o1; o1 = o2; o2 = o3; --have;
    return r;
  }
  __device__ float uniform() { return (float)(next() >> 8) * (1.0f / 16777216.0f); }          // [0,1)
  __device__ float uniform_open() { return ((float)(next() >> 8) + 1.0f) * (1.0f / 16777216.0f); }  // (0,1]
  __device__ float normal() {  // Box-Muller, one value per call (the twin is discarded: draws are cheap)
    const float u1 = uniform_open(), u2 = uniform();
    return sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
  }
  __device__ float laplace() {  // unit-scale Laplace by inverse CDF, as torch.distributions.Laplace.rsample
    const float u = 2.0f * uniform() - 1.0f;   // [-1,1)
    const float a = fabsf(u);
    const float m = -log1pf(-fminf(a, 0.99999994f));
    return u < 0.f ? -m : m;
  }
  __device__ float gamma(float a) {  // Marsaglia-Tsang; shape < 1 via the U^(1/a) boost
    float boost = 1.f;
    if (a < 1.f) { boost = powf(uniform_open(), 1.f / a); a += 1.f; }
    const float d = a - 1.f / 3.f, c = 1.f / sqrtf(9.f * d);
    for (int it = 0; it < 64; ++it) {
      const float x = normal();
      float v = 1.f + c * x;
      if (v <= 0.f) continue;
      v = v * v * v;
      const float u = uniform_open();
      if (logf(u) < 0.5f * x * x + d - d * v + d * logf(v)) return d * v * boost;
    }
    return d * boost;
  }
  __device__ float gennorm(float p) {  // +-Gamma(1/p,1)^(1/p)  (spaces_utils.py:95-102)
    const float gmm = gamma(1.f / p);
    const float mag = powf(gmm, 1.f / p);
    return (next() & 1u) ? mag : -mag;
  }
};

struct Desc {
  int space, dist, n;
  float box_min, box_max, scale, shape_p;
  uint64_t seed; uint32_t stream_id;
  const float* svec; int64_t lds;     // optional per-coordinate scale (spaces.py:60-72: `std` may be a tensor), row stride 0 = one row
};

// scale of coordinate k of row i: the descriptor's scalar, times the per-coordinate tensor when there is one
__device__ __forceinline__ float scale_of(const Desc& d, int64_t i, int k) {
  return d.svec ? d.scale * d.svec[i * d.lds + k] : d.scale;
}
__device__ __forceinline__ float noise(Philox& g, const Desc& d, float sc) {
  switch (d.dist) {
    case CLICA_DIST_NORMAL: return sc * g.normal();
    case CLICA_DIST_LAPLACE: return sc * g.laplace();
    case CLICA_DIST_GENNORM: return sc * g.gennorm(d.shape_p);
    default: return 0.f;
  }
}


// marginal draw z and conditional draw z~ | z of element `idx` (both coordinate-wise kinds): the body of sample_pair_elem_k
struct PairArgs {
  Desc dm, dc;
  const float* mmean; int64_t ldmm;
  float* z; int64_t ldz; float* zt; int64_t ldzt;
  int64_t M; const int32_t* step_dev;
};
__device__ __forceinline__ void sample_pair_elem(const PairArgs& a, const int64_t idx) {
  const Desc& dm = a.dm; const Desc& dc = a.dc;
  if (idx >= a.M * dm.n) return;
  const int64_t i = idx / dm.n;
  const int k = (int)(idx - i * dm.n);
  const uint32_t step = a.step_dev ? (uint32_t)a.step_dev[0] : 0u;
  float v;
  {
    Philox g(dm.seed, (uint32_t)idx, step, dm.stream_id);
    if (dm.dist == CLICA_DIST_UNIFORM) {
      v = g.uniform() * (dm.box_max - dm.box_min) + dm.box_min;
    } else {
      const float m = a.mmean[i * a.ldmm + k];
      v = m + noise(g, dm, dm.scale);
      if (dm.space == CLICA_SPACE_BOX)
        for (int it = 0; it < 4096 && !(v >= dm.box_min && v <= dm.box_max); ++it) v = m + noise(g, dm, dm.scale);
    }
    a.z[i * a.ldz + k] = v;
  }
  {
    Philox g(dc.seed, (uint32_t)idx, step, dc.stream_id);
    const float m = v;
    float w = m + noise(g, dc, dc.scale);
    if (dc.space == CLICA_SPACE_BOX)
      for (int it = 0; it < 4096 && !(w >= dc.box_min && w <= dc.box_max); ++it) w = m + noise(g, dc, dc.scale);
    a.zt[i * a.ldzt + k] = w;
  }
}
}  // namespace rng
}  // namespace clica

// host (sampler.hip): validates a clica_sample_pair call and fills the device arguments of the one-launch pair draw.
// returns CLICA_OK with *mergeable = 1, CLICA_OK with *mergeable = 0 for the kinds that take two row-wise launches, or an error code
int clica_sample_pair_args(const clica_sampler_desc* marginal, const clica_sampler_desc* conditional, const float* marginal_mean, int64_t ldmm,
                           float* z, int64_t ldz, float* zt, int64_t ldzt, int64_t M, const int32_t* step_dev,
                           clica::rng::PairArgs* out, int* mergeable);
