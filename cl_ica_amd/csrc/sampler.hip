// On-device latent samplers (Philox4x32-10, counter-based: no state, no host sync, graph-replayable)
// replacing the reference's host RNG + rejection loops:
//   box    uniform / truncated conditional   /root/reference/spaces.py:273-351 + spaces_utils.py:106-142
//   sphere uniform / projected conditional   spaces.py:134-231
//   R^n    normal / laplace / gen. normal    spaces.py:44-119, spaces_utils.py:82-103
//   von Mises-Fisher (Wood's rejection)      spaces.py:233-257 -> vmf.py:48-134
// The reference's truncated_rejection_resampling redraws only the out-of-range ELEMENTS each round
// (with a host sync per round); that is per-element rejection sampling, done here in a register loop.
// RNG streams cannot match torch/NumPy bit-for-bit: parity is distributional (tests/test_gpu_samplers).
#include "common.h"
#include "sampler_dev.h"

namespace clica {
namespace rng {
// one thread per ELEMENT for the coordinate-wise kinds (box, R^n): B*n threads instead of B
__global__ __launch_bounds__(THREADS) void sample_elem_k(Desc d, const float* __restrict__ mean, int64_t ldm,
                                                        float* __restrict__ out, int64_t ldo, int64_t M,
                                                        const int32_t* __restrict__ step_dev) {
  const int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  if (idx >= M * d.n) return;
  const int64_t i = idx / d.n;
  const int k = (int)(idx - i * d.n);
  const uint32_t step = step_dev ? (uint32_t)step_dev[0] : 0u;
  Philox g(d.seed, (uint32_t)idx, step, d.stream_id);
  float v;
  if (d.dist == CLICA_DIST_UNIFORM) {                        // spaces.py:273-277
    v = g.uniform() * (d.box_max - d.box_min) + d.box_min;
  } else {
    const float m = mean[i * ldm + k], sc = scale_of(d, i, k);
    v = m + noise(g, d, sc);
    if (d.space == CLICA_SPACE_BOX)                          // spaces.py:279-351: truncate per element
      for (int it = 0; it < 4096 && !(v >= d.box_min && v <= d.box_max); ++it) v = m + noise(g, d, sc);
  }
  out[i * ldo + k] = v;
}

// marginal draw z and conditional draw z~ | z of the same element in one launch (both coordinate-wise kinds):
// the same Philox counters as two sample_elem_k launches, hence the same numbers
__global__ __launch_bounds__(THREADS) void sample_pair_elem_k(PairArgs a) {
  sample_pair_elem(a, (int64_t)blockIdx.x * THREADS + threadIdx.x);
}

// one thread per sample row
__global__ __launch_bounds__(THREADS) void sample_k(Desc d, const float* __restrict__ mean, int64_t ldm,
                                                   float* __restrict__ out, int64_t ldo, int64_t M,
                                                   const int32_t* __restrict__ step_dev) {
  const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  if (i >= M) return;
  const uint32_t step = step_dev ? (uint32_t)step_dev[0] : 0u;
  Philox g(d.seed, (uint32_t)i, step, d.stream_id);
  const int n = d.n;
  float* o = out + i * ldo;
  const float* mu = mean ? mean + i * ldm : nullptr;   // ldm == 0 broadcasts one row

  if (d.dist == CLICA_DIST_VMF) {
    // Wood (1994) / Ulrich rejection for w = <x, mu>  (vmf.py:88-114), then a tangent direction (vmf.py:125-134)
    const float dim = (float)(n - 1), kappa = d.scale;
    const float b = dim / (sqrtf(4.f * kappa * kappa + dim * dim) + 2.f * kappa);
    const float x0 = (1.f - b) / (1.f + b);
    const float c = kappa * x0 + dim * logf(1.f - x0 * x0);
    float w = 1.f;
    for (int it = 0; it < 1000; ++it) {
      const float g1 = g.gamma(0.5f * dim), g2 = g.gamma(0.5f * dim);
      const float z = g1 / (g1 + g2);                       // Beta(d/2, d/2)
      w = (1.f - (1.f + b) * z) / (1.f - (1.f - b) * z);
      const float u = g.uniform_open();
      if (kappa * w + dim * logf(1.f - x0 * w) - c >= logf(u)) break;
    }
    float dot = 0.f, mm = 0.f;
    for (int k = 0; k < n; ++k) { const float v = g.normal(); o[k] = v; dot += v * mu[k]; mm += mu[k] * mu[k]; }
    const float coef = dot / sqrtf(mm);                      // mu * <mu,v> / |mu|  (vmf.py:128-132)
    float ss = 0.f;
    for (int k = 0; k < n; ++k) { const float t = o[k] - mu[k] * coef; o[k] = t; ss += t * t; }
    const float sc = sqrtf(fmaxf(1.f - w * w, 0.f)) / sqrtf(ss);
    for (int k = 0; k < n; ++k) o[k] = o[k] * sc + w * mu[k];
    return;
  }

  if (d.space == CLICA_SPACE_BOX) {
    if (d.dist == CLICA_DIST_UNIFORM) {                      // spaces.py:273-277
      const float span = d.box_max - d.box_min;
      for (int k = 0; k < n; ++k) o[k] = g.uniform() * span + d.box_min;
    } else {                                                 // spaces.py:279-351: truncate per element
      for (int k = 0; k < n; ++k) {
        const float m = mu[k], sc = scale_of(d, i, k);
        float v = m + noise(g, d, sc);
        for (int it = 0; it < 4096 && !(v >= d.box_min && v <= d.box_max); ++it) v = m + noise(g, d, sc);
        o[k] = v;
      }
    }
    return;
  }

  // sphere / R^n: mean + noise, sphere projects back (the radius is ignored by the reference,
  // spaces.py:134-138,168: always unit norm)
  float ss = 0.f;
  for (int k = 0; k < n; ++k) {
    float v;
    if (d.dist == CLICA_DIST_UNIFORM) v = g.normal();        // spaces.py:134-138
    else v = mu[k] + noise(g, d, scale_of(d, i, k));
    o[k] = v; ss += v * v;
  }
  if (d.space == CLICA_SPACE_SPHERE) {
    const float inv = 1.f / sqrtf(ss);
    for (int k = 0; k < n; ++k) o[k] *= inv;
  }
}
}  // namespace rng
}  // namespace clica

using namespace clica;

static bool rowwise_kind(const clica_sampler_desc* d) { return d->space == CLICA_SPACE_SPHERE || d->dist == CLICA_DIST_VMF; }

int clica_sample_pair_args(const clica_sampler_desc* marginal, const clica_sampler_desc* conditional, const float* marginal_mean, int64_t ldmm,
                           float* z, int64_t ldz, float* zt, int64_t ldzt, int64_t M, const int32_t* step_dev,
                           clica::rng::PairArgs* out, int* mergeable) {
  CLICA_CHECK_ARG(marginal && conditional && z && zt && M > 0 && out && mergeable, "clica_sample_pair: bad argument");
  CLICA_CHECK_ARG(marginal->n == conditional->n, "clica_sample_pair: the two descriptors disagree on n");
  CLICA_CHECK_ARG(conditional->dist != CLICA_DIST_UNIFORM, "clica_sample_pair: the second draw must be a conditional kind");
  *mergeable = 0;
  if (rowwise_kind(marginal) || rowwise_kind(conditional) || (marginal->dist != CLICA_DIST_UNIFORM && ldmm == 0)) return CLICA_OK;     // row-wise kinds: two launches
  // validate through the single-draw entry's rules without launching twice
  const clica_sampler_desc* both[2] = {marginal, conditional};
  for (const clica_sampler_desc* d : both) {
    CLICA_CHECK_ARG(d->n >= 1 && ldz >= d->n && ldzt >= d->n, "clica_sample_pair: leading dimension < n");
    CLICA_CHECK_ARG(d->space >= CLICA_SPACE_REAL && d->space <= CLICA_SPACE_SPHERE, "clica_sample_pair: unknown space %d", d->space);
    CLICA_CHECK_ARG(d->dist >= CLICA_DIST_UNIFORM && d->dist <= CLICA_DIST_GENNORM, "clica_sample_pair: unknown distribution %d", d->dist);
    if (d->dist == CLICA_DIST_UNIFORM) CLICA_CHECK_ARG(d->space != CLICA_SPACE_REAL, "clica_sample_pair: uniform is not defined on R^n");
    if (d->dist == CLICA_DIST_GENNORM) CLICA_CHECK_ARG(d->shape_p > 0.f, "clica_sample_pair: generalized normal needs shape_p > 0");
    if (d->space == CLICA_SPACE_BOX) CLICA_CHECK_ARG(d->box_max > d->box_min, "clica_sample_pair: empty box");
  }
  if (marginal->dist != CLICA_DIST_UNIFORM) CLICA_CHECK_ARG(marginal_mean != nullptr && ldmm >= marginal->n, "clica_sample_pair: marginal mean missing");
  out->dm = rng::Desc{marginal->space, marginal->dist, marginal->n, marginal->box_min, marginal->box_max, marginal->scale, marginal->shape_p,
                      marginal->seed, marginal->stream_id, nullptr, 0};
  out->dc = rng::Desc{conditional->space, conditional->dist, conditional->n, conditional->box_min, conditional->box_max, conditional->scale,
                      conditional->shape_p, conditional->seed, conditional->stream_id, nullptr, 0};
  out->mmean = marginal_mean; out->ldmm = ldmm; out->z = z; out->ldz = ldz; out->zt = zt; out->ldzt = ldzt; out->M = M; out->step_dev = step_dev;
  *mergeable = 1;
  return CLICA_OK;
}

extern "C" int clica_sample_pair(const clica_sampler_desc* marginal, const clica_sampler_desc* conditional,
                                 const float* marginal_mean, int64_t ldmm, float* z, int64_t ldz, float* zt, int64_t ldzt,
                                 int64_t M, const int32_t* step_dev, clica_stream_t stream) {
  rng::PairArgs pa;
  int one = 0;
  int rc = clica_sample_pair_args(marginal, conditional, marginal_mean, ldmm, z, ldz, zt, ldzt, M, step_dev, &pa, &one);
  if (rc) return rc;
  if (!one) {
    rc = clica_sample(marginal, marginal_mean, ldmm, z, ldz, M, step_dev, stream);     // row-wise kinds: two launches
    if (rc) return rc;
    return clica_sample(conditional, z, ldz, zt, ldzt, M, step_dev, stream);
  }
  hipLaunchKernelGGL(rng::sample_pair_elem_k, dim3((unsigned)ceil_div(M * marginal->n, rng::THREADS)), dim3(rng::THREADS), 0,
                     as_stream(stream), pa);
  return launch_status("clica_sample_pair");
}

extern "C" int clica_sample(const clica_sampler_desc* d, const float* mean, int64_t ldm,
                            float* out, int64_t ldo, int64_t M, const int32_t* step_dev,
                            clica_stream_t stream) {
  return clica_sample_scaled(d, mean, ldm, nullptr, 0, out, ldo, M, step_dev, stream);
}

extern "C" int clica_sample_scaled(const clica_sampler_desc* d, const float* mean, int64_t ldm,
                                   const float* scale_vec, int64_t lds,
                                   float* out, int64_t ldo, int64_t M, const int32_t* step_dev,
                                   clica_stream_t stream) {
  CLICA_CHECK_ARG(d && out && M > 0, "clica_sample: bad argument");
  if (scale_vec)
    CLICA_CHECK_ARG((lds == 0 || lds >= d->n) && d->dist >= CLICA_DIST_NORMAL && d->dist <= CLICA_DIST_GENNORM,
                    "clica_sample_scaled: a per-coordinate scale needs a location-scale kind (normal / laplace / gennorm) and lds = 0 or >= n");
  CLICA_CHECK_ARG(d->n >= 1 && ldo >= d->n, "clica_sample: n=%d ldo=%lld", d->n, (long long)ldo);
  CLICA_CHECK_ARG(d->space >= CLICA_SPACE_REAL && d->space <= CLICA_SPACE_SPHERE, "clica_sample: unknown space %d", d->space);
  CLICA_CHECK_ARG(d->dist >= CLICA_DIST_UNIFORM && d->dist <= CLICA_DIST_VMF, "clica_sample: unknown distribution %d", d->dist);
  if (d->dist == CLICA_DIST_UNIFORM)
    CLICA_CHECK_ARG(d->space != CLICA_SPACE_REAL, "clica_sample: uniform is not defined on R^n (spaces.py:44-45)");
  else
    CLICA_CHECK_ARG(mean != nullptr && (ldm == 0 || ldm >= d->n), "clica_sample: conditional kinds need `mean`");
  if (d->dist == CLICA_DIST_VMF)
    CLICA_CHECK_ARG(d->space == CLICA_SPACE_SPHERE && d->n >= 2 && d->scale > 0.f, "clica_sample: vMF needs the sphere, n >= 2, kappa > 0");
  if (d->dist == CLICA_DIST_GENNORM) CLICA_CHECK_ARG(d->shape_p > 0.f, "clica_sample: generalized normal needs shape_p > 0");
  if (d->space == CLICA_SPACE_BOX) CLICA_CHECK_ARG(d->box_max > d->box_min, "clica_sample: empty box");
  rng::Desc q{d->space, d->dist, d->n, d->box_min, d->box_max, d->scale, d->shape_p, d->seed, d->stream_id, scale_vec, lds};
  const bool rowwise = d->space == CLICA_SPACE_SPHERE || d->dist == CLICA_DIST_VMF;   // need the row norm
  if (rowwise)
    hipLaunchKernelGGL(rng::sample_k, dim3((unsigned)ceil_div(M, rng::THREADS)), dim3(rng::THREADS), 0, as_stream(stream),
                       q, mean, ldm, out, ldo, M, step_dev);
  else
    hipLaunchKernelGGL(rng::sample_elem_k, dim3((unsigned)ceil_div(M * d->n, rng::THREADS)), dim3(rng::THREADS), 0,
                       as_stream(stream), q, mean, ldm, out, ldo, M, step_dev);
  return launch_status("clica_sample");
}
