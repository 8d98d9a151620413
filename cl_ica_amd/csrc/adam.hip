// Fused Adam over one flat parameter arena: torch.optim.Adam(lr, betas, eps) exactly as the
// reference constructs it (/root/reference/main_mlp.py:312; no weight decay, no amsgrad):
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2
//   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// HBM-bound: 16 B read + 12 B written per parameter, one launch for all layers.
#include "common.h"
#include "split16.h"
#include "adam_math.h"

namespace clica {
namespace adam {
constexpr int THREADS = 256;

__global__ __launch_bounds__(THREADS) void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                 float* __restrict__ v, int64_t count, float lr, float b1, float b2, float eps,
                                                 float gscale, int32_t* __restrict__ step_dev, int32_t* __restrict__ ticket, int t_offset,
                                                 s16::Split16State* s16_state, int s16_layers) {
  // f16x2 encoder arithmetic: the scale update of the step (split16.h) rides in this launch as its FIRST kS16UpdateBlocks workgroups,
  // one tensor each -- every producer of the step has finished by stream order, and the next step's pack launch (the first reader of
  // the new scales) comes behind it
  const unsigned nupd = s16_state ? (unsigned)s16::kS16UpdateBlocks : 0u;
  // (with a fused tick the counter has NOT been advanced yet: a withheld step has nothing to take back, and -- its workgroups returning
  //  below -- nobody advances it)
  if (blockIdx.x < nupd) { s16::split16_update_tensor(s16_state, s16_layers, (int)blockIdx.x, ticket ? nullptr : step_dev); return; }
  const unsigned nwork = gridDim.x - nupd, block = blockIdx.x - nupd;
  // the guard (split16.h): a producer of this step cut a tensor that had outgrown its scale -- parameters and moments stay as they are,
  // the update above takes the step counter back and the next replay redoes the step on the scales this one measured
  if (s16_state && s16::s16_step_poisoned(s16_state)) return;
  __shared__ Consts s_c;
  if (threadIdx.x == 0) s_c = consts_of(step_dev[0] + t_offset, lr, b1, b2);
  __syncthreads();
  const Consts c = s_c;
  const int64_t n4 = count / 4;
  const int64_t stride = (int64_t)nwork * THREADS;
  float4* p4 = reinterpret_cast<float4*>(p); const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m); float4* v4 = reinterpret_cast<float4*>(v);
  for (int64_t i = (int64_t)block * THREADS + threadIdx.x; i < n4; i += stride) {
    float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
    float* pa = &pp.x; float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) update(pa[u], ga[u], ma[u], va[u], c, b1, b2, eps, gscale);
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
  }
  for (int64_t i = n4 * 4 + (int64_t)block * THREADS + threadIdx.x; i < count; i += stride) {
    float pn = p[i], mn = m[i], vn = v[i];
    update(pn, g[i], mn, vn, c, b1, b2, eps, gscale);
    m[i] = mn; v[i] = vn; p[i] = pn;
  }
  // optional fused tick: every workgroup read *step_dev on entry (same thread, earlier in program order than its
  // arrival below), so the LAST one to arrive may advance it.  Relaxed arrival counter, NO fence: nothing but that
  // read has to be ordered, and an agent-scope release here would write back the XCD's dirty L2 lines -- the
  // parameter update itself -- once per workgroup (measured: +20 us).
  if (ticket) {
    __syncthreads();
    if (threadIdx.x == 0) {
      if (__hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)nwork - 1) {
        step_dev[0] += 1;
        __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
}  // namespace adam
}  // namespace clica

using namespace clica;

static int adam_launch(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                       float lr, float beta1, float beta2, float eps, float grad_scale,
                       int32_t* step_dev, int32_t* ticket, int t_offset, clica_stream_t stream, void* s16_state = nullptr, int s16_layers = 0) {
  CLICA_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && step_dev && count > 0, "clica_adam_step: bad argument");
  CLICA_CHECK_ARG(((uintptr_t)param % 16 == 0) && ((uintptr_t)grad % 16 == 0) && ((uintptr_t)exp_avg % 16 == 0) && ((uintptr_t)exp_avg_sq % 16 == 0),
                  "clica_adam_step: arenas must be 16-byte aligned");
  int64_t blocks = ceil_div(ceil_div(count, 4), adam::THREADS);
  if (blocks > kNumCU * 8) blocks = kNumCU * 8;
  // fused tick: every workgroup ends with one agent-scope atomic on the SAME word -- 832 of them serialise to ~8 us behind a 7 us update
  // (measured, 852 k parameters); one workgroup per CU walks the arena with the grid stride instead
  if (ticket && blocks > kNumCU) blocks = kNumCU;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam::adam_k, dim3((unsigned)blocks + (s16_state ? (unsigned)s16::kS16UpdateBlocks : 0u)), dim3(adam::THREADS), 0, as_stream(stream),
                     param, grad, exp_avg, exp_avg_sq, count, lr, beta1, beta2, eps, grad_scale, step_dev, ticket, t_offset,
                     reinterpret_cast<s16::Split16State*>(s16_state), s16_layers);
  return launch_status("clica_adam_step");
}

extern "C" int clica_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                               float lr, float beta1, float beta2, float eps, float grad_scale,
                               const int32_t* step_dev, clica_stream_t stream) {
  return adam_launch(param, grad, exp_avg, exp_avg_sq, count, lr, beta1, beta2, eps, grad_scale, const_cast<int32_t*>(step_dev), nullptr, 1, stream);
}

extern "C" int clica_adam_step_tick(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                                    float lr, float beta1, float beta2, float eps, float grad_scale,
                                    int32_t* step_dev, int32_t* ticket, clica_stream_t stream) {
  CLICA_CHECK_ARG(ticket != nullptr, "clica_adam_step_tick: ticket is NULL");
  return adam_launch(param, grad, exp_avg, exp_avg_sq, count, lr, beta1, beta2, eps, grad_scale, step_dev, ticket, 1, stream);
}

extern "C" int clica_adam_step_at(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                                  float lr, float beta1, float beta2, float eps, float grad_scale,
                                  const int32_t* step_dev, int32_t t_offset, clica_stream_t stream) {
  CLICA_CHECK_ARG(t_offset == 0 || t_offset == 1, "clica_adam_step_at: t_offset must be 0 or 1");
  return adam_launch(param, grad, exp_avg, exp_avg_sq, count, lr, beta1, beta2, eps, grad_scale, const_cast<int32_t*>(step_dev), nullptr, t_offset, stream);
}

// clica_adam_step_tick + the f16x2 encoder arithmetic's scale update in the same launch (the drop-in optimizer's step: update number
// *step_dev + 1, counter advanced by the launch's last workgroup -- unless the guard withheld the step)
extern "C" int clica_adam_step_s16_tick(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                                        float lr, float beta1, float beta2, float eps, float grad_scale,
                                        int32_t* step_dev, int32_t* ticket, void* split16_state, int32_t n_layers, clica_stream_t stream) {
  CLICA_CHECK_ARG(ticket != nullptr, "clica_adam_step_s16_tick: ticket is NULL");
  CLICA_CHECK_ARG(split16_state && n_layers >= 1 && n_layers <= 8, "clica_adam_step_s16_tick: bad state / layer count");
  return adam_launch(param, grad, exp_avg, exp_avg_sq, count, lr, beta1, beta2, eps, grad_scale, step_dev, ticket, 1, stream, split16_state, n_layers);
}

// clica_adam_step_at + the f16x2 encoder arithmetic's scale update (clica_split16_update) in the same launch
extern "C" int clica_adam_step_s16(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                                   float lr, float beta1, float beta2, float eps, float grad_scale,
                                   const int32_t* step_dev, int32_t t_offset, void* split16_state, int32_t n_layers, clica_stream_t stream) {
  CLICA_CHECK_ARG(t_offset == 0 || t_offset == 1, "clica_adam_step_s16: t_offset must be 0 or 1");
  CLICA_CHECK_ARG(split16_state && n_layers >= 1 && n_layers <= 8, "clica_adam_step_s16: bad state / layer count");
  return adam_launch(param, grad, exp_avg, exp_avg_sq, count, lr, beta1, beta2, eps, grad_scale, const_cast<int32_t*>(step_dev), nullptr, t_offset, stream,
                     split16_state, n_layers);
}
