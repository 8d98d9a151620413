// The Adam update of one element, shared by the optimizer launch (adam.hip) and the weight-gradient reduction that applies it in its
// own epilogue (linear.hip: slab_reduce_group_k with ReduceGroupArgs::adam).  torch.optim.Adam(lr, betas, eps) exactly as the reference
// constructs it (/root/reference/main_mlp.py:312; no weight decay, no amsgrad):
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
#pragma once
#include "common.h"

namespace clica {
namespace adam {
struct Consts { float step_size, inv_bc2_sqrt; };
// bias corrections in double, like the Python side of torch.optim.Adam (one thread per workgroup calls this)
__device__ __forceinline__ Consts consts_of(int step, float lr, float b1, float b2) {
  const double t = (double)step;
  const double bc1 = 1.0 - pow((double)b1, t);
  const double bc2 = 1.0 - pow((double)b2, t);
  Consts c;
  c.step_size = (float)((double)lr / bc1);
  c.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  return c;
}
// Every operation spelled out with its rounding (no contraction left to the compiler): the update is inlined into two different
// kernels and the data-parallel path (separate launch) must reproduce the single-GPU path (reduction epilogue) bit for bit.
__device__ __forceinline__ void update(float& p, float g, float& m, float& v, const Consts& c, float b1, float b2, float eps, float gscale) {
  const float gr = __fmul_rn(g, gscale);
  m = __fmaf_rn(b1, m, __fmul_rn(1.f - b1, gr));
  v = __fmaf_rn(b2, v, __fmul_rn(__fmul_rn(1.f - b2, gr), gr));
  const float denom = __fmaf_rn(__fsqrt_rn(v), c.inv_bc2_sqrt, eps);
  p = __fsub_rn(p, __fdiv_rn(__fmul_rn(c.step_size, m), denom));
}
}  // namespace adam
}  // namespace clica
