// The guard words of the p = 2 matrix-core loss sweeps (lp_mfma.h explains them) and what a gated kernel reads -- in a header of
// its own because the encoder's backward chain (fused_mlp.hip) reads the same words when it sums the sweep's partials itself.
#pragma once
namespace clica {
namespace lp2 {
constexpr int W_RUN_M = 0, W_M64 = 2, W_V64 = 4, W_CALL = 6, W_FALLBACKS = 7, W_MAXABS_CUR = 8, W_MAXABS_NEXT = 9,
              W_ORIGIN_CUR = 16, W_ORIGIN_NEXT = 32, W_ORIGIN_USED = 48;
// what a gated kernel reads: {M64, V64} -> fall back when this call's M exceeds the limit or a row violated the grid.  Every prep
// workgroup sends its M with the call's tag, so after prep the tag of W_M64 is this call's id; W_V64 is only written on a violation
// and counts when it carries the same tag.  (The id is 32 bits: a workspace is good for 4.29e9 calls.)
__device__ __forceinline__ bool guard_violated(const float* words) {
  const unsigned long long* w64 = reinterpret_cast<const unsigned long long*>(words);
  const unsigned long long m64 = w64[W_M64 / 2], v64 = w64[W_V64 / 2];
  return (v64 >> 32) == (m64 >> 32) && (v64 & 1ull) != 0ull;
}
__device__ __forceinline__ bool guard_falls_back(const float* words, float limit) {
  const unsigned long long* w64 = reinterpret_cast<const unsigned long long*>(words);
  const unsigned long long m64 = w64[W_M64 / 2], v64 = w64[W_V64 / 2];
  const float m = __uint_as_float((unsigned)m64);
  return m > limit || ((v64 >> 32) == (m64 >> 32) && (v64 & 1ull) != 0ull);
}
}  // namespace lp2
}  // namespace clica
