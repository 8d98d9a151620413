// Device helpers shared by lp_mfma.hip and the loss finalize (lp_loss.hip), which writes the pool's feature planes itself on one rank.
#pragma once
#include "lp_mfma.h"

namespace clica {
namespace lp2 {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// v = hi + mid + lo exactly (8 + 8 + 8 significand bits, truncation); returns the pieces' upper halves in the LOW 16 bits
__device__ __forceinline__ void split3(float v, unsigned& hi, unsigned& mid, unsigned& lo) {
  const unsigned b = __float_as_uint(v);
  const unsigned hb = b & 0xffff0000u;
  const float r1 = v - __uint_as_float(hb);
  const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(mb);
  hi = hb >> 16; mid = mb >> 16; lo = __float_as_uint(r2) >> 16;
}
__device__ __forceinline__ u32x4_t pack8(const unsigned (&v)[8]) {
  return (u32x4_t){v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)};
}

// One 16-byte vector of each of the three feature planes of pool tile `tile`: (t, h, slot) = (K = 16 half, k half, feature slot < 32);
// slot f < 16: y_f = (p'_0 .. p'_{n-1}, 1, 0 ..), slot 16 + f: u_j y_f;  element e = pool row 16 t + 8 (e / 4) + 4 h + e % 4 of the tile.
// row(j, f) -> coordinate f of pool row j (j < rows), u(j) -> C_j 2^-L_j.
template <class RowFn, class UFn>
__device__ __forceinline__ void feat_vectors(int64_t tile, int t, int h, int slot, int64_t rows, int n, const float* __restrict__ origin, float pre2,
                                             RowFn row, UFn u, u32x4_t* __restrict__ FP) {
  const int f = slot & 15;
  unsigned hb[8], mb[8], lb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int64_t j = tile * ROWS + 16 * t + 8 * (e >> 2) + 4 * h + (e & 3);
    const bool live = j < rows, ok = live && f < n;
    const float x = pre2 * ((ok ? row(j, f) : 0.f) - origin[f < n ? f : 0]);
    float y = ok ? x : ((live && f == n) ? 1.f : 0.f);
    if (slot >= 16) y *= live ? u(j) : 0.f;
    split3(y, hb[e], mb[e], lb[e]);
  }
  FP[(((tile * 3 + 0) * 2 + t) * 2 + h) * 32 + slot] = pack8(hb);
  FP[(((tile * 3 + 1) * 2 + t) * 2 + h) * 32 + slot] = pack8(mb);
  FP[(((tile * 3 + 2) * 2 + t) * 2 + h) * 32 + slot] = pack8(lb);
}

}  // namespace lp2
}  // namespace clica
