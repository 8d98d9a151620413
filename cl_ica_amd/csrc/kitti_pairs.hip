// KITTI-masks temporal pairs, gathered and interleaved on the device  --  the batch assembly of
// /root/reference/kitti_masks/dataset.py:90-131 (__getitem__: frame `start` and frame `end` of one pedestrian sequence, uint8 mask
// * 255 -> float32 / 255, channel dim added) and :134-142 (custom_collate: inputs = [first_0, second_0, first_1, second_1, ...],
// labels likewise).  The reference builds the batch on DataLoader workers from a pickled list of bool arrays and copies it to the
// GPU; here every frame of the data set lives in HBM as one uint8 tensor [F][H*W] (4 KB per 64 x 64 mask) and ONE launch writes the
// interleaved float32 batch [2B][1][H][W] and the interleaved latents [2B][3] from the two frame indices of each pair.
// HBM-bound: 2 B H W bytes read, 8 B H W bytes written (B = 1024 pairs of 64 x 64: 8.4 MB + 33.6 MB).
#include "common.h"

namespace clica {
namespace kitti {
constexpr int THREADS = 256;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(THREADS) void pairs_k(const uint8_t* __restrict__ frames, int64_t frame_elems, int64_t n_frames,
                                                   const int64_t* __restrict__ first, const int64_t* __restrict__ second, int64_t n_pairs,
                                                   float* __restrict__ images, const float* __restrict__ latents, int n_lat,
                                                   float* __restrict__ labels) {
  const int64_t quads = frame_elems / 4;                    // 4 pixels (one 32-bit load, one 16-byte store) per thread step
  const int64_t total = 2 * n_pairs * quads;
  for (int64_t t = (int64_t)blockIdx.x * THREADS + threadIdx.x; t < total; t += (int64_t)gridDim.x * THREADS) {
    const int64_t img = t / quads, q = t - img * quads;     // image 2 i = first of pair i, 2 i + 1 = second
    int64_t f = (img & 1) ? second[img >> 1] : first[img >> 1];
    f = f < 0 ? 0 : (f >= n_frames ? n_frames - 1 : f);     // (indices are validated on the host; never read out of bounds)
    const uint32_t px = *reinterpret_cast<const uint32_t*>(frames + f * frame_elems + 4 * q);
    // astype(uint8) * 255 ... / 255.0 (:100-101, :128-131): 0 -> 0.0f, 1 -> 1.0f (255 / 255.0 is exactly 1); other byte values v
    // (never produced by the bool masks) follow the same formula with uint8 wrap-around: ((v * 255) & 255) / 255
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t v = (px >> (8 * e)) & 255u;
      o[e] = (float)((v * 255u) & 255u) / 255.0f;
    }
    *reinterpret_cast<f32x4*>(images + img * frame_elems + 4 * q) = o;
  }
  if (labels) {
    for (int64_t t = (int64_t)blockIdx.x * THREADS + threadIdx.x; t < 2 * n_pairs * n_lat; t += (int64_t)gridDim.x * THREADS) {
      const int64_t img = t / n_lat; const int k = (int)(t - img * n_lat);
      int64_t f = (img & 1) ? second[img >> 1] : first[img >> 1];
      f = f < 0 ? 0 : (f >= n_frames ? n_frames - 1 : f);
      labels[t] = latents[f * n_lat + k];
    }
  }
}
}  // namespace kitti
}  // namespace clica

using namespace clica;

extern "C" int clica_kitti_gather_pairs(const uint8_t* frames, int64_t frame_elems, int64_t n_frames, const int64_t* first_frame,
                                        const int64_t* second_frame, int64_t n_pairs, float* images, const float* latents, int32_t n_latents,
                                        float* labels, clica_stream_t stream) {
  CLICA_CHECK_ARG(frames && first_frame && second_frame && images && n_pairs > 0 && n_frames > 0, "clica_kitti_gather_pairs: NULL pointer / empty batch");
  CLICA_CHECK_ARG(frame_elems > 0 && frame_elems % 4 == 0, "clica_kitti_gather_pairs: %lld pixels per frame (a multiple of 4 required)", (long long)frame_elems);
  CLICA_CHECK_ARG((reinterpret_cast<uintptr_t>(frames) & 3) == 0 && (reinterpret_cast<uintptr_t>(images) & 15) == 0,
                  "clica_kitti_gather_pairs: frames must be 4-byte, images 16-byte aligned");
  CLICA_CHECK_ARG(!labels || (latents && n_latents >= 1), "clica_kitti_gather_pairs: labels requested without a latent table");
  const int64_t work = 2 * n_pairs * (frame_elems / 4);
  const int64_t blocks = ceil_div(work, kitti::THREADS);
  hipLaunchKernelGGL(kitti::pairs_k, dim3((unsigned)(blocks < 8 * kNumCU ? blocks : 8 * kNumCU)), dim3(kitti::THREADS), 0, as_stream(stream),
                     frames, frame_elems, n_frames, first_frame, second_frame, n_pairs, images, latents, (int)n_latents, labels);
  return launch_status("clica_kitti_gather_pairs");
}
