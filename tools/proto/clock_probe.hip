// Average shader clock over a stretch of work: one thread stamps {s_memtime (shader cycles), s_memrealtime (100 MHz)}.
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void stamp_k(uint64_t* out) {
  if (threadIdx.x == 0) { out[0] = clock64(); out[1] = wall_clock64(); }
}
extern "C" int clock_stamp(uint64_t* out, void* stream) {
  hipLaunchKernelGGL(stamp_k, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
  return (int)hipGetLastError();
}
