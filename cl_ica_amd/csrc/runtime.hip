// Error plumbing + tiny utility kernels of the C ABI (include/clica.h).
#include "common.h"
#include <string.h>

namespace clica {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return CLICA_E_HIP;
  }
  return CLICA_OK;
}

__global__ void tick_k(int32_t* c) { *c += 1; }
// slot[0] += 1; slot[1 + 2 * (old count % cap) + which] = wall clock (100 MHz constant-rate counter, s_memrealtime).  See clica_stamp.
__global__ void stamp_k(unsigned long long* slot, int which, int cap) {
  const unsigned long long t = wall_clock64();
  const unsigned long long c = which ? slot[0] - 1 : slot[0];
  slot[1 + 2 * (c % (unsigned long long)cap) + which] = t;
  if (!which) slot[0] = c + 1;
}
// values[0..n) -> host-visible memory, then a sequence number behind a system-scope release.  See clica_publish_host.
__global__ void publish_host_k(const float* __restrict__ src, int n, float* host_dst, unsigned int* seq_dev, unsigned int* host_seq) {
  if ((int)threadIdx.x < n) host_dst[threadIdx.x] = src[threadIdx.x];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int s = *seq_dev + 1u;
    *seq_dev = s;
    __hip_atomic_store(host_seq, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// Shader-clock probe: ONE wave that samples (wall clock, core-clock counter) every `period` wall ticks, n times, and sleeps in between.
// Launched on a side stream it shares a CU with whatever the main stream runs (16 registers, no LDS: it fits next to the 8-wave
// workgroups of the encoder kernels), so the cycles it counts between two samples are that XCD's clock over that stretch of time.
// (The core-clock counters are per CU and stop with it: stamps taken by different launches on different CUs cannot be subtracted --
// tools/proto/xcc_probe.hip shows two workgroups of one launch on the same XCD 19 M cycles apart.)
__global__ void clock_probe_k(unsigned long long* samples, int n, int period) {
  unsigned long long next = wall_clock64();
  for (int i = 0; i < n; ++i) {
    unsigned long long t;
    do { __builtin_amdgcn_s_sleep(8); t = wall_clock64(); } while (t < next);
    samples[2 * i] = t;
    samples[2 * i + 1] = __builtin_amdgcn_s_memtime();
    next = t + (unsigned long long)period;
  }
}
}  // namespace clica

extern "C" int clica_clock_probe(unsigned long long* samples, int32_t n, int32_t period_ticks, clica_stream_t stream) {
  CLICA_CHECK_ARG(samples != nullptr && n >= 1 && period_ticks >= 1, "clica_clock_probe: bad argument");
  hipLaunchKernelGGL(clica::clock_probe_k, dim3(1), dim3(1), 0, clica::as_stream(stream), samples, (int)n, (int)period_ticks);
  return clica::launch_status("clica_clock_probe");
}

extern "C" const char* clica_last_error(void) { return clica::g_err; }
extern "C" int clica_version(void) { return 100; }

extern "C" int clica_stamp(unsigned long long* slot, int32_t which, int32_t capacity, clica_stream_t stream) {
  CLICA_CHECK_ARG(slot != nullptr && capacity >= 1 && (which == 0 || which == 1), "clica_stamp: bad argument");
  hipLaunchKernelGGL(clica::stamp_k, dim3(1), dim3(1), 0, clica::as_stream(stream), slot, (int)which, (int)capacity);
  return clica::launch_status("clica_stamp");
}

extern "C" int clica_publish_host(const float* src, int32_t n, float* host_dst, uint32_t* seq_dev, uint32_t* host_seq, clica_stream_t stream) {
  CLICA_CHECK_ARG(src != nullptr && host_dst != nullptr && seq_dev != nullptr && host_seq != nullptr && n >= 1 && n <= 64,
                  "clica_publish_host: bad argument (1 <= n <= 64)");
  hipLaunchKernelGGL(clica::publish_host_k, dim3(1), dim3(64), 0, clica::as_stream(stream), src, (int)n, host_dst, (unsigned int*)seq_dev,
                     (unsigned int*)host_seq);
  return clica::launch_status("clica_publish_host");
}

extern "C" int clica_tick(int32_t* counter, clica_stream_t stream) {
  CLICA_CHECK_ARG(counter != nullptr, "clica_tick: counter is NULL");
  hipLaunchKernelGGL(clica::tick_k, dim3(1), dim3(1), 0, clica::as_stream(stream), counter);
  return clica::launch_status("clica_tick");
}

// After a failed stream capture (e.g. a collective backend that cannot be captured) the stream may still be in capture
// mode and the runtime's per-thread "last error" holds the capture error, which the next launch_status() would report for
// an innocent kernel.  Ends the capture if one is active on `stream`, waits for the device and clears the error state.
extern "C" int clica_abort_capture(clica_stream_t stream) {
  hipStream_t s = clica::as_stream(stream);
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture(s, &g);       // returns the invalidation error; the stream leaves capture mode
    if (g) (void)hipGraphDestroy(g);
  }
  (void)hipDeviceSynchronize();
  for (int i = 0; i < 8 && hipGetLastError() != hipSuccess; ++i) {}
  return CLICA_OK;
}
