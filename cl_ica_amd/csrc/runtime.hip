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
}  // namespace clica

extern "C" const char* clica_last_error(void) { return clica::g_err; }
extern "C" int clica_version(void) { return 100; }

extern "C" int clica_tick(int32_t* counter, clica_stream_t stream) {
  CLICA_CHECK_ARG(counter != nullptr, "clica_tick: counter is NULL");
  hipLaunchKernelGGL(clica::tick_k, dim3(1), dim3(1), 0, clica::as_stream(stream), counter);
  return clica::launch_status("clica_tick");
}
