// Shared helpers for libclica_hip.so (gfx950 only; no portability layer on purpose).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include "clica.h"

namespace clica {

void set_error(const char* fmt, ...);
int launch_status(const char* what);

static inline hipStream_t as_stream(clica_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr int kNumCU = 256;  // MI355X

}  // namespace clica

#define CLICA_CHECK_ARG(cond, ...)            \
  do {                                        \
    if (!(cond)) {                            \
      ::clica::set_error(__VA_ARGS__);        \
      return CLICA_E_INVALID;                 \
    }                                         \
  } while (0)
