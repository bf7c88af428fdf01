// Shared helpers for the gfx950 kernels (host + device).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "gad.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void gad_set_error(const char* fmt, ...);

#define GAD_CHECK(cond, ...)        \
  do {                              \
    if (!(cond)) {                  \
      gad_set_error(__VA_ARGS__);   \
      return 1;                     \
    }                               \
  } while (0)

#define GAD_LAUNCH_CHECK(name)                                            \
  do {                                                                    \
    hipError_t e_ = hipGetLastError();                                    \
    if (e_ != hipSuccess) {                                               \
      gad_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return 2;                                                           \
    }                                                                     \
  } while (0)

static inline bool gad_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int64_t gad_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// wave64 sum reduction (all lanes get the total)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
