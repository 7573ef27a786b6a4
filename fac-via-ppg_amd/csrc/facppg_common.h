// Internal helpers shared by the translation units of libfacppg_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "facppg.h"

namespace facppg {

void set_error(const char* fmt, ...);

#define FACPPG_HIP_CHECK(expr)                                                          \
  do {                                                                                  \
    hipError_t e__ = (expr);                                                            \
    if (e__ != hipSuccess) {                                                            \
      ::facppg::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return FACPPG_EHIP;                                                               \
    }                                                                                   \
  } while (0)

#define FACPPG_REQUIRE(cond, code, ...)  \
  do {                                   \
    if (!(cond)) {                       \
      ::facppg::set_error(__VA_ARGS__);  \
      return (code);                     \
    }                                    \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;  // CDNA wavefront

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// One exact-fp32 MFMA: D[32x32] += A[32x2] * B[2x32]; lane l holds A[l&31][l>>5], B[l>>5][l&31].
__device__ __forceinline__ f32x16 mfma32x32x2(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] * B[4x16].  Lane l supplies A[l % 16][l / 16] and
// B[l / 16][l % 16]; it holds D[4 * (l / 16) + r][l % 16] in c[r].
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16x16x4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

}  // namespace facppg
