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

// log det W and W^-T of one small mixing matrix (c <= 8) by LU with partial pivoting, one thread: (called by ONE wave of 64 lanes; uses 528 bytes of LDS.)  What torch.logdet and its
// backward do through rocSOLVER in ~22 tiny launches per flow and direction (glow.py:100: log_det_W = B * L * logdet(W)).
// det <= 0 follows torch.logdet: NaN for a negative determinant, -inf for a singular matrix.
__device__ inline void logdet_wave(const float* __restrict__ W, int c, float* __restrict__ logdet, float* __restrict__ winv_t) {
  // one wave; lane (i, j) owns element [i][j] of the matrix and of the accumulating inverse (Gauss-Jordan on [A | I] with
  // partial pivoting).  (A one-thread version with the two 8 x 8 arrays in scratch took 72 us per call.)
  __shared__ float A[8][8], Iv[8][8];
  __shared__ int s_piv, s_singular;
  __shared__ float s_sign, s_log;
  const int tid = threadIdx.x, i = tid >> 3, j = tid & 7;
  const bool in = i < c && j < c;
  A[i][j] = in ? W[i * c + j] : (i == j ? 1.0f : 0.0f);
  Iv[i][j] = i == j ? 1.0f : 0.0f;
  if (tid == 0) { s_sign = 1.0f; s_log = 0.0f; s_singular = 0; }
  __syncthreads();
  for (int k = 0; k < c; ++k) {
    if (tid == 0) {
      int piv = k;
      float best = fabsf(A[k][k]);
      for (int r = k + 1; r < c; ++r)
        if (fabsf(A[r][k]) > best) { best = fabsf(A[r][k]); piv = r; }
      s_piv = piv;
      if (best == 0.0f) s_singular = 1;
    }
    __syncthreads();
    if (s_singular) break;
    const int piv = s_piv;
    if (piv != k && i == k) {   // row k's lanes exchange rows k and piv
      float t = A[k][j]; A[k][j] = A[piv][j]; A[piv][j] = t;
      t = Iv[k][j]; Iv[k][j] = Iv[piv][j]; Iv[piv][j] = t;
    }
    __syncthreads();
    const float d = A[k][k];
    __syncthreads();
    if (tid == 0) {
      if (piv != k) s_sign = -s_sign;
      if (d < 0.0f) s_sign = -s_sign;
      s_log += logf(fabsf(d));
    }
    if (i == k) { const float r = 1.0f / d; A[k][j] *= r; Iv[k][j] *= r; }
    __syncthreads();
    const float f = A[i][k], akj = A[k][j], ikj = Iv[k][j];
    __syncthreads();
    if (i != k) { A[i][j] = fmaf(-f, akj, A[i][j]); Iv[i][j] = fmaf(-f, ikj, Iv[i][j]); }
    __syncthreads();
  }
  if (s_singular) {
    if (tid == 0) *logdet = -INFINITY;
    if (in) winv_t[i * c + j] = NAN;
    return;
  }
  if (tid == 0) *logdet = s_sign > 0.0f ? s_log : NAN;
  if (in) winv_t[i * c + j] = Iv[j][i];
}

}  // namespace facppg
