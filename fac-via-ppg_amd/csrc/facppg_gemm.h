// Internal (not part of the C ABI): generic exact-fp32 MFMA "tapped GEMM" used by the Tacotron
// and STFT entry points.
//
//   C[b][m][n] = epilogue( sum_{tap, c} W[m][c][tap] * X[b][c][n + (tap - pad) * dil] )
//
// i.e. a 1-D convolution over channel-major activations X[b][Cin][ldx] (positions contiguous),
// which for taps = 1 is a plain matrix product (Linear layers applied to [C][T] activations, DFT
// bases, the mel filterbank).  Columns outside [0, n_valid[b]) read as zero (the convolution's
// zero padding; also what makes padded batches equal to independent batch-1 runs).
//
// W is pre-packed once into the MFMA A-operand image (pack_a): float4 index
// (mb * KG + g) * 64 + lane holds row mb*32 + (lane & 31), K entries 8g + 4(lane>>5) + {0..3},
// K ordered tap-major (k = tap * Cin + c), zero padded to a multiple of 64.
#pragma once
#include "facppg_common.h"

namespace facppg {

enum GemmAct { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2, ACT_LOG_CLAMP = 3 /* log(max(v, 1e-5)) */ };

inline int gemm_kpad(int K) { return round_up(K, 64); }
// + 3 k-groups: k_gemm prefetches up to 3 groups past the last row block (never used)
inline size_t packed_a_float4s(int M, int K) { return (size_t)(round_up(M, 32) / 32) * (gemm_kpad(K) / 8 + 1) * 64 + 3 * 64; }

// src element (m, c, tap) at src[(m * Cin + c) * taps + tap]  (torch Conv1d / Linear weight layout)
int pack_a(const float* src, int M, int Cin, int taps, float4* dst, hipStream_t s);
// general form: element (m, c, tap) at src[off + m*sm + c*sc + tap*st] (transposed / tap-reversed views)
int pack_a_strided(const float* src, int M, int Cin, int taps, long sm, long sc, long st, long off, float4* dst, hipStream_t s);

struct GemmArgs {
  const float4* A = nullptr;  // packed weights
  int M = 0;                  // output rows
  int Cin = 0, taps = 1, dil = 1, pad = 0;
  const float* X = nullptr;   // [B][Cin][ldx]
  long x_bs = 0;
  int ldx = 0;
  int N = 0;                  // columns (max over the batch)
  // column window (streaming convolutions: a layer is extended by the columns whose inputs have become final): output
  // columns [col0, N) only; source columns are valid in [0, src_hi) (src_hi = 0: up to the output bound, i.e. the
  // convolution's zero padding starts where the outputs end); *skip != 0 (device): the launch does nothing
  int col0 = 0;
  int src_hi = 0;
  const int* skip = nullptr;
  const int* n_valid = nullptr;  // optional per-batch column count (device)
  int n_valid_mul = 1;           // valid columns = n_valid[b] * n_valid_mul + n_valid_add
  int n_valid_add = 0;
  const float* bias = nullptr;   // per row, optional
  const float* scale = nullptr;  // per row, optional: v = v * scale + shift (eval BatchNorm)
  const float* shift = nullptr;
  int act = ACT_NONE;
  const uint8_t* mask = nullptr;  // optional keep-mask [B][M][ldmask]: v *= 2*mask (dropout p=0.5)
  long mask_bs = 0;
  int ldmask = 0;
  const float* res = nullptr;     // optional residual add [B][M][ldres]
  long res_bs = 0;
  int ldres = 0;
  float* C = nullptr;             // [B][M][ldc]
  long c_bs = 0;
  int ldc = 0;
  int c_transposed = 0;           // 1: store C[b][n][m] (address n*ldc + m)
  // gate backward (training): v = d(acts)[m][n]; with T = tanh(pre_t), S = sigmoid(pre_s) saved by the
  // forward at gate_ts[b][m][n] / gate_ts[b][M+m][n]: C[m] = v*S*(1-T^2), C[M+m] = v*T*S*(1-S)
  const float* gate_ts = nullptr;
  long gate_bs = 0;
  int ldgate = 0;
  int B = 1;
  // optional scratch for split-K (small-N products): [splits][B][M][N] partial sums; gemm_launch
  // decides whether and how far to split
  float* splitk_ws = nullptr;
  size_t splitk_ws_bytes = 0;
};

int gemm_launch(const GemmArgs& a, hipStream_t s);

}  // namespace facppg
