// Generic exact-fp32 MFMA tapped GEMM (see facppg_gemm.h).  One workgroup = 4 waves computes a
// 128 (rows) x 64 (columns) tile: wave w owns rows 32w..32w+31 (one MFMA row block) and both
// 32-column blocks.  The activation operand is staged through LDS in [64 k-rows][64 columns]
// chunks (double buffered, register-staged so the global loads of chunk c+1 fly under the MFMAs
// of chunk c); the weight operand streams straight from its packed image into registers.
#include "facppg_gemm.h"

namespace facppg {
namespace {

constexpr int TN = 64, KCH = 64;

__global__ void k_pack_a(const float* __restrict__ src, float4* __restrict__ dst, int M, int Cin, int taps, int KG, long sm,
                         long sc, long st, long off) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int MB = (M + 31) / 32;
  if (idx >= MB * (KG + 1) * 64) return;
  const int lane = idx & 63, g = (idx >> 6) % (KG + 1), mb = (idx >> 6) / (KG + 1);
  const int m = mb * 32 + (lane & 31), K = Cin * taps;
  float v[4];
  for (int s = 0; s < 4; ++s) {
    const int k = 8 * g + 4 * (lane >> 5) + s;
    const int tap = k / Cin, c = k - tap * Cin;
    v[s] = (m < M && k < K && g < KG) ? src[off + m * sm + c * sc + tap * st] : 0.0f;
  }
  dst[idx] = make_float4(v[0], v[1], v[2], v[3]);
}

struct KArgs {
  GemmArgs g;
  int KG;  // k-groups of 8 (Kpad / 8)
  int sk;  // K splits (1 = none): blockIdx.z = b * sk + ks, partial sums go to g.splitk_ws
};

// bias, eval-BN scale/shift, activation, dropout mask, residual, gate backward, store
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, int b, int m, int n, float v) {
  if (p.bias) v += p.bias[m];
  if (p.scale) v = v * p.scale[m] + p.shift[m];
  if (p.act == ACT_RELU) v = fmaxf(v, 0.0f);
  else if (p.act == ACT_TANH) v = tanhf(v);
  else if (p.act == ACT_LOG_CLAMP) v = logf(fmaxf(v, 1e-5f));
  if (p.mask) v = v * (float)p.mask[(size_t)b * p.mask_bs + (size_t)m * p.ldmask + n] * 2.0f;
  if (p.res) v += p.res[(size_t)b * p.res_bs + (size_t)m * p.ldres + n];
  if (p.gate_ts) {
    const float T = p.gate_ts[(size_t)b * p.gate_bs + (size_t)m * p.ldgate + n];
    const float S = p.gate_ts[(size_t)b * p.gate_bs + (size_t)(p.M + m) * p.ldgate + n];
    p.C[(size_t)b * p.c_bs + (size_t)m * p.ldc + n] = v * S * (1.0f - T * T);
    p.C[(size_t)b * p.c_bs + (size_t)(p.M + m) * p.ldc + n] = v * T * S * (1.0f - S);
  } else if (p.c_transposed) p.C[(size_t)b * p.c_bs + (size_t)n * p.ldc + m] = v;
  else p.C[(size_t)b * p.c_bs + (size_t)m * p.ldc + n] = v;
}

__global__ __launch_bounds__(256) void k_gemm(KArgs ka) {
  const GemmArgs& p = ka.g;
  __shared__ __attribute__((aligned(16))) float smem[2 * KCH * TN];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, kh = lane >> 5;
  const int b = blockIdx.z / ka.sk, ks = blockIdx.z % ka.sk, n0 = p.col0 + blockIdx.x * TN;
  const int Nb = p.n_valid ? min(p.N, p.n_valid[b] * p.n_valid_mul + p.n_valid_add) : p.N;
  if (n0 >= Nb) return;
  if (p.skip && *p.skip) return;
  const int Ns = p.src_hi > 0 ? (p.n_valid ? min(p.src_hi, p.n_valid[b] * p.n_valid_mul + p.n_valid_add) : p.src_hi) : Nb;
  const int mb = blockIdx.y * 4 + w;
  const int MB = (p.M + 31) / 32;
  const bool active = mb < MB;
  const int K = p.Cin * p.taps;
  const int nch_all = ka.KG / 8;
  const int c_lo = ks * nch_all / ka.sk, c_hi = (ks + 1) * nch_all / ka.sk;   // this split's chunks

  f32x16 acc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.0f; acc[1][r] = 0.0f; }

  const float* xb = p.X + (size_t)b * p.x_bs;
  const int col = n0 + lane;
  float stg[16];
  auto stage_load = [&](int c) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int kk = c * KCH + w * 16 + j;
      const int tap = kk / p.Cin, ch = kk - tap * p.Cin;
      const int sc = col + (tap - p.pad) * p.dil;
      stg[j] = (kk < K && sc >= 0 && sc < Ns) ? xb[(size_t)ch * p.ldx + sc] : 0.0f;
    }
  };
  auto stage_write = [&](int buf) {
    float* dst = smem + buf * (KCH * TN) + (w * 16) * TN + lane;
#pragma unroll
    for (int j = 0; j < 16; ++j) dst[j * TN] = stg[j];
  };

  // Weights: a ring of 4 register sets, fetched 3 k-groups (48 MFMAs) ahead.  Vector-memory loads
  // retire in order, so the wait on a weight load also waits for the older activation staging loads:
  // the staging load is therefore unconditional (no branch whose shorter path would drag the wait to
  // the first MFMA group; the last chunk re-stages itself) and sits 3 groups ahead of the first
  // younger weight load that is waited on.  The packed image carries one zero k-group of padding per
  // row block and row blocks are contiguous, so reading up to 3 groups past a row block's end stays
  // inside the image except for the last row block, which pack_a pads (packed_a_float4s).
  const float4* ap = p.A + (size_t)(active ? mb : 0) * (ka.KG + 1) * 64 + lane;
  constexpr int RING = 4;
  float4 ar[RING];
  stage_load(c_lo);
#pragma unroll
  for (int i = 0; i < RING - 1; ++i) ar[i] = ap[(size_t)(c_lo * 8 + i) * 64];
  stage_write(c_lo & 1);
  __syncthreads();
  for (int c = c_lo; c < c_hi; ++c) {
    stage_load(c + 1 < c_hi ? c + 1 : c);
    const float* lb = smem + (c & 1) * (KCH * TN) + (4 * kh) * TN + li;
    const int G = c * 8;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      ar[(g + RING - 1) % RING] = ap[(size_t)(G + g + RING - 1) * 64];
      __builtin_amdgcn_sched_barrier(0);
      const float4 a0 = ar[g % RING];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float av = s == 0 ? a0.x : s == 1 ? a0.y : s == 2 ? a0.z : a0.w;
        acc[0] = mfma32x32x2(av, lb[(8 * g + s) * TN], acc[0]);
        acc[1] = mfma32x32x2(av, lb[(8 * g + s) * TN + 32], acc[1]);
      }
    }
    stage_write((c + 1) & 1);
    __syncthreads();
  }
  if (!active) return;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int n = n0 + cb * 32 + li;
    if (n >= Nb) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mb * 32 + 8 * (r >> 2) + (r & 3) + 4 * kh;
      if (m >= p.M) continue;
      if (ka.sk > 1) p.splitk_ws[(((size_t)ks * p.B + b) * p.M + m) * p.N + n] = acc[cb][r];
      else gemm_epilogue(p, b, m, n, acc[cb][r]);
    }
  }
}

// second pass of a split-K product: fixed-order sum of the partials, then the epilogue
__global__ void k_gemm_reduce(KArgs ka) {
  const GemmArgs& p = ka.g;
  const int n = p.col0 + blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y, b = blockIdx.z;
  const int Nb = p.n_valid ? min(p.N, p.n_valid[b] * p.n_valid_mul + p.n_valid_add) : p.N;
  if (n >= Nb) return;
  if (p.skip && *p.skip) return;
  float v = 0.0f;
  for (int ks = 0; ks < ka.sk; ++ks) v += p.splitk_ws[(((size_t)ks * p.B + b) * p.M + m) * p.N + n];
  gemm_epilogue(p, b, m, n, v);
}

}  // namespace

int pack_a_strided(const float* src, int M, int Cin, int taps, long sm, long sc, long st, long off, float4* dst, hipStream_t s) {
  const int KG = gemm_kpad(Cin * taps) / 8;
  const int total = (round_up(M, 32) / 32) * (KG + 1) * 64;
  k_pack_a<<<(total + 255) / 256, 256, 0, s>>>(src, dst, M, Cin, taps, KG, sm, sc, st, off);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

int pack_a(const float* src, int M, int Cin, int taps, float4* dst, hipStream_t s) {
  return pack_a_strided(src, M, Cin, taps, (long)Cin * taps, taps, 1, 0, dst, s);
}

int gemm_launch(const GemmArgs& a, hipStream_t s) {
  FACPPG_REQUIRE(a.A && a.X && a.C && a.M > 0 && a.N > 0 && a.Cin > 0 && a.taps > 0 && a.B > 0 && a.col0 >= 0 && a.col0 < a.N, FACPPG_EINVAL,
                 "gemm_launch: bad arguments");
  KArgs ka;
  ka.g = a;
  ka.KG = gemm_kpad(a.Cin * a.taps) / 8;
  ka.sk = 1;
  dim3 grid((a.N - a.col0 + TN - 1) / TN, (round_up(a.M, 32) / 32 + 3) / 4, a.B);
  // Long reductions are split over K when the caller lent a partial-sum buffer: a product with few columns
  // (one short utterance through the encoder / postnet) otherwise launches a few dozen workgroups with a
  // long serial K loop each.  The split factor depends on K ALONE -- not on the batch or the padded
  // length -- so an utterance is summed in the same order whatever batch it rides in (padded batches equal
  // independent runs bit for bit); large products pay a little extra partial-sum traffic for that.
  const int nch = ka.KG / 8;
  if (a.splitk_ws && nch >= 8) {
    const int sk = nch / 4 < 16 ? nch / 4 : 16;
    const size_t need = (size_t)sk * a.B * a.M * a.N * sizeof(float);
    FACPPG_REQUIRE(need <= a.splitk_ws_bytes, FACPPG_EWORKSPACE, "gemm_launch: split-K buffer has %zu bytes, needs %zu",
                   a.splitk_ws_bytes, need);
    ka.sk = sk;
  }
  grid.z = a.B * ka.sk;
  k_gemm<<<grid, 256, 0, s>>>(ka);
  if (ka.sk > 1) k_gemm_reduce<<<dim3((a.N - a.col0 + 255) / 256, a.M, a.B), 256, 0, s>>>(ka);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

}  // namespace facppg
