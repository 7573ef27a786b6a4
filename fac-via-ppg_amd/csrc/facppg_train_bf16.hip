// WaveGlow training step with bf16 MFMA operands (BASELINE config 5): one flow's WN stack (src/waveglow/glow.py:154-175)
// forward-with-save and its complete backward -- data gradients AND weight gradients -- on v_mfma_f32_32x32x16_bf16,
// fp32 accumulation, fp32 master weights and fp32 gradients.
//
// Layout.  The bf16 MFMA wants 8 CONSECUTIVE reduction indices per lane for both operands.  The data GEMMs of the
// layer (dilated conv + conditioning, res/skip conv, and their transposes in the backward) reduce over CHANNELS, so
// training activations are stored POSITION-MAJOR, channels contiguous ("channels last"):
//     h_i   [B][HALO + Lr + HALO][256] bf16   layer inputs; the zero margin rows ARE the conv's zero padding
//     ts_i  [B][Lr][512] bf16                 tanh | sigmoid halves of the gate (saved for the backward)
//     acts_i[B][Lr][256] bf16                 gated activations (operand of the res/skip conv and of its weight gradient)
//     skip  [B][Lr][256] fp32                 running skip sum
//     spect [B][Lr][640] bf16                 conditioning
//     dpre_i[B][HALO + Lr + HALO][512] bf16,  dh_i, dskip [B][Lr][256] bf16
// so a lane's 8 reduction values are one 16-byte load, a dilated tap is a ROW offset (always aligned), and rows
// >= L stay zero (never written), which is the reference's zero padding at the end of the segment.
// The weight gradients reduce over POSITIONS instead; k_wgrad transposes 8x8 blocks in registers while staging
// (16-byte loads along channels, v_perm, 16-byte LDS stores along positions).
//
// Kernels: k_pack_bf16 (fp32 weights -> bf16 A-operand images, once per step), k_bgemm (128 x 128 tiles, A operand
// streamed from its packed image into registers, B operand staged through LDS [column][k]; epilogues: gate,
// res/skip, gate backward, transposed-conv accumulate, plain), k_wgrad (NT products batched over layers and taps),
// k_colsum (bias gradients) and the <= 8-channel start / end convs and their backward as streaming kernels.
#include <algorithm>
#include <cstring>

#include "facppg_common.h"

namespace facppg {
namespace {

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int C = 256, NCOND = 640, HALO = 128;
constexpr int BM = 128, BN = 128, KC = 64;     // k_bgemm tile and K chunk
constexpr int LDB = KC + 8;                    // LDS row pitch in bf16 (144 B: conflict-free ds_read_b128)
constexpr int MAXSEG = 8;
// padded positions per batch item: a multiple of the 128-column GEMM tile, so a tile's staging loads never leave the
// batch item's rows (rows >= L are zero)
__host__ __device__ inline int pad_len(int L) { return round_up(L, BN); }

__device__ __forceinline__ bf16_t f2bf(float f) {   // round to nearest even
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned pack2(float lo, float hi) { return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16); }
__device__ __forceinline__ float lo2f(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float hi2f(unsigned v) { return __uint_as_float(v & 0xffff0000u); }

__device__ __forceinline__ f32x16 mfma_bf16(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------------------
// A-operand image: uint4 index (mb * KG + g) * 64 + lane holds packed row 32*mb + (lane & 31) and the 8 reduction
// entries 16*g + 8*(lane >> 5) + {0..7}.  Element (row m, entry k = k_base + tap*Cin + c) comes from
// src[off + m*sm + c*sc + tap*st].  GATE_ROWS: packed row rho of block mb is tanh/sigmoid row of channel
// 16*mb + ((rho>>3)&1)*8 + (rho&7), sigmoid for rho >= 16 -- so an MFMA lane holds both halves of its channels.
// ------------------------------------------------------------------------------------------------------------
struct PackArgs {
  const float* src;
  uint4* dst;
  int M, KG, k_base, Cin, taps, gate_rows;
  long sm, sc, st, off;
};
__device__ __forceinline__ int gate_row_src(int mb, int rho) { return (rho >= 16 ? C : 0) + 16 * mb + ((rho >> 3) & 1) * 8 + (rho & 7); }

__global__ void k_pack_bf16(PackArgs p) {
  const int ng = p.Cin * p.taps / 16;                       // k16 groups written by this call
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int MB = (p.M + 31) / 32;
  if (idx >= MB * ng * 64) return;
  const int lane = idx & 63, gl = (idx >> 6) % ng, mb = (idx >> 6) / ng;
  const int rho = lane & 31;
  const int m = p.gate_rows ? gate_row_src(mb, rho) : mb * 32 + rho;
  unsigned w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = 16 * gl + 8 * (lane >> 5) + 2 * j + e;
      const int tap = k / p.Cin, c = k - tap * p.Cin;
      v[e] = m < p.M ? p.src[p.off + m * p.sm + c * p.sc + tap * p.st] : 0.0f;
    }
    w[j] = pack2(v[0], v[1]);
  }
  p.dst[((size_t)mb * p.KG + p.k_base / 16 + gl) * 64 + lane] = make_uint4(w[0], w[1], w[2], w[3]);
}

// ------------------------------------------------------------------------------------------------------------
// k_bgemm: out[b][n][m] = epilogue( sum_k A[m][k] * X[b][n + shift(k)][c(k)] ), the reduction being a list of
// segments (each a multiple of 64 channels of one position-major tensor at one row shift).
// ------------------------------------------------------------------------------------------------------------
struct Seg {
  const bf16_t* x;   // [B][rows][ld]
  long bs;           // batch stride (elements)
  int ld, row0;      // row of position 0 (+ the tap's shift)
  int nch;           // channels in this segment (multiple of 64)
};
enum { EP_GATE = 0, EP_RESSKIP = 1, EP_BWD_GATE = 2, EP_BWD_CONV = 3, EP_ACC_F32 = 4 };
struct BGemmArgs {
  const uint4* A;
  int KG;            // k16 groups per row block in the image
  int M, N, B;       // rows (multiple of 32), valid positions per batch item, batch
  int nseg;
  Seg seg[MAXSEG];
  int mode;
  const float* bias;        // GATE: b1[512] (in + cond, source row order); RESSKIP: b2
  // GATE
  bf16_t* acts; bf16_t* ts; int Lr;
  // RESSKIP (res rows < C unless `last`): h_out = h_in + res, skip (+)= skip part
  const bf16_t* h_in; bf16_t* h_out; long h_bs; int h_row0; float* skip; int first, last;
  // BWD_GATE: ts (above) + dpre [B][HALO + Lr + HALO][512]
  bf16_t* dpre; long dpre_bs;
  // BWD_CONV: dh_out = dh_next (may be null) + v
  const bf16_t* dh_next; bf16_t* dh_out;
  // ACC_F32: out [B][Lr][ldo] (+)= v
  float* outf; int ldo, accumulate;
};

template <int MODE>
__device__ __forceinline__ void bgemm_store4(const BGemmArgs& p, int b, int n, int mb, int q, int kh, const float (&v)[4],
                                             const float (&v2)[4]) {
  // v: accumulator rows 8q + 4kh + {0..3} of block mb at position n  (GATE: v = tanh rows (q < 2), v2 = matching sigmoid rows)
  if constexpr (MODE == EP_GATE) {
    const int ch = 16 * mb + q * 8 + 4 * kh;
    float T[4], S[4], a[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float x = v[t] + p.bias[ch + t], y = v2[t] + p.bias[C + ch + t];
      const float ea = __expf(-2.0f * fminf(fmaxf(x, -15.0f), 15.0f));
      T[t] = __fdividef(1.0f - ea, 1.0f + ea);
      S[t] = __fdividef(1.0f, 1.0f + __expf(-y));
      a[t] = T[t] * S[t];
    }
    const size_t row = (size_t)b * p.Lr + n;
    *reinterpret_cast<uint2*>(p.acts + row * C + ch) = make_uint2(pack2(a[0], a[1]), pack2(a[2], a[3]));
    *reinterpret_cast<uint2*>(p.ts + row * 2 * C + ch) = make_uint2(pack2(T[0], T[1]), pack2(T[2], T[3]));
    *reinterpret_cast<uint2*>(p.ts + row * 2 * C + C + ch) = make_uint2(pack2(S[0], S[1]), pack2(S[2], S[3]));
  } else if constexpr (MODE == EP_RESSKIP) {
    const int m = 32 * mb + 8 * q + 4 * kh;
    if (!p.last && m < C) {
      const size_t o = (size_t)b * p.h_bs + (size_t)(p.h_row0 + n) * C + m;
      const uint2 hin = *reinterpret_cast<const uint2*>(p.h_in + o);
      const float r0 = v[0] + p.bias[m] + lo2f(hin.x), r1 = v[1] + p.bias[m + 1] + hi2f(hin.x);
      const float r2 = v[2] + p.bias[m + 2] + lo2f(hin.y), r3 = v[3] + p.bias[m + 3] + hi2f(hin.y);
      *reinterpret_cast<uint2*>(p.h_out + o) = make_uint2(pack2(r0, r1), pack2(r2, r3));
    } else {
      const int cs = p.last ? m : m - C;
      float4* dst = reinterpret_cast<float4*>(p.skip + ((size_t)b * p.Lr + n) * C + cs);
      float4 s = p.first ? make_float4(0.f, 0.f, 0.f, 0.f) : *dst;
      s.x += v[0] + p.bias[m]; s.y += v[1] + p.bias[m + 1]; s.z += v[2] + p.bias[m + 2]; s.w += v[3] + p.bias[m + 3];
      *dst = s;
    }
  } else if constexpr (MODE == EP_BWD_GATE) {
    const int ch = 32 * mb + 8 * q + 4 * kh;
    const size_t row = (size_t)b * p.Lr + n;
    const uint2 Tp = *reinterpret_cast<const uint2*>(p.ts + row * 2 * C + ch);
    const uint2 Sp = *reinterpret_cast<const uint2*>(p.ts + row * 2 * C + C + ch);
    const float T[4] = {lo2f(Tp.x), hi2f(Tp.x), lo2f(Tp.y), hi2f(Tp.y)}, S[4] = {lo2f(Sp.x), hi2f(Sp.x), lo2f(Sp.y), hi2f(Sp.y)};
    float dt[4], ds[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { dt[t] = v[t] * S[t] * (1.0f - T[t] * T[t]); ds[t] = v[t] * T[t] * S[t] * (1.0f - S[t]); }
    bf16_t* d = p.dpre + (size_t)b * p.dpre_bs + (size_t)(HALO + n) * 2 * C + ch;
    *reinterpret_cast<uint2*>(d) = make_uint2(pack2(dt[0], dt[1]), pack2(dt[2], dt[3]));
    *reinterpret_cast<uint2*>(d + C) = make_uint2(pack2(ds[0], ds[1]), pack2(ds[2], ds[3]));
  } else if constexpr (MODE == EP_BWD_CONV) {
    const int ch = 32 * mb + 8 * q + 4 * kh;
    const size_t o = ((size_t)b * p.Lr + n) * C + ch;
    float r[4] = {v[0], v[1], v[2], v[3]};
    if (p.dh_next) {
      const uint2 d = *reinterpret_cast<const uint2*>(p.dh_next + o);
      r[0] += lo2f(d.x); r[1] += hi2f(d.x); r[2] += lo2f(d.y); r[3] += hi2f(d.y);
    }
    *reinterpret_cast<uint2*>(p.dh_out + o) = make_uint2(pack2(r[0], r[1]), pack2(r[2], r[3]));
  } else {
    const int m = 32 * mb + 8 * q + 4 * kh;
    float4* dst = reinterpret_cast<float4*>(p.outf + ((size_t)b * p.Lr + n) * p.ldo + m);
    float4 s = p.accumulate ? *dst : make_float4(0.f, 0.f, 0.f, 0.f);
    s.x += v[0]; s.y += v[1]; s.z += v[2]; s.w += v[3];
    *dst = s;
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_bgemm(BGemmArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[2][BN * LDB];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, kh = lane >> 5;
  const int b = blockIdx.z, n0 = blockIdx.x * BN;
  const int mb = blockIdx.y * 4 + w;
  const bool active = mb * 32 < p.M;
  const uint4* ap = p.A + (size_t)(active ? mb : 0) * p.KG * 64 + lane;
  // staging: thread -> rows srow + 32*j (j < 4) of the [128 positions][64 k] chunk, 16 bytes at k = 8*sk
  const int srow = tid >> 3, sk = tid & 7;
  int nchunks = 0;
  for (int s = 0; s < p.nseg; ++s) nchunks += p.seg[s].nch / KC;

  f32x16 acc[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;

  // chunk c -> (segment, channel offset): walked incrementally
  int seg_i = 0, seg_c = 0;
  auto chunk_src = [&](int rowj) -> const uint4* {
    const Seg& sg = p.seg[seg_i];
    return reinterpret_cast<const uint4*>(sg.x + (size_t)b * sg.bs + (size_t)(sg.row0 + n0 + rowj) * sg.ld + seg_c + 8 * sk);
  };
  auto advance = [&]() {
    seg_c += KC;
    if (seg_c >= p.seg[seg_i].nch && seg_i + 1 < p.nseg) { ++seg_i; seg_c = 0; }
  };
  uint4 stg[4], a_cur[4], a_nxt[4];
  auto stage_load = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) stg[j] = *chunk_src(srow + 32 * j);
  };
  auto stage_write = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4*>(&lds[buf][(srow + 32 * j) * LDB + 8 * sk]) = stg[j];
  };
  stage_load();
#pragma unroll
  for (int s = 0; s < 4; ++s) a_cur[s] = ap[(size_t)s * 64];
  stage_write(0);
  advance();
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    if (more) {
      stage_load();
#pragma unroll
      for (int s = 0; s < 4; ++s) a_nxt[s] = ap[(size_t)((c + 1) * 4 + s) * 64];
    }
    const bf16_t* lb = &lds[c & 1][li * LDB + 8 * kh];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const uint4 bv = *reinterpret_cast<const uint4*>(lb + cb * 32 * LDB + 16 * s);
        acc[cb] = mfma_bf16(a_cur[s], bv, acc[cb]);
      }
    }
    if (more) {
      stage_write((c + 1) & 1);
      advance();
#pragma unroll
      for (int s = 0; s < 4; ++s) a_cur[s] = a_nxt[s];
    }
    __syncthreads();
  }
  if (!active) return;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int n = n0 + cb * 32 + li;
    if (n >= p.N) continue;
    if constexpr (MODE == EP_GATE) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float v[4] = {acc[cb][4 * q], acc[cb][4 * q + 1], acc[cb][4 * q + 2], acc[cb][4 * q + 3]};
        const float v2[4] = {acc[cb][8 + 4 * q], acc[cb][8 + 4 * q + 1], acc[cb][8 + 4 * q + 2], acc[cb][8 + 4 * q + 3]};
        bgemm_store4<MODE>(p, b, n, mb, q, kh, v, v2);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float v[4] = {acc[cb][4 * q], acc[cb][4 * q + 1], acc[cb][4 * q + 2], acc[cb][4 * q + 3]};
        bgemm_store4<MODE>(p, b, n, mb, q, kh, v, v);
      }
    }
  }
}

int bgemm_launch(const BGemmArgs& a, hipStream_t s) {
  int K = 0;
  for (int i = 0; i < a.nseg; ++i) {
    FACPPG_REQUIRE(a.seg[i].nch % KC == 0 && a.seg[i].nch > 0, FACPPG_EINVAL, "bgemm: segment of %d channels", a.seg[i].nch);
    K += a.seg[i].nch;
  }
  FACPPG_REQUIRE(K == a.KG * 16 && a.M % 32 == 0 && a.nseg >= 1 && a.nseg <= MAXSEG, FACPPG_EINVAL, "bgemm: bad shape K=%d KG=%d M=%d", K, a.KG, a.M);
  const dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, a.B);
  switch (a.mode) {
    case EP_GATE: k_bgemm<EP_GATE><<<grid, 256, 0, s>>>(a); break;
    case EP_RESSKIP: k_bgemm<EP_RESSKIP><<<grid, 256, 0, s>>>(a); break;
    case EP_BWD_GATE: k_bgemm<EP_BWD_GATE><<<grid, 256, 0, s>>>(a); break;
    case EP_BWD_CONV: k_bgemm<EP_BWD_CONV><<<grid, 256, 0, s>>>(a); break;
    default: k_bgemm<EP_ACC_F32><<<grid, 256, 0, s>>>(a); break;
  }
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

// ------------------------------------------------------------------------------------------------------------
// k_wgrad: out[m][k] = sum_{b, n < L} dY[b][n][m] * X[b][n + shift][k]   (fp32 out), batched over problems
// (layers x taps) in blockIdx.z.  128 x 128 output tiles, 4 waves as 2 x 2 (64 x 64 each), 64 positions per
// chunk; both operands are position-major in memory and are transposed 8 x 8 in registers on their way to the
// LDS images [channel][position].
// ------------------------------------------------------------------------------------------------------------
struct WgradProb {
  const bf16_t* dy0;  // rows m < msplit   [B][rows][ldy]
  const bf16_t* dy1;  // rows m >= msplit  (may be null)
  long dy_bs; int ldy, dy_row0, msplit;
  const bf16_t* x; long x_bs; int ldx, x_row0;   // x_row0 includes the tap shift
  float* out; long o_sm, o_sk;                    // out[m * o_sm + k * o_sk]
  int M, K;
};
constexpr int MAXPROB = 24;
struct WgradArgs {
  WgradProb prob[MAXPROB];
  int B, L, Lr;
};
constexpr int LDP = 64 + 8;   // LDS pitch (positions) of the transposed images

// 8 x 8 transpose of 16-bit elements held as r[pos][4 dwords] -> t[ch][4 dwords]
__device__ __forceinline__ void transpose8x8(const uint4 (&r)[8], uint4 (&t)[8]) {
  const unsigned* ri = reinterpret_cast<const unsigned*>(r);
  unsigned* ti = reinterpret_cast<unsigned*>(t);
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned a = ri[(2 * q) * 4 + (c >> 1)], b2 = ri[(2 * q + 1) * 4 + (c >> 1)];
      // low half = position 2q, high half = position 2q+1, both channel c
      ti[c * 4 + q] = (c & 1) ? __builtin_amdgcn_perm(b2, a, 0x07060302u) : __builtin_amdgcn_perm(b2, a, 0x05040100u);
    }
}

__global__ __launch_bounds__(256) void k_wgrad(WgradArgs wa) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[2][128 * LDP];   // [A | B][channel][position]; the next chunk waits in registers
  const WgradProb& p = wa.prob[blockIdx.z];
  const int m0 = blockIdx.y * 128, k0 = blockIdx.x * 128;
  if (m0 >= p.M || k0 >= p.K) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, kh = lane >> 5;
  const int wm = w >> 1, wk = w & 1;
  // staging role: threads 0..127 transpose dY blocks, 128..255 X blocks; block = (pb: 8 positions, cb: 8 channels)
  const int isx = tid >> 7, blk = tid & 127, pb = blk >> 4, cb8 = blk & 15;
  const int nlc = (wa.L + 63) / 64, nchunks = wa.B * nlc;
  const bf16_t* src;
  long sbs; int sld, srow0, ch0;
  bool live;
  if (!isx) {
    const bool second = p.dy1 && m0 >= p.msplit;
    src = second ? p.dy1 : p.dy0; sbs = p.dy_bs; sld = p.ldy; srow0 = p.dy_row0;
    ch0 = (second ? m0 - p.msplit : m0) + 8 * cb8;
    live = m0 + 8 * cb8 < p.M;
  } else {
    src = p.x; sbs = p.x_bs; sld = p.ldx; srow0 = p.x_row0; ch0 = k0 + 8 * cb8;
    live = k0 + 8 * cb8 < p.K;
  }
  uint4 stg[8];
  auto stage_load = [&](int c) {
    const int b = c / nlc, n = (c - b * nlc) * 64 + 8 * pb;
    const bf16_t* s0 = src + (size_t)b * sbs + (size_t)(srow0 + n) * sld + ch0;
#pragma unroll
    for (int i = 0; i < 8; ++i) stg[i] = live ? *reinterpret_cast<const uint4*>(s0 + (size_t)i * sld) : make_uint4(0, 0, 0, 0);
  };
  auto stage_write = [&]() {
    uint4 t[8];
    transpose8x8(stg, t);
    bf16_t* d = &lds[isx][(8 * cb8) * LDP + 8 * pb];
#pragma unroll
    for (int c = 0; c < 8; ++c) *reinterpret_cast<uint4*>(d + c * LDP) = t[c];
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  stage_load(0);
  stage_write();
  __syncthreads();
  const bf16_t* la = &lds[0][(64 * wm + li) * LDP + 8 * kh];
  const bf16_t* lb = &lds[1][(64 * wk + li) * LDP + 8 * kh];
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    if (more) stage_load(c + 1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      uint4 av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        av[i] = *reinterpret_cast<const uint4*>(la + i * 32 * LDP + 16 * s);
        bv[i] = *reinterpret_cast<const uint4*>(lb + i * 32 * LDP + 16 * s);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_bf16(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();            // every wave is done reading this chunk
    if (more) stage_write();
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + 64 * wk + 32 * j + li;
      if (k >= p.K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 64 * wm + 32 * i + 8 * (r >> 2) + (r & 3) + 4 * kh;
        if (m < p.M) p.out[m * p.o_sm + k * p.o_sk] = acc[i][j][r];
      }
    }
}

// out[m] = sum_{b, n < L} y[b][row0 + n][m]  (bias gradients), fixed summation order; blockIdx.y = problem
struct ColsumProb { const bf16_t* y; long bs; int ld, row0, M; float* out; };
struct ColsumArgs { ColsumProb prob[16]; int B, L; };
__global__ __launch_bounds__(256) void k_colsum(ColsumArgs ca) {
  const ColsumProb& p = ca.prob[blockIdx.y];
  const int m = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  __shared__ float red[4][64];
  float v = 0.0f;
  if (m < p.M)
    for (int b = 0; b < ca.B; ++b)
      for (int n = part; n < ca.L; n += 4) v += bf2f(p.y[(size_t)b * p.bs + (size_t)(p.row0 + n) * p.ld + m]);
  red[part][threadIdx.x & 63] = v;
  __syncthreads();
  if (part == 0 && m < p.M) p.out[m] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---- the <= 8-channel edges of the stack ----------------------------------------------------------------------
// start conv (glow.py:156): h0[b][HALO + n][c] = sum_j Ws[c][j] a0[b][j][n] + bs[c]
__global__ void k_t_start(const float* __restrict__ a0, const float* __restrict__ w, const float* __restrict__ bias,
                          bf16_t* __restrict__ h0, int nin, int L, int Lp) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y, c4 = (threadIdx.x & 63) * 4;
  if (n >= L) return;
  float v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) v[t] = bias[c4 + t];
  for (int j = 0; j < nin; ++j) {
    const float a = a0[((size_t)b * nin + j) * L + n];
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = fmaf(w[(c4 + t) * nin + j], a, v[t]);
  }
  *reinterpret_cast<uint2*>(h0 + ((size_t)b * Lp + HALO + n) * C + c4) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
}

// end conv (glow.py:175): out[b][j][n] = sum_c We[j][c] skip[b][n][c] + be[j]; one wave per position
__global__ __launch_bounds__(256) void k_t_end(const float* __restrict__ skip, const float* __restrict__ w, const float* __restrict__ bias,
                                               float* __restrict__ out, int nout, int L, int Lr) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y, lane = threadIdx.x & 63;
  if (n >= L) return;
  const float4 s = *reinterpret_cast<const float4*>(skip + ((size_t)b * Lr + n) * C + 4 * lane);
  for (int j = 0; j < nout; ++j) {
    const float4 ww = *reinterpret_cast<const float4*>(w + j * C + 4 * lane);
    float v = s.x * ww.x + s.y * ww.y + s.z * ww.z + s.w * ww.w;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) out[((size_t)b * nout + j) * L + n] = v + bias[j];
  }
}

// backward of the end conv w.r.t. its input: dskip[b][n][c] = sum_j We[j][c] dout[b][j][n]  (bf16)
__global__ void k_t_end_bwd(const float* __restrict__ dout, const float* __restrict__ w, bf16_t* __restrict__ dskip, int nout, int L, int Lr) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y, c4 = (threadIdx.x & 63) * 4;
  if (n >= L) return;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < nout; ++j) {
    const float d = dout[((size_t)b * nout + j) * L + n];
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = fmaf(w[j * C + c4 + t], d, v[t]);
  }
  *reinterpret_cast<uint2*>(dskip + ((size_t)b * Lr + n) * C + c4) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
}

// small-channel weight gradients: out[j][c] = sum_{b,n} small[b][j][n] * wide[b][n][c]  (j < nj <= 8, c < 256),
// `wide` fp32 or bf16; plus optional column sums of small (bias of the end conv) and of wide (bias of the start conv).
// Partial sums per workgroup, then a fixed-order sum (deterministic).
template <bool WIDE_BF16>
__global__ __launch_bounds__(256) void k_small_wgrad_part(const float* __restrict__ small, const void* __restrict__ wide, long wide_bs,
                                                          int wide_row0, float* __restrict__ part, int nj, int B, int L, int nparts) {
  const int c = threadIdx.x;
  float acc[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) acc[j] = 0.0f;
  const long total = (long)B * L;
  for (long i = blockIdx.x; i < total; i += nparts) {
    const int b = (int)(i / L), n = (int)(i - (long)b * L);
    float wv;
    if constexpr (WIDE_BF16) wv = bf2f(reinterpret_cast<const bf16_t*>(wide)[(size_t)b * wide_bs + (size_t)(wide_row0 + n) * C + c]);
    else wv = reinterpret_cast<const float*>(wide)[(size_t)b * wide_bs + (size_t)(wide_row0 + n) * C + c];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < nj) acc[j] = fmaf(small[((size_t)b * nj + j) * L + n], wv, acc[j]);
    acc[8] += wv;
  }
#pragma unroll
  for (int j = 0; j < 9; ++j) part[((size_t)blockIdx.x * 9 + j) * C + c] = acc[j];
}
// out_w[(j, c)] laid out by (o_sj, o_sc); out_wsum[c] = column sums of wide (may be null)
__global__ void k_small_wgrad_sum(const float* __restrict__ part, int nparts, int nj, float* __restrict__ out_w, int o_sj, int o_sc,
                                  float* __restrict__ out_wsum) {
  const int c = threadIdx.x, j = blockIdx.x;
  float v = 0.0f;
  for (int p = 0; p < nparts; ++p) v += part[((size_t)p * 9 + j) * C + c];
  if (j < nj) out_w[j * o_sj + c * o_sc] = v;
  else if (j == 8 && out_wsum) out_wsum[c] = v;
}
// out[j] = sum_{b,n} small[b][j][n]
__global__ __launch_bounds__(256) void k_small_rowsum(const float* __restrict__ small, float* __restrict__ out, int nj, int B, int L) {
  const int j = blockIdx.x;
  __shared__ float red[256];
  float v = 0.0f;
  for (int b = 0; b < B; ++b)
    for (int n = threadIdx.x; n < L; n += 256) v += small[((size_t)b * nj + j) * L + n];
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[j] = red[0];
}
// backward of the start conv w.r.t. its input: da0[b][j][n] = sum_c Ws[c][j] dh0[b][n][c]; one wave per position
__global__ __launch_bounds__(256) void k_t_start_bwd(const bf16_t* __restrict__ dh0, const float* __restrict__ w, float* __restrict__ da0,
                                                     int nin, int L, int Lr) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y, lane = threadIdx.x & 63;
  if (n >= L) return;
  const uint2 d = *reinterpret_cast<const uint2*>(dh0 + ((size_t)b * Lr + n) * C + 4 * lane);
  const float dv[4] = {lo2f(d.x), hi2f(d.x), lo2f(d.y), hi2f(d.y)};
  for (int j = 0; j < nin; ++j) {
    float v = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) v = fmaf(w[(4 * lane + t) * nin + j], dv[t], v);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) da0[((size_t)b * nin + j) * L + n] = v;
  }
}

// fp32 channel-major [B][Cn][ldi] (first L columns) -> bf16 position-major [B][Lr][Cn]; rows >= L zero
__global__ void k_to_posmajor_bf16(const float* __restrict__ src, bf16_t* __restrict__ dst, int Cn, int L, int Lr, int ldi) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, n0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, n = n0 + tx;
    tile[r][tx] = (c < Cn && n < L) ? src[((size_t)b * Cn + c) * ldi + n] : 0.0f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int n = n0 + r, c = c0 + tx;
    if (n < Lr && c < Cn) dst[((size_t)b * Lr + n) * Cn + c] = f2bf(tile[tx][r]);
  }
}
// fp32 position-major [B][Lr][Cn] -> fp32 channel-major [B][Cn][ldo] (first L columns)
__global__ void k_from_posmajor_f32(const float* __restrict__ src, float* __restrict__ dst, int Cn, int L, int Lr, int ldo) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, n0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int n = n0 + r, c = c0 + tx;
    tile[r][tx] = (n < L && c < Cn) ? src[((size_t)b * Lr + n) * Cn + c] : 0.0f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, n = n0 + tx;
    if (c < Cn && n < L) dst[((size_t)b * Cn + c) * ldo + n] = tile[tx][r];
  }
}

// ---- layouts of the caller-owned buffers -------------------------------------------------------------------------
struct StateLayout { size_t h, ts, acts, skip, total; size_t h_one, ts_one, acts_one; };
StateLayout state_layout(int nl, int B, int Lr) {
  StateLayout s;
  const int Lp = HALO + Lr + HALO;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  s.h_one = (size_t)B * Lp * C * 2; s.ts_one = (size_t)B * Lr * 2 * C * 2; s.acts_one = (size_t)B * Lr * C * 2;
  s.h = take(s.h_one * (nl + 1)); s.ts = take(s.ts_one * nl); s.acts = take(s.acts_one * nl); s.skip = take((size_t)B * Lr * C * 4);
  s.total = off;
  return s;
}
struct ScratchLayout { size_t w1, w2, rst, int_, condt, dpre, dh, dskip, part, total; size_t w1_one, w2_one, rst_one, int_one, dpre_one, dh_one; };
constexpr int K1 = 3 * C + NCOND;   // 1408
constexpr int SMALL_PARTS = 256;
ScratchLayout scratch_layout(int nl, int B, int Lr) {
  ScratchLayout s;
  const int Lp = HALO + Lr + HALO;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  s.w1_one = (size_t)16 * (K1 / 16) * 64 * 16; s.w2_one = (size_t)16 * (C / 16) * 64 * 16;
  s.rst_one = (size_t)8 * (2 * C / 16) * 64 * 16; s.int_one = (size_t)8 * (3 * 2 * C / 16) * 64 * 16;
  s.dpre_one = (size_t)B * Lp * 2 * C * 2; s.dh_one = (size_t)B * Lr * C * 2;
  s.w1 = take(s.w1_one * nl); s.w2 = take(s.w2_one * nl); s.rst = take(s.rst_one * nl); s.int_ = take(s.int_one * nl);
  s.condt = take((size_t)(NCOND / 32) * (nl * 2 * C / 16) * 64 * 16);
  s.dpre = take(s.dpre_one * nl); s.dh = take(s.dh_one * (nl + 1)); s.dskip = take(s.dh_one);
  s.part = take((size_t)SMALL_PARTS * 9 * C * 4);
  s.total = off;
  return s;
}

int pack_launch(const float* src, uint4* dst, int M, int KG, int k_base, int Cin, int taps, long sm, long sc, long st, long off,
                int gate_rows, hipStream_t s) {
  PackArgs p{src, dst, M, KG, k_base, Cin, taps, gate_rows, sm, sc, st, off};
  const int total = ((M + 31) / 32) * (Cin * taps / 16) * 64;
  k_pack_bf16<<<(total + 255) / 256, 256, 0, s>>>(p);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

int check_wn(const facppg_wn_weights* w, int n_in, int nl, int B, int L) {
  FACPPG_REQUIRE(w && w->start_w && w->start_b && w->end_w && w->end_b, FACPPG_EINVAL, "NULL weight pointer");
  FACPPG_REQUIRE(n_in >= 1 && n_in <= 4 && nl >= 1 && nl <= 8 && B > 0 && B <= 65535 && L > 0, FACPPG_EINVAL, "bad n_in/n_layers/B/L");
  for (int i = 0; i < nl; ++i)
    FACPPG_REQUIRE(w->in_w[i] && w->in_b[i] && w->cond_w[i] && w->cond_b[i] && w->rs_w[i] && w->rs_b[i], FACPPG_EINVAL,
                   "NULL weight pointer (layer %d)", i);
  return FACPPG_OK;
}

__global__ void k_add2(const float* a, const float* b, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}

}  // namespace
}  // namespace facppg

using namespace facppg;

extern "C" int facppg_wn_bf16_padded_len(int L) { return L > 0 ? pad_len(L) : 0; }

extern "C" size_t facppg_wn_bf16_state_bytes(int n_layers, int B, int L) {
  if (n_layers < 1 || n_layers > 8 || B <= 0 || L <= 0) return 0;
  return state_layout(n_layers, B, pad_len(L)).total;
}
extern "C" size_t facppg_wn_bf16_scratch_bytes(int n_layers, int B, int L) {
  if (n_layers < 1 || n_layers > 8 || B <= 0 || L <= 0) return 0;
  return scratch_layout(n_layers, B, pad_len(L)).total + 2 * C * 4 * 8;
}

extern "C" int facppg_spect_to_bf16(const float* spect_dev, int B, int channels, int L, int ld, void* out_dev, void* stream) {
  FACPPG_REQUIRE(spect_dev && out_dev && B > 0 && channels > 0 && L > 0 && ld >= L, FACPPG_EINVAL, "bad argument");
  const int Lr = pad_len(L);
  k_to_posmajor_bf16<<<dim3((Lr + 31) / 32, (channels + 31) / 32, B), 256, 0, (hipStream_t)stream>>>(spect_dev, (bf16_t*)out_dev, channels, L,
                                                                                                      Lr, ld);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

extern "C" int facppg_posmajor_to_f32(const float* src_dev, int B, int channels, int L, float* out_dev, int ld, void* stream) {
  FACPPG_REQUIRE(src_dev && out_dev && B > 0 && channels > 0 && L > 0 && ld >= L, FACPPG_EINVAL, "bad argument");
  const int Lr = pad_len(L);
  k_from_posmajor_f32<<<dim3((L + 31) / 32, (channels + 31) / 32, B), 256, 0, (hipStream_t)stream>>>(src_dev, out_dev, channels, L, Lr, ld);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

// WN.forward (glow.py:154-175) with bf16 MFMA operands, keeping what the backward needs in `state`.
extern "C" int facppg_wn_forward_bf16(const facppg_wn_weights* wts, int n_in, int nl, const float* a0_dev, const void* spect_pm_dev, int B,
                                      int L, float* out_dev, void* state_dev, size_t state_bytes, void* scratch_dev, size_t scratch_bytes,
                                      void* stream_) {
  if (int rc = check_wn(wts, n_in, nl, B, L)) return rc;
  FACPPG_REQUIRE(a0_dev && spect_pm_dev && out_dev && state_dev && scratch_dev, FACPPG_EINVAL, "NULL argument");
  const int Lr = pad_len(L), Lp = HALO + Lr + HALO;
  const StateLayout st = state_layout(nl, B, Lr);
  const ScratchLayout sc = scratch_layout(nl, B, Lr);
  FACPPG_REQUIRE(state_bytes >= st.total, FACPPG_EWORKSPACE, "state has %zu bytes, need %zu", state_bytes, st.total);
  FACPPG_REQUIRE(scratch_bytes >= sc.total + 2 * C * 4 * 8, FACPPG_EWORKSPACE, "scratch has %zu bytes, need %zu", scratch_bytes, sc.total + 2 * C * 4 * 8);
  hipStream_t s = (hipStream_t)stream_;
  char* S = (char*)state_dev;
  char* W = (char*)scratch_dev;
  float* b1 = (float*)(W + sc.total);   // [nl][512] summed biases
  FACPPG_HIP_CHECK(hipMemsetAsync(S + st.h, 0, st.h_one * (nl + 1), s));   // zero margins and rows >= L
  FACPPG_HIP_CHECK(hipMemsetAsync(S + st.ts, 0, st.skip - st.ts, s));      // ts, acts: rows >= L meet zero gradients in k_wgrad, but 0 * NaN = NaN
  for (int i = 0; i < nl; ++i) {
    const int last = i == nl - 1;
    uint4* w1 = (uint4*)(W + sc.w1 + sc.w1_one * i);
    // K order: tap 0 | tap 1 | tap 2 | cond; gate-interleaved rows
    if (int rc = pack_launch(wts->in_w[i], w1, 2 * C, K1 / 16, 0, C, 3, (long)C * 3, 3, 1, 0, 1, s)) return rc;
    if (int rc = pack_launch(wts->cond_w[i], w1, 2 * C, K1 / 16, 3 * C, NCOND, 1, NCOND, 1, 0, 0, 1, s)) return rc;
    if (int rc = pack_launch(wts->rs_w[i], (uint4*)(W + sc.w2 + sc.w2_one * i), last ? C : 2 * C, C / 16, 0, C, 1, C, 1, 0, 0, 0, s)) return rc;
    k_add2<<<2, 256, 0, s>>>(wts->in_b[i], wts->cond_b[i], b1 + 2 * C * i, 2 * C);
  }
  const dim3 egrid((L + 3) / 4, B);
  k_t_start<<<egrid, 256, 0, s>>>(a0_dev, wts->start_w, wts->start_b, (bf16_t*)(S + st.h), n_in, L, Lp);
  for (int i = 0; i < nl; ++i) {
    const int last = i == nl - 1, d = 1 << i;
    const bf16_t* h_in = (const bf16_t*)(S + st.h + st.h_one * i);
    BGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = (const uint4*)(W + sc.w1 + sc.w1_one * i); g.KG = K1 / 16; g.M = 2 * C; g.N = L; g.B = B; g.nseg = 4;
    for (int t = 0; t < 3; ++t) g.seg[t] = Seg{h_in, (long)Lp * C, C, HALO + (t - 1) * d, C};
    g.seg[3] = Seg{(const bf16_t*)spect_pm_dev, (long)Lr * NCOND, NCOND, 0, NCOND};
    g.mode = EP_GATE; g.bias = b1 + 2 * C * i; g.Lr = Lr;
    g.acts = (bf16_t*)(S + st.acts + st.acts_one * i); g.ts = (bf16_t*)(S + st.ts + st.ts_one * i);
    if (int rc = bgemm_launch(g, s)) return rc;
    BGemmArgs r;
    memset(&r, 0, sizeof(r));
    r.A = (const uint4*)(W + sc.w2 + sc.w2_one * i); r.KG = C / 16; r.M = last ? C : 2 * C; r.N = L; r.B = B; r.nseg = 1;
    r.seg[0] = Seg{g.acts, (long)Lr * C, C, 0, C};
    r.mode = EP_RESSKIP; r.bias = wts->rs_b[i]; r.Lr = Lr; r.h_in = h_in; r.h_out = (bf16_t*)(S + st.h + st.h_one * (i + 1));
    r.h_bs = (long)Lp * C; r.h_row0 = HALO; r.skip = (float*)(S + st.skip); r.first = i == 0; r.last = last;
    if (int rc = bgemm_launch(r, s)) return rc;
  }
  k_t_end<<<egrid, 256, 0, s>>>((const float*)(S + st.skip), wts->end_w, wts->end_b, out_dev, 2 * n_in, L, Lr);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

// Complete backward of the stack: da0, dspect (position-major fp32, overwritten) and every weight / bias gradient.
extern "C" int facppg_wn_backward_bf16(const facppg_wn_weights* wts, const facppg_wn_grads* gr, int n_in, int nl, const float* a0_dev,
                                       const void* spect_pm_dev, const float* dout_dev, int B, int L, const void* state_dev,
                                       size_t state_bytes, float* da0_dev, float* dspect_pm_dev, void* scratch_dev, size_t scratch_bytes,
                                       void* stream_) {
  if (int rc = check_wn(wts, n_in, nl, B, L)) return rc;
  FACPPG_REQUIRE(gr && a0_dev && spect_pm_dev && dout_dev && state_dev && da0_dev && dspect_pm_dev && scratch_dev, FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(gr->start_w && gr->start_b && gr->end_w && gr->end_b, FACPPG_EINVAL, "NULL gradient pointer");
  for (int i = 0; i < nl; ++i)
    FACPPG_REQUIRE(gr->in_w[i] && gr->in_b[i] && gr->cond_w[i] && gr->cond_b[i] && gr->rs_w[i] && gr->rs_b[i], FACPPG_EINVAL,
                   "NULL gradient pointer (layer %d)", i);
  const int Lr = pad_len(L), Lp = HALO + Lr + HALO;
  const StateLayout st = state_layout(nl, B, Lr);
  const ScratchLayout sc = scratch_layout(nl, B, Lr);
  FACPPG_REQUIRE(state_bytes >= st.total, FACPPG_EWORKSPACE, "state has %zu bytes, need %zu", state_bytes, st.total);
  FACPPG_REQUIRE(scratch_bytes >= sc.total, FACPPG_EWORKSPACE, "scratch has %zu bytes, need %zu", scratch_bytes, sc.total);
  hipStream_t s = (hipStream_t)stream_;
  const char* S = (const char*)state_dev;
  char* W = (char*)scratch_dev;
  const int nout = 2 * n_in;
  // transposed operand images
  uint4* condt = (uint4*)(W + sc.condt);
  for (int i = 0; i < nl; ++i) {
    const int last = i == nl - 1;
    // dacts[c] = sum_r Wrs[r][c] * [dh_next (res rows) | dskip (skip rows)][r]; last layer: skip rows only
    if (int rc = pack_launch(wts->rs_w[i], (uint4*)(W + sc.rst + sc.rst_one * i), C, (last ? C : 2 * C) / 16, 0, last ? C : 2 * C, 1, 1, C, 0, 0, 0, s))
      return rc;
    // dh[m] += sum_{tap,o} Win[o][m][tap] * dpre[n - (tap-1) d][o]
    if (int rc = pack_launch(wts->in_w[i], (uint4*)(W + sc.int_ + sc.int_one * i), C, 3 * 2 * C / 16, 0, 2 * C, 3, 3, (long)C * 3, 1, 0, 0, s)) return rc;
    // dspect[j] = sum_{i,o} Wcond_i[o][j] * dpre_i[o]: one image, K = nl * 512
    if (int rc = pack_launch(wts->cond_w[i], condt, NCOND, nl * 2 * C / 16, i * 2 * C, 2 * C, 1, 1, NCOND, 0, 0, 0, s)) return rc;
  }
  FACPPG_HIP_CHECK(hipMemsetAsync(W + sc.dpre, 0, sc.dpre_one * nl + 0, s));
  FACPPG_HIP_CHECK(hipMemsetAsync(W + sc.dh, 0, sc.dh_one * (nl + 1), s));
  FACPPG_HIP_CHECK(hipMemsetAsync(W + sc.dskip, 0, sc.dh_one, s));
  bf16_t* dskip = (bf16_t*)(W + sc.dskip);
  const dim3 egrid((L + 3) / 4, B);
  k_t_end_bwd<<<egrid, 256, 0, s>>>(dout_dev, wts->end_w, dskip, nout, L, Lr);
  {  // end conv: weight [nout][256] and bias gradients
    float* part = (float*)(W + sc.part);
    k_small_wgrad_part<false><<<SMALL_PARTS, 256, 0, s>>>(dout_dev, S + st.skip, (long)Lr * C, 0, part, nout, B, L, SMALL_PARTS);
    k_small_wgrad_sum<<<9, 256, 0, s>>>(part, SMALL_PARTS, nout, gr->end_w, C, 1, nullptr);
    k_small_rowsum<<<nout, 256, 0, s>>>(dout_dev, gr->end_b, nout, B, L);
  }
  for (int i = nl - 1; i >= 0; --i) {
    const int last = i == nl - 1, d = 1 << i;
    bf16_t* dpre = (bf16_t*)(W + sc.dpre + sc.dpre_one * i);
    const bf16_t* dh_next = (const bf16_t*)(W + sc.dh + sc.dh_one * (i + 1));
    BGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = (const uint4*)(W + sc.rst + sc.rst_one * i); g.KG = (last ? C : 2 * C) / 16; g.M = C; g.N = L; g.B = B;
    if (last) { g.nseg = 1; g.seg[0] = Seg{dskip, (long)Lr * C, C, 0, C}; }
    else { g.nseg = 2; g.seg[0] = Seg{dh_next, (long)Lr * C, C, 0, C}; g.seg[1] = Seg{dskip, (long)Lr * C, C, 0, C}; }
    g.mode = EP_BWD_GATE; g.Lr = Lr; g.ts = (bf16_t*)(S + st.ts + st.ts_one * i); g.dpre = dpre; g.dpre_bs = (long)Lp * 2 * C;
    if (int rc = bgemm_launch(g, s)) return rc;
    BGemmArgs t;
    memset(&t, 0, sizeof(t));
    t.A = (const uint4*)(W + sc.int_ + sc.int_one * i); t.KG = 3 * 2 * C / 16; t.M = C; t.N = L; t.B = B; t.nseg = 3;
    for (int tp = 0; tp < 3; ++tp) t.seg[tp] = Seg{dpre, (long)Lp * 2 * C, 2 * C, HALO - (tp - 1) * d, 2 * C};
    t.mode = EP_BWD_CONV; t.Lr = Lr; t.dh_next = last ? nullptr : dh_next; t.dh_out = (bf16_t*)(W + sc.dh + sc.dh_one * i);
    if (int rc = bgemm_launch(t, s)) return rc;
  }
  {  // dspect over all layers at once
    BGemmArgs c;
    memset(&c, 0, sizeof(c));
    c.A = condt; c.KG = nl * 2 * C / 16; c.M = NCOND; c.N = L; c.B = B; c.nseg = nl;
    for (int i = 0; i < nl; ++i) c.seg[i] = Seg{(const bf16_t*)(W + sc.dpre + sc.dpre_one * i), (long)Lp * 2 * C, 2 * C, HALO, 2 * C};
    c.mode = EP_ACC_F32; c.Lr = Lr; c.outf = dspect_pm_dev; c.ldo = NCOND; c.accumulate = 0;
    if (int rc = bgemm_launch(c, s)) return rc;
  }
  const bf16_t* dh0 = (const bf16_t*)(W + sc.dh);
  k_t_start_bwd<<<egrid, 256, 0, s>>>(dh0, wts->start_w, da0_dev, n_in, L, Lr);
  {  // start conv: weight [256][n_in] and bias [256] gradients
    float* part = (float*)(W + sc.part);
    k_small_wgrad_part<true><<<SMALL_PARTS, 256, 0, s>>>(a0_dev, dh0, (long)Lr * C, 0, part, n_in, B, L, SMALL_PARTS);
    k_small_wgrad_sum<<<9, 256, 0, s>>>(part, SMALL_PARTS, n_in, gr->start_w, 1, n_in, gr->start_b);
  }
  // weight gradients of the three convs of every layer: NT products over positions, batched over layers (x taps)
  {
    WgradArgs wa;
    memset(&wa, 0, sizeof(wa));
    wa.B = B; wa.L = L; wa.Lr = Lr;
    for (int i = 0; i < nl; ++i)
      for (int tp = 0; tp < 3; ++tp) {
        WgradProb& p = wa.prob[i * 3 + tp];
        p.dy0 = (const bf16_t*)(W + sc.dpre + sc.dpre_one * i); p.dy1 = nullptr; p.dy_bs = (long)Lp * 2 * C; p.ldy = 2 * C; p.dy_row0 = HALO; p.msplit = 0;
        p.x = (const bf16_t*)(S + st.h + st.h_one * i); p.x_bs = (long)Lp * C; p.ldx = C; p.x_row0 = HALO + (tp - 1) * (1 << i);
        p.out = gr->in_w[i] + tp; p.o_sm = (long)C * 3; p.o_sk = 3; p.M = 2 * C; p.K = C;
      }
    k_wgrad<<<dim3(C / 128, 2 * C / 128, nl * 3), 256, 0, s>>>(wa);
    memset(&wa.prob, 0, sizeof(wa.prob));
    for (int i = 0; i < nl; ++i) {
      WgradProb& p = wa.prob[i];
      p.dy0 = (const bf16_t*)(W + sc.dpre + sc.dpre_one * i); p.dy_bs = (long)Lp * 2 * C; p.ldy = 2 * C; p.dy_row0 = HALO;
      p.x = (const bf16_t*)spect_pm_dev; p.x_bs = (long)Lr * NCOND; p.ldx = NCOND; p.x_row0 = 0;
      p.out = gr->cond_w[i]; p.o_sm = NCOND; p.o_sk = 1; p.M = 2 * C; p.K = NCOND;
    }
    k_wgrad<<<dim3(NCOND / 128, 2 * C / 128, nl), 256, 0, s>>>(wa);
    memset(&wa.prob, 0, sizeof(wa.prob));
    for (int i = 0; i < nl; ++i) {
      const int last = i == nl - 1;
      WgradProb& p = wa.prob[i];
      if (last) { p.dy0 = dskip; p.dy1 = nullptr; p.msplit = 0; p.M = C; }
      else { p.dy0 = (const bf16_t*)(W + sc.dh + sc.dh_one * (i + 1)); p.dy1 = dskip; p.msplit = C; p.M = 2 * C; }
      p.dy_bs = (long)Lr * C; p.ldy = C; p.dy_row0 = 0;
      p.x = (const bf16_t*)(S + st.acts + st.acts_one * i); p.x_bs = (long)Lr * C; p.ldx = C; p.x_row0 = 0;
      p.out = gr->rs_w[i]; p.o_sm = C; p.o_sk = 1; p.K = C;
    }
    k_wgrad<<<dim3(C / 128, 2 * C / 128, nl), 256, 0, s>>>(wa);
  }
  {  // bias gradients: column sums of dpre_i (in + cond biases share them) and of [dh_{i+1} | dskip]
    ColsumArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.B = B; ca.L = L;
    for (int i = 0; i < nl; ++i) ca.prob[i] = ColsumProb{(const bf16_t*)(W + sc.dpre + sc.dpre_one * i), (long)Lp * 2 * C, 2 * C, HALO, 2 * C, gr->in_b[i]};
    k_colsum<<<dim3(2 * C / 64, nl), 256, 0, s>>>(ca);
    memset(&ca.prob, 0, sizeof(ca.prob));
    int np = 0;
    for (int i = 0; i < nl; ++i) {
      const int last = i == nl - 1;
      if (!last) ca.prob[np++] = ColsumProb{(const bf16_t*)(W + sc.dh + sc.dh_one * (i + 1)), (long)Lr * C, C, 0, C, gr->rs_b[i]};
      ca.prob[np++] = ColsumProb{dskip, (long)Lr * C, C, 0, C, gr->rs_b[i] + (last ? 0 : C)};
    }
    k_colsum<<<dim3(C / 64, np), 256, 0, s>>>(ca);
    for (int i = 0; i < nl; ++i) FACPPG_HIP_CHECK(hipMemcpyAsync(gr->cond_b[i], gr->in_b[i], 2 * C * 4, hipMemcpyDeviceToDevice, s));
  }
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}
