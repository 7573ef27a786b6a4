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
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <vector>

#include "facppg_common.h"
#include "facppg_gemm.h"

namespace facppg {
namespace {

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nt_load16(const void* p) {
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

constexpr int C = 256, NCOND = 640, HALO = 128;
constexpr int BM = 128, BN = 128, KC = 64;     // k_bgemm tile and K chunk
constexpr int LDB = KC + 8;                    // LDS row pitch in bf16 (144 B: conflict-free ds_read_b128)
constexpr int MAXSEG = 8;
// padded positions per batch item: a multiple of the 128-column GEMM tile, so a tile's staging loads never leave the
// batch item's rows (rows >= L are zero)
__host__ __device__ inline int pad_len(int L) { return round_up(L, BN); }

// fp32 -> bf16, round to nearest even: v_cvt_pk_bf16_f32 (gfx950), one instruction per PAIR (the integer sequence -- bfe, add3,
// and, perm -- was a fifth of the gate epilogue's instructions)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2(f, 0.0f) & 0xffffu); }
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
// the WN gate (glow.py:8-14 fused_add_tanh_sigmoid_multiply): T = tanh(x) = (1 - e^-2x) / (1 + e^-2x), S = sigmoid(y); v_exp_f32
// and v_rcp_f32 directly (__fdividef compiles to the full IEEE division sequence: ~10 instructions per quotient, half of the
// gate epilogue).  Shared by k_bgemm<EP_GATE> and k_wn_fwd, which must agree bit for bit.
__device__ __forceinline__ void gate_ts(float x, float y, float& T, float& S) {
  const float ea = __builtin_amdgcn_exp2f(fminf(fmaxf(x, -15.0f), 15.0f) * -2.8853900817779268f);
  T = (1.0f - ea) * __builtin_amdgcn_rcpf(1.0f + ea);
  S = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y * -1.4426950408889634f));
}
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
// accumulator r of lane half kh in gate row block mb <-> bias entry: tanh rows r < 8, sigmoid rows r >= 8 (+ C)
__device__ __forceinline__ void gate_bias_rows(const float* __restrict__ bias, int mb, int kh, float (&bi)[16]) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float4 bt = *reinterpret_cast<const float4*>(bias + 16 * mb + 8 * q + 4 * kh);
    const float4 bs = *reinterpret_cast<const float4*>(bias + C + 16 * mb + 8 * q + 4 * kh);
    bi[4 * q] = bt.x; bi[4 * q + 1] = bt.y; bi[4 * q + 2] = bt.z; bi[4 * q + 3] = bt.w;
    bi[8 + 4 * q] = bs.x; bi[8 + 4 * q + 1] = bs.y; bi[8 + 4 * q + 2] = bs.z; bi[8 + 4 * q + 3] = bs.w;
  }
}
// d(tanh * sigmoid) w.r.t. the two pre-activations (shared by k_bgemm<EP_BWD_GATE> and k_wn_bwd, which agree bit for bit)
__device__ __forceinline__ void gate_bwd(float v, float T, float S, float& dt, float& ds) {
  dt = v * S * (1.0f - T * T);
  ds = v * T * S * (1.0f - S);
}
__device__ __forceinline__ int gate_row_src(int mb, int rho) { return (rho >= 16 ? C : 0) + 16 * mb + ((rho >> 3) & 1) * 8 + (rho & 7); }

constexpr int MAXPACK = 48;   // (one flow's images of both directions; 48 x 72 B of kernel arguments)
struct PackBatch { PackArgs e[MAXPACK]; };
__global__ void k_pack_bf16(PackBatch pb) {
  const PackArgs& p = pb.e[blockIdx.y];
  const int ng = p.Cin * p.taps / 16;                       // k16 groups written by this entry
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

// The same images for sources whose rows are contiguous along the reduction (sc == taps, st == 1: the forward images).  k_pack_bf16
// gives a lane one ROW, i.e. 64 lanes read 64 different cache lines per load; here consecutive threads take consecutive 8-channel
// groups of one row -- a thread reads 8 * taps contiguous floats (16-byte loads) and writes one 16-byte piece per tap.
__global__ void k_pack_rows_bf16(PackBatch pb) {
  const PackArgs& p = pb.e[blockIdx.y];
  const int c8n = p.Cin / 8, MB = (p.M + 31) / 32;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= MB * 32 * c8n) return;
  const int c8 = idx % c8n, rr = idx / c8n, mb = rr >> 5, rho = rr & 31;
  const int m = p.gate_rows ? gate_row_src(mb, rho) : mb * 32 + rho;
  float v[24];
  const int nf = 8 * p.taps;            // taps <= 3
  if (m < p.M) {
    const float4* src = reinterpret_cast<const float4*>(p.src + p.off + (size_t)m * p.sm + (size_t)c8 * nf);
#pragma unroll
    for (int q = 0; q < 6; ++q)
      if (q * 4 < nf) {
        const float4 x = src[q];
        v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
      }
  } else {
#pragma unroll
    for (int q = 0; q < 24; ++q) v[q] = 0.0f;
  }
#pragma unroll
  for (int tap = 0; tap < 3; ++tap)
    if (tap < p.taps) {
      float e[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) e[i] = p.taps == 1 ? v[i] : (p.taps == 3 ? v[(3 * i + tap) % 24] : v[(2 * i + tap) % 24]);
      const int k = p.k_base + tap * p.Cin + c8 * 8;
      p.dst[((size_t)mb * p.KG + (k >> 4)) * 64 + rho + 32 * ((k >> 3) & 1)] = make_uint4(pack2(e[0], e[1]), pack2(e[2], e[3]), pack2(e[4], e[5]), pack2(e[6], e[7]));
    }
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
      gate_ts(v[t], v2[t], T[t], S[t]);
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
    for (int t = 0; t < 4; ++t) gate_bwd(v[t], T[t], S[t], dt[t], ds[t]);
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

// NCB = 32-column blocks per tile: 4 (128 columns) once that gives >= 1.5 workgroups per CU, else 2 (64 columns: twice
// the workgroups, half the LDS and accumulators each).  The operands of chunk c+PD-1 (weights) / c+PD (activations) are
// requested before the MFMAs of chunk c and held in a statically indexed register ring (the chunk loop is unrolled by PD);
// workgroups that share a column tile (the M tiles) are dealt to the SAME XCD back to back, so all but the first find
// the activation tile in that XCD's L2.
// Measured (rocprofv3, B = 12, gate GEMM 21.6 GFLOP): non-temporal loads for the ACTIVATION operand (streamed once per
// workgroup) 66 -> 41 us -- they stop evicting the weight images, which every workgroup re-reads, from L1/L2; non-temporal
// WEIGHT loads are slower (76 us).  With them: PD 1 / 2 = 41.6 / 40.9 us (gate), 29.1 / 26.6 us (transposed conv); the XCD-aware
// order 48 -> 41.6 us; before them deeper staging only made things worse (PD 1 / 2 / 3 / 4: 68 / 82 / 86 / 91 us) -- more
// streaming lines in flight evicted more weights.  k_wgrad's operands are re-read by neighbouring tiles: non-temporal loads
// there cost 10 % (230 vs 209 us), so it keeps plain loads.
constexpr int PD = 2;   // chunks requested ahead; 3 / 4 / 6 re-measured in round 3: 12.3 / 12.3 / 12.6 ms at batch 3 (12.3 with 2), 25.2 / 25.0 / 26.7 at batch 12 (24.4)
template <int MODE, int NCB>
__global__ __launch_bounds__(256) void k_bgemm(BGemmArgs p) {
  constexpr int BNt = 32 * NCB;
  __shared__ __attribute__((aligned(16))) bf16_t lds[2][BNt * LDB];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, kh = lane >> 5;
  // workgroup lin runs on XCD lin % 8; slot = lin / 8 walks (M tile fastest, then this XCD's column tiles)
  const int nm = (p.M + BM - 1) / BM, ncol = (p.N + BNt - 1) / BNt;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ct = (slot / nm) * 8 + xcd, mt = slot % nm;
  if (ct >= ncol * p.B) return;
  const int b = ct / ncol, n0 = (ct - b * ncol) * BNt;
  const int mb = mt * 4 + w;
  const bool active = mb * 32 < p.M;
  const uint4* ap = p.A + (size_t)(active ? mb : 0) * p.KG * 64 + lane;
  // staging: thread -> rows srow + 32*j (j < NCB) of the [BNt positions][64 k] chunk, 16 bytes at k = 8*sk
  const int srow = tid >> 3, sk = tid & 7;
  int nchunks = 0;
  for (int s = 0; s < p.nseg; ++s) nchunks += p.seg[s].nch / KC;

  f32x16 acc[NCB];
  if constexpr (MODE == EP_GATE) {
    // the gate's accumulators START from the bias (b_in + b_cond) of their rows -- as k_wn_fwd's do, bit for bit
    float bi[16];
    gate_bias_rows(p.bias, active ? mb : 0, kh, bi);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cb][r] = bi[r];
  } else {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
  }

  // the chunk whose loads are issued next: (segment, channel offset), walked incrementally
  int seg_i = 0, seg_c = 0;
  auto stage_load = [&](uint4 (&stg)[NCB]) {
    const Seg& sg = p.seg[seg_i];
    const bf16_t* base = sg.x + (size_t)b * sg.bs + (size_t)(sg.row0 + n0 + srow) * sg.ld + seg_c + 8 * sk;
#pragma unroll
    for (int j = 0; j < NCB; ++j) stg[j] = nt_load16(base + (size_t)(32 * j) * sg.ld);
    // advance to the next chunk; at the very end stay on the last one: the requests of the last iterations are issued
    // UNCONDITIONALLY (they re-read the last chunk, nothing consumes them) -- a branch around a load makes hipcc's waitcnt pass
    // merge the two paths to vmcnt(0) at the next use, which drains every prefetch and exposes a full memory round trip per chunk
    if (seg_c + KC < sg.nch) seg_c += KC;
    else if (seg_i + 1 < p.nseg) { ++seg_i; seg_c = 0; }
  };
  auto stage_write = [&](int buf, const uint4 (&stg)[NCB]) {
#pragma unroll
    for (int j = 0; j < NCB; ++j) *reinterpret_cast<uint4*>(&lds[buf][(srow + 32 * j) * LDB + 8 * sk]) = stg[j];
  };
  auto load_a = [&](uint4 (&a)[4], int c) {
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = ap[(size_t)(min(c, nchunks - 1) * 4 + s) * 64];
  };
  auto compute = [&](int c, const uint4 (&a)[4]) {
    const bf16_t* lb = &lds[c & 1][li * LDB + 8 * kh];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
        acc[cb] = mfma_bf16(a[s], *reinterpret_cast<const uint4*>(lb + cb * 32 * LDB + 16 * s), acc[cb]);
  };
  // rings: chunk c lives in slot c % PD.  Activations of chunk c leave their slot for LDS at the end of iteration c-1,
  // so iteration c refills that slot with chunk c+PD; weights of chunk c are used during iteration c, so iteration c
  // refills the slot of chunk c-1 with chunk c+PD-1.
  uint4 st[PD][NCB], ar[PD][4];
  stage_load(st[0]);
  stage_write(0, st[0]);
#pragma unroll
  for (int u = 1; u < PD; ++u) stage_load(st[u]);
#pragma unroll
  for (int u = 0; u < PD - 1; ++u) load_a(ar[u], u);
  __syncthreads();
  // the chunk loop runs in whole groups of PD iterations (static ring indices); iterations past the last chunk multiply the
  // re-read last chunk by ... nothing: their MFMAs are skipped by a uniform branch that contains no memory operation
  for (int c0 = 0; c0 < nchunks; c0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const int c = c0 + u;
      load_a(ar[(u + PD - 1) % PD], c + PD - 1);
      stage_load(st[u]);
      __builtin_amdgcn_sched_barrier(0);   // keep the requests HERE, PD chunks ahead (hipcc otherwise sinks them next to their use)
      if (c < nchunks) compute(c, ar[u]);
      __builtin_amdgcn_sched_barrier(0);
      stage_write((c + 1) & 1, st[(u + 1) % PD]);
      __syncthreads();
    }
  }
  if (!active) return;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
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

template <int MODE>
void bgemm_dispatch(const BGemmArgs& a, hipStream_t s) {
  const int nm = (a.M + BM - 1) / BM;
  const long wide = (long)((a.N + 127) / 128) * nm * a.B;
  // 128-column tiles once they give >= 1.5 workgroups per CU, else 64-column tiles (twice the workgroups).  A 256 x 128 /
  // 8-wave tile with both operands through LDS was built and measured slower in every mode (profiles/r02_experiments.txt).
  if (wide >= 384) {
    const int ct = ((a.N + 127) / 128) * a.B;
    k_bgemm<MODE, 4><<<dim3(8 * nm * ((ct + 7) / 8)), 256, 0, s>>>(a);
  } else {
    const int ct = ((a.N + 63) / 64) * a.B;
    k_bgemm<MODE, 2><<<dim3(8 * nm * ((ct + 7) / 8)), 256, 0, s>>>(a);
  }
}

int bgemm_launch(const BGemmArgs& a, hipStream_t s) {
  int K = 0;
  for (int i = 0; i < a.nseg; ++i) {
    FACPPG_REQUIRE(a.seg[i].nch % KC == 0 && a.seg[i].nch > 0, FACPPG_EINVAL, "bgemm: segment of %d channels", a.seg[i].nch);
    K += a.seg[i].nch;
  }
  FACPPG_REQUIRE(K == a.KG * 16 && a.M % 32 == 0 && a.nseg >= 1 && a.nseg <= MAXSEG, FACPPG_EINVAL, "bgemm: bad shape K=%d KG=%d M=%d", K, a.KG, a.M);
  switch (a.mode) {
    case EP_GATE: bgemm_dispatch<EP_GATE>(a, s); break;
    case EP_RESSKIP: bgemm_dispatch<EP_RESSKIP>(a, s); break;
    case EP_BWD_GATE: bgemm_dispatch<EP_BWD_GATE>(a, s); break;
    case EP_BWD_CONV: bgemm_dispatch<EP_BWD_CONV>(a, s); break;
    default: bgemm_dispatch<EP_ACC_F32>(a, s); break;
  }
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

// ------------------------------------------------------------------------------------------------------------
// k_wn_fwd: ONE launch per WN layer of the training forward (glow.py:160-173): gate GEMM [512 x 1408] -> tanh * sigmoid
// -> res/skip GEMM [512 x 256] for a tile of TN (64 or 32) positions, the gated tile handed over in LDS.  Same operand images, same
// K order (tap 0 | tap 1 | tap 2 | cond) and the same 16-entry MFMA steps as k_bgemm<EP_GATE> followed by
// k_bgemm<EP_RESSKIP>, so every stored value (acts, ts, h_out, skip) has the SAME BITS as the two-launch path
// (tests/test_gpu_train_bf16.py).  Shape: 8 waves, wave w owns gate row blocks 2w, 2w+1 (channels 32w .. 32w+31, both halves) x
// 64 positions: 4 MFMAs per pair of A fragments (global -> registers, a ring refilled one 128-entry chunk ahead) and pair
// of B fragments (LDS [position][k], pitch 272 B): half the LDS reads per MFMA of the 32 x 128 wave tile of k_bgemm, one
// barrier per 128 reduction entries instead of per 64, one launch ramp per layer instead of two.
// ------------------------------------------------------------------------------------------------------------
// debugging aid shared by the two fused kernels: per-phase wall_clock64 stamps of every tile, appended to the file
// The *_STAMPS debugging aids allocate, synchronise and free inside the launch helper: not while the stream is being captured into
// a HIP graph (waveglow.graphed.GraphedTrainStep) -- there the variable is ignored.
const char* stamps_path(const char* name, hipStream_t s) {
  const char* path = getenv(name);
  if (!path) return nullptr;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
  return path;
}
int dump_stamps(const char* path, const char* what, const unsigned long long* dev, int ntiles, int per_tile, int d, hipStream_t s) {
  std::vector<unsigned long long> h((size_t)ntiles * per_tile);
  FACPPG_HIP_CHECK(hipMemcpyAsync(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost, s));
  FACPPG_HIP_CHECK(hipStreamSynchronize(s));
  if (FILE* f = fopen(path, "a")) {
    fprintf(f, "launch %s ntiles %d d %d\n", what, ntiles, d);
    for (int t = 0; t < ntiles; ++t) {
      for (int j = 0; j < per_tile; ++j) fprintf(f, "%llu ", h[(size_t)t * per_tile + j] ? h[(size_t)t * per_tile + j] - h[(size_t)t * per_tile] : 0ull);
      fprintf(f, "\n");
    }
    fclose(f);
  }
  return FACPPG_OK;
}

constexpr int FKC = 128, FLDB = FKC + 8, FLDA = C + 8, FLDT = 2 * C + 8, FLDO = 2 * C + 4;
// LDS of k_wn_fwd: TN * FLDO * 4 bytes: staging (TN * 544) -> gated + tanh|sigmoid tiles (TN * 1568) -> fp32 res/skip tile (TN * 2064)
constexpr int FNCH = (3 * C + NCOND) / FKC;   // 11 chunks
#ifndef FACPPG_FWD_RING
#define FACPPG_FWD_RING 8
#endif
constexpr int FRD = FACPPG_FWD_RING;          // A-fragment ring of k_wn_fwd, in 16-entry steps
struct WnFwdArgs {
  const uint4* A1;          // gate image, KG = 88, gate-interleaved rows
  const uint4* A2;          // res/skip image, KG = 16, M2 rows
  const bf16_t* h_in; bf16_t* h_out; long h_bs;     // [B][HALO + Lr + HALO][256]
  const bf16_t* spect; long sp_bs;                  // [B][Lr][640]
  const float* b1; const float* b2;
  bf16_t* acts; bf16_t* ts; float* skip;            // [B][Lr][256 | 512 | 256]
  int L, Lr, B, d, first, last, ntiles;
  unsigned long long* stamps;   // STAMP builds only: [tile][16] wall_clock64 (100 MHz) values of wave 0
};

// TN positions per tile: 64, or 32 when 64-position tiles would leave most of the chip idle (batch 3: 60 tiles)
template <int TN, bool STAMP>
__global__ __launch_bounds__(512) void k_wn_fwd(WnFwdArgs p) {
  constexpr int NCB = TN / 32;      // 32-position column blocks per wave
  // staging [2][TN][FLDB] during the gate GEMM; then the gated tile [TN][FLDA] + tanh|sigmoid [TN][FLDT]; then fp32 [TN][FLDO]
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  static_assert((size_t)FLDO * 4 >= (size_t)2 * FLDB * 2 && (size_t)FLDO * 4 >= (size_t)(FLDA + FLDT) * 2, "LDS phases");
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, kh = lane >> 5;
  // consecutive tiles (which share tap rows) go to the same XCD: workgroup lin runs on XCD lin % 8
  const int per = (p.ntiles + 7) >> 3;
  const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (tile >= p.ntiles) return;
  const int ncol = (p.L + TN - 1) / TN;
  const int b = tile / ncol, n0 = (tile - b * ncol) * TN;
  int stamp_i = 0;
  auto stamp = [&]() {
    if constexpr (STAMP) {
      if (tid == 0) p.stamps[(size_t)tile * 16 + stamp_i] = wall_clock64();
      ++stamp_i;
    }
  };
  stamp();
  constexpr int KG1 = (3 * C + NCOND) / 16, KG2 = C / 16;
  const uint4* ap0 = p.A1 + (size_t)(2 * w) * KG1 * 64 + lane;
  const uint4* ap1 = ap0 + (size_t)KG1 * 64;

  f32x16 acc[2][NCB];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float bi[16];     // the accumulators start from the summed biases of their rows
    gate_bias_rows(p.b1, 2 * w + i, kh, bi);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][cb][r] = bi[r];
  }

  // staging: a chunk is TN rows of 16 sixteen-byte pieces; thread -> pieces tid + 512 j (16 lanes per row)
  auto stage_load = [&](int c, uint4 (&stg)[NCB]) {
    c = min(c, FNCH - 1);       // the last iterations re-read the last chunk (unconditional requests, see k_bgemm)
#pragma unroll
    for (int j = 0; j < NCB; ++j) {
      const int e = tid + 512 * j, row = e >> 4, k8 = (e & 15) * 8;
      const bf16_t* src;
      if (c < 6) src = p.h_in + (size_t)b * p.h_bs + (size_t)(HALO + ((c >> 1) - 1) * p.d + n0 + row) * C + (c & 1) * FKC + k8;
      else src = p.spect + (size_t)b * p.sp_bs + (size_t)(n0 + row) * NCOND + (c - 6) * FKC + k8;
      stg[j] = nt_load16(src);
    }
  };
  auto stage_write = [&](int buf, const uint4 (&stg)[NCB]) {
#pragma unroll
    for (int j = 0; j < NCB; ++j) {
      const int e = tid + 512 * j;
      *reinterpret_cast<uint4*>(lds + buf * (TN * FLDB) + (e >> 4) * FLDB + (e & 15) * 8) = stg[j];
    }
  };
  // activation chunks are requested TWO chunks ahead (register ring st[chunk & 1]); the A fragments FRD 16-entry steps ahead (ring
  // ar[step % FRD]: 8 = one chunk -- the fragment of step t is requested 0.85 us of MFMA work before its use, less than an L2 round trip
  // under load, and half of every wave's time was s_waitcnt -- 12 / 16 = one and a half / two chunks)
  uint4 st[2][NCB], ar[FRD][2];
  stage_load(0, st[0]);
#pragma unroll
  for (int s = 0; s < FRD; ++s) { ar[s][0] = ap0[(size_t)s * 64]; ar[s][1] = ap1[(size_t)s * 64]; }
  stage_load(1, st[1]);
  stage_write(0, st[0]);
  stage_load(2, st[0]);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < FNCH; ++c) {
    // chunk c+1 (requested two iterations ago) goes to the other buffer, whose last readers left through the barrier below
    stage_write((c + 1) & 1, st[(c + 1) & 1]);
    stage_load(c + 3, st[(c + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);   // requests stay where they are written (hipcc otherwise sinks them to the end of the iteration, next to their use)
    const bf16_t* lb = lds + (c & 1) * (TN * FLDB) + li * FLDB + 8 * kh;
    uint4 bc[NCB], bn[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) bc[cb] = *reinterpret_cast<const uint4*>(lb + cb * 32 * FLDB);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)   // fragments of the next step requested before this step's MFMAs
        bn[cb] = s < 7 ? *reinterpret_cast<const uint4*>(lb + cb * 32 * FLDB + 16 * (s + 1)) : bc[cb];
      const int t = c * 8 + s;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[i][cb] = mfma_bf16(ar[t % FRD][i], bc[cb], acc[i][cb]);
      const size_t gnext = (size_t)min(t + FRD, FNCH * 8 - 1) * 64;     // (the last steps re-request the last fragment: unconditional requests)
      ar[t % FRD][0] = ap0[gnext];
      ar[t % FRD][1] = ap1[gnext];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) bc[cb] = bn[cb];
    }
    __syncthreads();
    stamp();
  }
  // res/skip image: first half of the reduction requested now, under the gate arithmetic
  const int M2 = p.last ? C : 2 * C;
  const bool active2 = 64 * w < M2;
  const uint4* rp0 = p.A2 + (size_t)(active2 ? 2 * w : 0) * KG2 * 64 + lane;
  const uint4* rp1 = rp0 + (size_t)KG2 * 64;
  uint4 a2[8][2];
#pragma unroll
  for (int g = 0; g < 8; ++g) { a2[g][0] = rp0[(size_t)g * 64]; a2[g][1] = rp1[(size_t)g * 64]; }
  const int nrows = min(TN, p.L - n0);    // rows >= L are never written (they stay zero)
  // what the second epilogue adds to, requested now: h_in rows (8 channels per piece) and the running skip sum (4 per piece)
  const size_t hrow0 = (size_t)b * p.h_bs + (size_t)(HALO + n0) * C, srow0 = ((size_t)b * p.Lr + n0) * C;
  uint4 hin[2 * NCB];
  float4 sold[4 * NCB];
#pragma unroll
  for (int j = 0; j < 2 * NCB; ++j) hin[j] = *reinterpret_cast<const uint4*>(p.h_in + hrow0 + (size_t)(tid + 512 * j) * 8);   // (rows < Lr exist; unused ones are never stored)
#pragma unroll
  for (int j = 0; j < 4 * NCB; ++j) sold[j] = *reinterpret_cast<const float4*>(p.skip + srow0 + (size_t)(tid + 512 * j) * 4);
  // gate: lane holds tanh rows (acc 0..7) and the matching sigmoid rows (acc 8..15) of channels 16 mb + 8 q + 4 kh + {0..3}.
  // Everything goes to LDS tiles [position][channel] first and leaves the CU as whole rows (16 bytes per lane, a tile is ONE
  // contiguous block of acts / ts): the lane-per-position stores of 8 bytes cost 7 - 13 us per tile, more than the GEMM.
  bf16_t* lts = lds + TN * FLDA;          // [TN][FLDT] tanh | sigmoid
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int ch = 16 * (2 * w + i) + 8 * q + 4 * kh;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        float T[4], S[4], a[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          gate_ts(acc[i][cb][4 * q + t], acc[i][cb][8 + 4 * q + t], T[t], S[t]);
          a[t] = T[t] * S[t];
        }
        const int row = cb * 32 + li;
        *reinterpret_cast<uint2*>(lds + row * FLDA + ch) = make_uint2(pack2(a[0], a[1]), pack2(a[2], a[3]));
        *reinterpret_cast<uint2*>(lts + row * FLDT + ch) = make_uint2(pack2(T[0], T[1]), pack2(T[2], T[3]));
        *reinterpret_cast<uint2*>(lts + row * FLDT + C + ch) = make_uint2(pack2(S[0], S[1]), pack2(S[2], S[3]));
      }
    }
  __syncthreads();
  stamp();
  if (active2) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][cb][r] = 0.0f;
    const bf16_t* la = lds + li * FLDA + 8 * kh;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      uint4 bf[NCB];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) bf[cb] = *reinterpret_cast<const uint4*>(la + cb * 32 * FLDA + 16 * g);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[i][cb] = mfma_bf16(a2[g & 7][i], bf[cb], acc[i][cb]);
      if (g < 8) { a2[g][0] = rp0[(size_t)(g + 8) * 64]; a2[g][1] = rp1[(size_t)(g + 8) * 64]; }
    }
  }
  // acts / ts leave as whole rows (fire and forget) behind the second GEMM's MFMAs
  {
    bf16_t* ga = p.acts + srow0;
    bf16_t* gt = p.ts + 2 * srow0;
#pragma unroll
    for (int j = 0; j < 2 * NCB; ++j) {
      const int e = tid + 512 * j, row = e >> 5, c8 = (e & 31) * 8;
      if (row < nrows) *reinterpret_cast<uint4*>(ga + (size_t)e * 8) = *reinterpret_cast<const uint4*>(lds + row * FLDA + c8);
    }
#pragma unroll
    for (int j = 0; j < 4 * NCB; ++j) {
      const int e = tid + 512 * j, row = e >> 6, c8 = (e & 63) * 8;
      if (row < nrows) *reinterpret_cast<uint4*>(gt + (size_t)e * 8) = *reinterpret_cast<const uint4*>(lts + row * FLDT + c8);
    }
  }
  __syncthreads();      // every wave is through with the gated tile: the fp32 tile [TN][FLDO] of v + bias takes the LDS over
  stamp();
  float* lo = reinterpret_cast<float*>(lds);
  if (active2) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = 32 * (2 * w + i) + 8 * q + 4 * kh;      // rows m .. m+3 of this lane
        const float4 bb = *reinterpret_cast<const float4*>(p.b2 + m);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          *reinterpret_cast<float4*>(lo + (cb * 32 + li) * FLDO + m) =
              make_float4(acc[i][cb][4 * q] + bb.x, acc[i][cb][4 * q + 1] + bb.y, acc[i][cb][4 * q + 2] + bb.z, acc[i][cb][4 * q + 3] + bb.w);
      }
  }
  __syncthreads();
  // res rows (< 256 unless last): h_out = (res + bias) + h_in; skip rows: skip = skip + (v + bias) -- whole rows per wave
  if (!p.last) {
    bf16_t* gh = p.h_out + hrow0;
#pragma unroll
    for (int j = 0; j < 2 * NCB; ++j) {
      const int e = tid + 512 * j, row = e >> 5, c8 = (e & 31) * 8;
      if (row >= nrows) continue;
      const float4 r0 = *reinterpret_cast<const float4*>(lo + row * FLDO + c8), r1 = *reinterpret_cast<const float4*>(lo + row * FLDO + c8 + 4);
      const uint4 h = hin[j];
      *reinterpret_cast<uint4*>(gh + (size_t)e * 8) =
          make_uint4(pack2(r0.x + lo2f(h.x), r0.y + hi2f(h.x)), pack2(r0.z + lo2f(h.y), r0.w + hi2f(h.y)),
                     pack2(r1.x + lo2f(h.z), r1.y + hi2f(h.z)), pack2(r1.z + lo2f(h.w), r1.w + hi2f(h.w)));
    }
  }
  {
    float* gs = p.skip + srow0;
    const int c0 = p.last ? 0 : C;
#pragma unroll
    for (int j = 0; j < 4 * NCB; ++j) {
      const int e = tid + 512 * j, row = e >> 6, c4 = (e & 63) * 4;
      if (row >= nrows) continue;
      const float4 v = *reinterpret_cast<const float4*>(lo + row * FLDO + c0 + c4);
      float4 sv = p.first ? make_float4(0.f, 0.f, 0.f, 0.f) : sold[j];
      sv.x += v.x; sv.y += v.y; sv.z += v.z; sv.w += v.w;
      *reinterpret_cast<float4*>(gs + (size_t)e * 4) = sv;
    }
  }
  stamp();
  if constexpr (STAMP) { __builtin_amdgcn_s_waitcnt(0); stamp(); }
}

template <int TN>
int wn_fwd_launch_t(WnFwdArgs& a, hipStream_t s) {
  constexpr size_t ldsb = (size_t)TN * FLDO * 4;     // 132 096 B (64 positions) / 66 048 B (32)
  a.ntiles = ((a.L + TN - 1) / TN) * a.B;
  const int per = (a.ntiles + 7) / 8;
  {
    // > 64 KB of dynamic LDS needs the attribute on every device the kernel runs on (see k_wgrad)
    static std::atomic<unsigned long long> attr_devices{0};
    int dev = 0;
    FACPPG_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 64 || !((attr_devices.load(std::memory_order_relaxed) >> dev) & 1ull)) {
      FACPPG_HIP_CHECK(hipFuncSetAttribute((const void*)k_wn_fwd<TN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
      FACPPG_HIP_CHECK(hipFuncSetAttribute((const void*)k_wn_fwd<TN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
      if (dev < 64) attr_devices.fetch_or(1ull << dev, std::memory_order_relaxed);
    }
  }
  if (const char* path = stamps_path("FACPPG_WN_FWD_STAMPS", s)) {   // debugging aid: per-phase stamps of every tile, appended to `path`
    unsigned long long* d = nullptr;
    FACPPG_HIP_CHECK(hipMalloc(&d, (size_t)a.ntiles * 16 * 8));
    FACPPG_HIP_CHECK(hipMemsetAsync(d, 0, (size_t)a.ntiles * 16 * 8, s));
    a.stamps = d;
    k_wn_fwd<TN, true><<<dim3(8 * per), 512, ldsb, s>>>(a);
    const int rc = dump_stamps(path, "fwd", d, a.ntiles, 16, a.d, s);
    FACPPG_HIP_CHECK(hipFree(d));
    return rc;
  }
  k_wn_fwd<TN, false><<<dim3(8 * per), 512, ldsb, s>>>(a);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}
// 64-position tiles once they give 160 workgroups, else 32-position tiles (FACPPG_TRAIN_TILE=64 / 32 forces one)
int tile_positions(int B, int L) {
  if (const char* e = getenv("FACPPG_TRAIN_TILE")) return atoi(e) == 32 ? 32 : 64;
  return (long)((L + 63) / 64) * B >= 160 ? 64 : 32;
}
int wn_fwd_launch(WnFwdArgs& a, hipStream_t s) {
  return tile_positions(a.B, a.L) == 64 ? wn_fwd_launch_t<64>(a, s) : wn_fwd_launch_t<32>(a, s);
}
// One launch per layer once its tiles fill most of the chip (a tile streams ALL of the layer's weights into its CU: with few
// tiles the two-launch layer, which deals the weight rows over four times as many workgroups, is as fast or faster).  Measured,
// whole step, segment 10 000: batch 3 (120 tiles of 32 positions) 10.25 ms fused / 10.20 two launches; batch 6 (240 of 32) 12.7 /
// 13.6; batch 12 (240 of 64) 18.5 / 20.4.  FACPPG_TRAIN_FUSED_FWD=1 / 0 forces either path (the bit-equality test and A/B timing).
bool fused_fwd_enabled(int B, int L) {
  if (const char* e = getenv("FACPPG_TRAIN_FUSED_FWD")) return e[0] != '0';
  return (long)((L + 31) / 32) * B >= 160;
}

// ------------------------------------------------------------------------------------------------------------
// k_wn_bwd: ONE launch per WN layer of the backward's data-gradient chain: the transposed dilated conv of layer i
//   dh_i = dh_{i+1} + sum_tap Win_i[:, :, tap]^T dpre_i(n - (tap - 1) d)                    [256 x 1536]
// and, on the same positions (the res/skip conv is 1 x 1), the gate backward of layer i-1
//   dpre_{i-1} = gate'(ts_{i-1}) * Wrs_{i-1}^T [dh_i ; dskip]                              [256 x 512]
// with the dh_i tile handed over in LDS -- the two launches k_bgemm<EP_BWD_CONV>(i), k_bgemm<EP_BWD_GATE>(i-1) of the
// layer loop, same operand images, same K order, same bits.  Shape as k_wn_fwd: tiles of TN positions, 8 waves x 32 rows,
// A fragments from global through a register ring one 128-entry chunk ahead, B operand through LDS.  What the epilogues
// add to / multiply with (dh_{i+1}, tanh | sigmoid of layer i-1) is brought into LDS as whole rows at the start, updated
// IN PLACE by the lanes that own the cells, and leaves as whole rows.
// ------------------------------------------------------------------------------------------------------------
constexpr int BNCH = 3 * 2 * C / FKC;   // 12 chunks
struct WnBwdArgs {
  const uint4* A1;          // layer i: Win^T image, M = 256, KG = 96 (tap-major)
  const uint4* A2;          // layer i-1: Wrs^T image, M = 256, KG = 32 (res rows | skip rows)
  const bf16_t* dpre_i; long dpre_bs;      // [B][HALO + Lr + HALO][512]
  const bf16_t* dh_next;                   // dh_{i+1} [B][Lr][256]; null for the last layer
  bf16_t* dh_out;                          // dh_i
  const bf16_t* dskip;                     // [B][Lr][256]
  const bf16_t* ts;                        // layer i-1 [B][Lr][512]
  bf16_t* dpre_out;                        // dpre_{i-1}
  int L, Lr, B, d, ntiles;
  unsigned long long* stamps;   // STAMP builds only
};

template <int TN, bool STAMP>
__global__ __launch_bounds__(512) void k_wn_bwd(WnBwdArgs p) {
  constexpr int NCB = TN / 32;
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  bf16_t* const ldh = lds + 2 * TN * FLDB;     // [TN][FLDA]: dh_{i+1} -> dh_i
  bf16_t* const lts = ldh + TN * FLDA;         // [TN][FLDT]: tanh | sigmoid -> dpre_{i-1}
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, kh = lane >> 5;
  const int per = (p.ntiles + 7) >> 3;
  const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (tile >= p.ntiles) return;
  const int ncol = (p.L + TN - 1) / TN;
  const int b = tile / ncol, n0 = (tile - b * ncol) * TN;
  const int nrows = min(TN, p.L - n0);
  int stamp_i = 0;
  auto stamp = [&]() {
    if constexpr (STAMP) {
      if (tid == 0) p.stamps[(size_t)tile * 24 + stamp_i] = wall_clock64();
      ++stamp_i;
    }
  };
  stamp();
  constexpr int KG1 = 3 * 2 * C / 16, KG2 = 2 * C / 16;
  const uint4* ap = p.A1 + (size_t)w * KG1 * 64 + lane;
  const size_t srow0 = ((size_t)b * p.Lr + n0) * C;

  f32x16 acc[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;

  auto stage_load = [&](int c, uint4 (&stg)[NCB]) {      // pieces tid + 512 j of the chunk's TN rows x 16 sixteen-byte pieces
    c = min(c, BNCH - 1);
#pragma unroll
    for (int j = 0; j < NCB; ++j) {
      const int e = tid + 512 * j;
      stg[j] = nt_load16(p.dpre_i + (size_t)b * p.dpre_bs + (size_t)(HALO - ((c >> 2) - 1) * p.d + n0 + (e >> 4)) * (2 * C) + (c & 3) * FKC + (e & 15) * 8);
    }
  };
  auto stage_write = [&](int buf, const uint4 (&stg)[NCB]) {
#pragma unroll
    for (int j = 0; j < NCB; ++j) {
      const int e = tid + 512 * j;
      *reinterpret_cast<uint4*>(lds + buf * (TN * FLDB) + (e >> 4) * FLDB + (e & 15) * 8) = stg[j];
    }
  };
  uint4 st[2][NCB], ar[8];
  stage_load(0, st[0]);
#pragma unroll
  for (int s = 0; s < 8; ++s) ar[s] = ap[(size_t)s * 64];
  {
    // whole rows of dh_{i+1} and of layer i-1's tanh | sigmoid (HBM reads: written a forward pass ago) into their LDS tiles, here in the
    // prologue: 4.7 - 6.5 us before the first MFMA instead of 1.6.  Everything else measured worse: requested inside the chunk loop,
    // every weight fragment requested after them waits for them (vmcnt retires in order: + 3 us per tile mid-loop, + 9 us from the last
    // chunk); brought in by a NINTH wave as LDS-DMA spans spread over the loop (no request of its own to hold back), the chunks
    // that run next to its requests take 2.0 - 2.2 us instead of 1.0 -- the CU's one vector-memory path serves the waves' requests in
    // order too (step 17.59 -> 17.88 ms at batch 12)
    const bf16_t* dn = p.dh_next ? p.dh_next : p.dskip;
    uint4 pd[2 * NCB], pt[4 * NCB];
#pragma unroll
    for (int j = 0; j < 2 * NCB; ++j) pd[j] = *reinterpret_cast<const uint4*>(dn + srow0 + (size_t)(tid + 512 * j) * 8);
#pragma unroll
    for (int j = 0; j < 4 * NCB; ++j) pt[j] = *reinterpret_cast<const uint4*>(p.ts + 2 * srow0 + (size_t)(tid + 512 * j) * 8);
    stage_load(1, st[1]);
    stage_write(0, st[0]);
    stage_load(2, st[0]);
#pragma unroll
    for (int j = 0; j < 2 * NCB; ++j) {
      const int e = tid + 512 * j;
      *reinterpret_cast<uint4*>(ldh + (e >> 5) * FLDA + (e & 31) * 8) = p.dh_next ? pd[j] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4 * NCB; ++j) {
      const int e = tid + 512 * j;
      *reinterpret_cast<uint4*>(lts + (e >> 6) * FLDT + (e & 63) * 8) = pt[j];
    }
  }
  __syncthreads();
  stamp();
#pragma unroll
  for (int c = 0; c < BNCH; ++c) {
    stage_write((c + 1) & 1, st[(c + 1) & 1]);
    stage_load(c + 3, st[(c + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    const bf16_t* lb = lds + (c & 1) * (TN * FLDB) + li * FLDB + 8 * kh;
    const size_t gnext = (size_t)(min(c + 1, BNCH - 1) * 8) * 64;
    uint4 bc[NCB], bn[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) bc[cb] = *reinterpret_cast<const uint4*>(lb + cb * 32 * FLDB);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) bn[cb] = s < 7 ? *reinterpret_cast<const uint4*>(lb + cb * 32 * FLDB + 16 * (s + 1)) : bc[cb];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[cb] = mfma_bf16(ar[s], bc[cb], acc[cb]);
      ar[s] = ap[gnext + (size_t)s * 64];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) bc[cb] = bn[cb];
    }
    __syncthreads();
    stamp();
  }
  // second GEMM's image (first half of its reduction) and the dskip rows, requested ahead of the first epilogue
  const uint4* rp = p.A2 + (size_t)w * KG2 * 64 + lane;
  uint4 a2[16], dsk[2 * NCB];
#pragma unroll
  for (int g = 0; g < 16; ++g) a2[g] = rp[(size_t)g * 64];
#pragma unroll
  for (int j = 0; j < 2 * NCB; ++j) dsk[j] = *reinterpret_cast<const uint4*>(p.dskip + srow0 + (size_t)(tid + 512 * j) * 8);
  // dh_i = dh_{i+1} + v, in place: lane (li, kh) owns rows 32 w + 8 q + 4 kh + {0..3} at positions 32 cb + li
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      uint2* cell = reinterpret_cast<uint2*>(ldh + (cb * 32 + li) * FLDA + 32 * w + 8 * q + 4 * kh);
      const uint2 dn = *cell;
      *cell = make_uint2(pack2(acc[cb][4 * q] + lo2f(dn.x), acc[cb][4 * q + 1] + hi2f(dn.x)),
                         pack2(acc[cb][4 * q + 2] + lo2f(dn.y), acc[cb][4 * q + 3] + hi2f(dn.y)));
    }
  bf16_t* const lsk = lds;                     // [TN][FLDA] dskip rows, over the staging buffers (their readers are through the loop's last barrier)
#pragma unroll
  for (int j = 0; j < 2 * NCB; ++j) {
    const int e = tid + 512 * j;
    *reinterpret_cast<uint4*>(lsk + (e >> 5) * FLDA + (e & 31) * 8) = dsk[j];
  }
  __syncthreads();
  stamp();
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
#pragma unroll
  for (int g = 0; g < 32; ++g) {
    const bf16_t* lt = (g < 16 ? ldh : lsk) + li * FLDA + 8 * kh + 16 * (g & 15);
    uint4 bf[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) bf[cb] = *reinterpret_cast<const uint4*>(lt + cb * 32 * FLDA);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[cb] = mfma_bf16(a2[g & 15], bf[cb], acc[cb]);
    if (g < 16) a2[g] = rp[(size_t)(g + 16) * 64];
  }
  stamp();
  // dh_i leaves as whole rows (its tile is complete since the barrier above)
#pragma unroll
  for (int j = 0; j < 2 * NCB; ++j) {
    const int e = tid + 512 * j, row = e >> 5;
    if (row < nrows) *reinterpret_cast<uint4*>(p.dh_out + srow0 + (size_t)e * 8) = *reinterpret_cast<const uint4*>(ldh + row * FLDA + (e & 31) * 8);
  }
  // gate backward in place: tanh | sigmoid cells of this lane -> d tanh-pre | d sigmoid-pre
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      uint2* cT = reinterpret_cast<uint2*>(lts + (cb * 32 + li) * FLDT + 32 * w + 8 * q + 4 * kh);
      uint2* cS = reinterpret_cast<uint2*>(lts + (cb * 32 + li) * FLDT + C + 32 * w + 8 * q + 4 * kh);
      const uint2 Tp = *cT, Sp = *cS;
      const float T[4] = {lo2f(Tp.x), hi2f(Tp.x), lo2f(Tp.y), hi2f(Tp.y)}, S[4] = {lo2f(Sp.x), hi2f(Sp.x), lo2f(Sp.y), hi2f(Sp.y)};
      float dt[4], ds[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) gate_bwd(acc[cb][4 * q + t], T[t], S[t], dt[t], ds[t]);
      *cT = make_uint2(pack2(dt[0], dt[1]), pack2(dt[2], dt[3]));
      *cS = make_uint2(pack2(ds[0], ds[1]), pack2(ds[2], ds[3]));
    }
  __syncthreads();
  stamp();
  bf16_t* gp = p.dpre_out + (size_t)b * p.dpre_bs + (size_t)(HALO + n0) * (2 * C);
#pragma unroll
  for (int j = 0; j < 4 * NCB; ++j) {
    const int e = tid + 512 * j, row = e >> 6;
    if (row < nrows) *reinterpret_cast<uint4*>(gp + (size_t)e * 8) = *reinterpret_cast<const uint4*>(lts + row * FLDT + (e & 63) * 8);
  }
  if constexpr (STAMP) { __builtin_amdgcn_s_waitcnt(0); stamp(); }
}

template <int TN>
int wn_bwd_launch_t(WnBwdArgs& a, hipStream_t s) {
  constexpr size_t ldsb = ((size_t)2 * TN * FLDB + (size_t)TN * FLDA + (size_t)TN * FLDT) * 2;   // 135 168 B (64 positions) / 67 584 B (32)
  a.ntiles = ((a.L + TN - 1) / TN) * a.B;
  const int per = (a.ntiles + 7) / 8;
  {
    static std::atomic<unsigned long long> attr_devices{0};
    int dev = 0;
    FACPPG_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 64 || !((attr_devices.load(std::memory_order_relaxed) >> dev) & 1ull)) {
      FACPPG_HIP_CHECK(hipFuncSetAttribute((const void*)k_wn_bwd<TN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
      FACPPG_HIP_CHECK(hipFuncSetAttribute((const void*)k_wn_bwd<TN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
      if (dev < 64) attr_devices.fetch_or(1ull << dev, std::memory_order_relaxed);
    }
  }
  if (const char* path = stamps_path("FACPPG_WN_BWD_STAMPS", s)) {
    unsigned long long* d = nullptr;
    FACPPG_HIP_CHECK(hipMalloc(&d, (size_t)a.ntiles * 24 * 8));
    FACPPG_HIP_CHECK(hipMemsetAsync(d, 0, (size_t)a.ntiles * 24 * 8, s));
    a.stamps = d;
    k_wn_bwd<TN, true><<<dim3(8 * per), 512, ldsb, s>>>(a);
    const int rc = dump_stamps(path, "bwd", d, a.ntiles, 24, a.d, s);
    FACPPG_HIP_CHECK(hipFree(d));
    return rc;
  }
  k_wn_bwd<TN, false><<<dim3(8 * per), 512, ldsb, s>>>(a);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}
int wn_bwd_launch(WnBwdArgs& a, hipStream_t s) {
  return tile_positions(a.B, a.L) == 64 ? wn_bwd_launch_t<64>(a, s) : wn_bwd_launch_t<32>(a, s);
}
// FACPPG_TRAIN_FUSED_BWD=1 / 0 forces either path.  The backward pair is worth one launch much earlier than the forward pair (its
// tiles stream 1 MB of weights, not 1.7, and the two launches it replaces are the more latency-bound ones): batch 3, whole step,
// 10.8 ms two launches / 10.5 fused with 64-position tiles / 10.2 with 32.
bool fused_bwd_enabled(int B, int L) {
  if (const char* e = getenv("FACPPG_TRAIN_FUSED_BWD")) return e[0] != '0';
  return (long)((L + 31) / 32) * B >= 40;
}

// ------------------------------------------------------------------------------------------------------------
// k_dspect: the conditioning gradient of one flow, dspect[b][n][j] (+)= sum_i sum_o Wcond_i[o][j] * dpre_i[b][n][o]
// ([640 x nl*512] x positions), fp32 out.  Same program shape as k_wn_bwd's first GEMM: a workgroup owns 64 positions and HALF of
// the 640 output channels -- 10 waves x 32 rows, A fragments of the `condt` image from L2 through a register ring one
// 128-entry chunk ahead, the dpre rows two chunks ahead through LDS -- and the fp32 tile leaves through LDS as whole rows
// (1 280 contiguous bytes per position).  Replaces k_bgemm<EP_ACC_F32> here (128 x 128 tiles, 4 waves, lane-per-position
// float4 read-modify-writes): 136 -> see DESIGN.md us per flow at batch 12.  Same K order, same bits.
// ------------------------------------------------------------------------------------------------------------
constexpr int DLDO = NCOND / 2 + 4;     // fp32 tile pitch
constexpr size_t kDspectLds = (size_t)64 * DLDO * 4;   // 82 944 B (staging 34 816 B first)
struct DspectArgs {
  const uint4* A;                  // condt image: M = 640, KG = nl * 32
  const bf16_t* dpre; long dpre_one, dpre_bs;   // layer i at dpre + i * dpre_one (elements); [B][HALO + Lr + HALO][512]
  float* out;                      // [B][Lr][640]
  int nl, L, Lr, B, accumulate, ntiles;
};
__global__ __launch_bounds__(640) void k_dspect(DspectArgs p) {
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  constexpr int TN = 64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, kh = lane >> 5;
  // (row half, position tile): the two halves of a position tile are neighbours on one XCD (they read the same dpre rows)
  const int per = (p.ntiles + 7) >> 3;
  const int t2 = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (t2 >= p.ntiles) return;
  const int mt = t2 & 1, tile = t2 >> 1;
  const int ncol = (p.L + TN - 1) / TN;
  const int b = tile / ncol, n0 = (tile - b * ncol) * TN;
  const int nrows = min(TN, p.L - n0);
  const int KG = p.nl * (2 * C / 16), nch = p.nl * (2 * C / FKC);
  const uint4* ap = p.A + (size_t)(10 * mt + w) * KG * 64 + lane;
  const int st_t = tid & 511;       // waves 8 and 9 repeat the pieces of waves 0 and 1 (same values to the same cells): no branch round a request

  f32x16 acc[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
  auto stage_load = [&](int c, uint4 (&stg)[2]) {
    c = min(c, nch - 1);
    const bf16_t* base = p.dpre + (size_t)(c >> 2) * p.dpre_one + (size_t)b * p.dpre_bs + (size_t)(HALO + n0) * (2 * C) + (c & 3) * FKC;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = st_t + 512 * j;
      stg[j] = nt_load16(base + (size_t)(e >> 4) * (2 * C) + (e & 15) * 8);
    }
  };
  auto stage_write = [&](int buf, const uint4 (&stg)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = st_t + 512 * j;
      *reinterpret_cast<uint4*>(lds + buf * (TN * FLDB) + (e >> 4) * FLDB + (e & 15) * 8) = stg[j];
    }
  };
  uint4 st[2][2], ar[8];
  stage_load(0, st[0]);
#pragma unroll
  for (int s = 0; s < 8; ++s) ar[s] = ap[(size_t)s * 64];
  stage_load(1, st[1]);
  stage_write(0, st[0]);
  stage_load(2, st[0]);
  __syncthreads();
  auto chunk = [&](int c, uint4 (&stn)[2]) {
    stage_write((c + 1) & 1, stn);
    stage_load(c + 3, stn);
    __builtin_amdgcn_sched_barrier(0);
    const bf16_t* lb = lds + (c & 1) * (TN * FLDB) + li * FLDB + 8 * kh;
    const size_t gnext = (size_t)(min(c + 1, nch - 1) * 8) * 64;
    uint4 bc[2], bn[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) bc[cb] = *reinterpret_cast<const uint4*>(lb + cb * 32 * FLDB);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) bn[cb] = s < 7 ? *reinterpret_cast<const uint4*>(lb + cb * 32 * FLDB + 16 * (s + 1)) : bc[cb];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc[cb] = mfma_bf16(ar[s], bc[cb], acc[cb]);
      ar[s] = ap[gnext + (size_t)s * 64];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) bc[cb] = bn[cb];
    }
    __syncthreads();
  };
  for (int c = 0; c < nch; c += 2) {     // (nch = 4 nl is even; the register ring of the staged rows is indexed statically)
    chunk(c, st[1]);
    chunk(c + 1, st[0]);
  }
  // fp32 tile [position][row of this half] over the staging buffers, then whole rows out
  float* lo = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
      *reinterpret_cast<float4*>(lo + (cb * 32 + li) * DLDO + 32 * w + 8 * q + 4 * kh) =
          make_float4(acc[cb][4 * q], acc[cb][4 * q + 1], acc[cb][4 * q + 2], acc[cb][4 * q + 3]);
  __syncthreads();
  float* go = p.out + ((size_t)b * p.Lr + n0) * NCOND + (NCOND / 2) * mt;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int e = tid + 640 * j, row = e / (NCOND / 8), c4 = (e - row * (NCOND / 8)) * 4;
    if (row >= nrows) continue;
    float4* dst = reinterpret_cast<float4*>(go + (size_t)row * NCOND + c4);
    const float4 v = *reinterpret_cast<const float4*>(lo + row * DLDO + c4);
    float4 o = p.accumulate ? *dst : make_float4(0.f, 0.f, 0.f, 0.f);
    o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
    *dst = o;
  }
}
int dspect_launch(DspectArgs& a, hipStream_t s) {
  a.ntiles = 2 * ((a.L + 63) / 64) * a.B;
  const int per = (a.ntiles + 7) / 8;
  {
    static std::atomic<unsigned long long> attr_devices{0};
    int dev = 0;
    FACPPG_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 64 || !((attr_devices.load(std::memory_order_relaxed) >> dev) & 1ull)) {
      FACPPG_HIP_CHECK(hipFuncSetAttribute((const void*)k_dspect, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDspectLds));
      if (dev < 64) attr_devices.fetch_or(1ull << dev, std::memory_order_relaxed);
    }
  }
  k_dspect<<<dim3(8 * per), 640, kDspectLds, s>>>(a);
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
  size_t pstride;    // floats per (problem, split) partial = max M * max K of the batch
  int xcd_map, tiles_k, tiles_m, ngroups;   // XCD-aware workgroup order (k_wgrad)
  unsigned long long* stamps;               // debugging aid (FACPPG_WGRAD_STAMPS): [workgroup < 64][24] wall_clock64 values of thread 0
  int nsplit;        // the B * ceil(L/64) position chunks are dealt to nsplit workgroups per output tile ...
  float* part;       // ... which leave partial sums [prob][split][M][K] here (nsplit > 1); k_wgrad_reduce adds them in order
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

template <bool STAMP>
__global__ __launch_bounds__(256) void k_wgrad(WgradArgs wa) {
  extern __shared__ __attribute__((aligned(16))) bf16_t lds_dyn[];      // [buffer][A | B][channel][position], double-buffered: one barrier per chunk
  bf16_t (*lds)[2][128 * LDP] = reinterpret_cast<bf16_t (*)[2][128 * LDP]>(lds_dyn);
  // The output tiles of one (problem, split) read the SAME position range of dY and X: they are dealt to ONE XCD back to back
  // (workgroup lin runs on XCD lin % 8), so that XCD's L2 fetches the rows once; groups go round the XCDs.
  int pi, split, m0, k0;
  if (wa.xcd_map) {
    const int tg = wa.tiles_k * wa.tiles_m, x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int grp = (j / tg) * 8 + x, t = j % tg;
    if (grp >= wa.ngroups) return;
    pi = grp / wa.nsplit; split = grp - pi * wa.nsplit;
    k0 = (t % wa.tiles_k) * 128; m0 = (t / wa.tiles_k) * 128;
  } else {
    pi = blockIdx.z / wa.nsplit; split = blockIdx.z - pi * wa.nsplit;
    m0 = blockIdx.y * 128; k0 = blockIdx.x * 128;
  }
  const WgradProb& p = wa.prob[pi];
  if (m0 >= p.M || k0 >= p.K) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, kh = lane >> 5;
  const int wm = w >> 1, wk = w & 1;
  // staging role: threads 0..127 transpose dY blocks, 128..255 X blocks; block = (pb: 8 positions, cb: 8 channels)
  // LDS image [channel][position], row pitch 144 B.  Position block pb of a channel row sits at column block (pb + f) & 7 with
  // f = (row >> 5) & 3: constant inside every 32-row fragment block, so the ds_read_b128 lane groups stay conflict-free, while the
  // 8 lanes of a store group take channel blocks 0,4,8,12,1,5,9,13 (or 2,6,.. / 3,7,..) -- four different f per bank half.
  // Measured: channel blocks in lane order, no swizzle: SQ_LDS_BANK_CONFLICT = 50 % of the LDS-active cycles, 202 us; position
  // blocks fastest (conflict-free but 16-byte pieces of 8 rows per 8 lanes on the global side): 168 us.
  const int isx = tid >> 7, blk = tid & 127, pb = blk >> 4, l16 = blk & 15, cb8 = 4 * (l16 & 3) + (l16 >> 2);
  const int nlc = (wa.L + 63) / 64, nall = wa.B * nlc;
  const int c_lo = (int)((long)nall * split / wa.nsplit), c_hi = (int)((long)nall * (split + 1) / wa.nsplit);
  const bf16_t* src;
  long sbs; int sld, srow0, ch0;
  if (!isx) {
    const bool second = p.dy1 && m0 >= p.msplit;
    src = second ? p.dy1 : p.dy0; sbs = p.dy_bs; sld = p.ldy; srow0 = p.dy_row0;
    ch0 = (second ? m0 - p.msplit : m0) + 8 * cb8;
  } else {
    src = p.x; sbs = p.x_bs; sld = p.ldx; srow0 = p.x_row0; ch0 = k0 + 8 * cb8;
  }
  auto stage_load = [&](int c, uint4 (&stg)[8]) {
    const int b = c / nlc, n = (c - b * nlc) * 64 + 8 * pb;
    const bf16_t* s0 = src + (size_t)b * sbs + (size_t)(srow0 + n) * sld + ch0;
#pragma unroll
    for (int i = 0; i < 8; ++i) stg[i] = *reinterpret_cast<const uint4*>(s0 + (size_t)i * sld);   // (M, K multiples of 128: no guard, no branch)
  };
  auto stage_write = [&](int buf, const uint4 (&stg)[8]) {
    uint4 t[8];
    transpose8x8(stg, t);
    bf16_t* d = &lds[buf][isx][(8 * cb8) * LDP + 8 * ((pb + (cb8 >> 2)) & 7)];
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
  const int la_row = (64 * wm + li) * LDP, lb_row = (64 * wk + li) * LDP;
  // the rows of a chunk come from HBM or another tile's L2 lines and a chunk needs ALL of its 8 requests per thread: requested
  // two chunks ahead the loop ran at the latency of its slowest request (0.88 us per chunk of 0.21 us of MFMA work, in-kernel
  // stamps); the register ring is WG_PD chunks deep
  constexpr int WG_PD = 2;     // 4 measured: the first workgroups of a launch run their chunks in 0.88 us instead of 1.2, the launch as a whole 164 us instead of 124 (more lines in flight evict more of what the neighbouring tiles re-read)
  uint4 st[WG_PD][8];
  if (c_lo >= c_hi) return;          // (uniform per workgroup)
  const int c_last = c_hi - 1;
  unsigned long long* stp = (STAMP && blockIdx.x < 64 && tid == 0) ? wa.stamps + (size_t)blockIdx.x * 24 : nullptr;   // (STAMP builds only: a store
  int stn = 0;                                                                                                            //  behind a branch in the loop costs the prefetch)
  if (STAMP && stp) stp[stn++] = wall_clock64();
  stage_load(c_lo, st[0]);
#pragma unroll
  for (int u = 1; u < WG_PD; ++u) stage_load(min(c_lo + u, c_last), st[u]);
  stage_write(0, st[0]);
  __syncthreads();
  // iteration c (ring slot k = (c - c_lo) % WG_PD, free since chunk c went to LDS an iteration ago): chunk c + WG_PD requested into
  // slot k, MFMAs of chunk c from LDS, then chunk c+1 (slot k+1, requested WG_PD - 1 iterations ago) replaces chunk c-1 in LDS
  auto iter = [&](int c, uint4 (&s_new)[8], uint4 (&s_nxt)[8]) {
    stage_load(min(c + WG_PD, c_last), s_new);   // unconditional (see k_bgemm): the last iterations re-read the last chunk
    __builtin_amdgcn_sched_barrier(0);   // keep the requests where they are
    const int buf = (c - c_lo) & 1;
    const bf16_t* la = &lds[buf][0][la_row];
    const bf16_t* lb = &lds[buf][1][lb_row];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      uint4 av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        av[i] = *reinterpret_cast<const uint4*>(la + i * 32 * LDP + 8 * ((2 * s + kh + 2 * wm + i) & 7));
        bv[i] = *reinterpret_cast<const uint4*>(lb + i * 32 * LDP + 8 * ((2 * s + kh + 2 * wk + i) & 7));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_bf16(av[i], bv[j], acc[i][j]);
    }
    // chunk c+1 goes to the OTHER buffer (last read in iteration c-1, which every wave left through the barrier below)
    stage_write(buf ^ 1, s_nxt);
    __syncthreads();
    if constexpr (STAMP) { if (stp && stn < 22) stp[stn++] = wall_clock64(); }
  };
  int c = c_lo;
  for (; c + WG_PD - 1 < c_hi; c += WG_PD) {
#pragma unroll
    for (int u = 0; u < WG_PD; ++u) iter(c + u, st[u], st[(u + 1) % WG_PD]);
  }
#pragma unroll
  for (int u = 0; u < WG_PD - 1; ++u)
    if (c + u < c_hi) iter(c + u, st[u], st[(u + 1) % WG_PD]);
  if constexpr (STAMP) { if (stp) { stp[22] = wall_clock64(); stp[23] = (unsigned long long)(c_hi - c_lo); } }
  float* part = wa.nsplit > 1 ? wa.part + ((size_t)pi * wa.nsplit + split) * wa.pstride : nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + 64 * wk + 32 * j + li;
      if (k >= p.K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 64 * wm + 32 * i + 8 * (r >> 2) + (r & 3) + 4 * kh;
        if (m >= p.M) continue;
        if (part) part[(size_t)m * p.K + k] = acc[i][j][r];
        else p.out[m * p.o_sm + k * p.o_sk] = acc[i][j][r];
      }
    }
}

// k_wgrad2: the same products on 256 x 256 output tiles -- 8 waves as 2 x 4, 128 x 64 each -- for the launches with enough positions
// per workgroup.  Why: a 128 x 128 tile needs 32 KB of operand rows per 64 positions, i.e. per 512 MFMA cycles of its four waves;
// k_wgrad's workgroups were measured (in-kernel stamps, tools/wgrad_phase_probe.py) at 0.9 - 1.2 us per chunk = 27 - 36 GB/s of
// rows per CU, 9 TB/s over the chip, with the matrix pipe 23 % busy: the rows' delivery, not the MFMAs, sets the pace, and at
// that tile size even the L1 fill rate (64 B/clk) caps the pipe at ~50 %.  A 256 x 256 tile needs half the bytes per FLOP.  Same
// staging scheme (8 x 8 register transposes into [channel][position] images, column-block swizzle), same reduction order inside a
// workgroup; the split of the positions over workgroups differs (ordered reduce, deterministic).
__global__ __launch_bounds__(512) void k_wgrad2(WgradArgs wa) {
  extern __shared__ __attribute__((aligned(16))) bf16_t lds_dyn[];      // [buffer][A | B][256 channels][LDP]
  bf16_t (*lds)[2][256 * LDP] = reinterpret_cast<bf16_t (*)[2][256 * LDP]>(lds_dyn);
  const int tg = wa.tiles_k * wa.tiles_m, x = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int grp = (jx / tg) * 8 + x, t = jx % tg;
  if (grp >= wa.ngroups) return;
  const int pi = grp / wa.nsplit, split = grp - pi * wa.nsplit;
  const int k0 = (t % wa.tiles_k) * 256, m0 = (t / wa.tiles_k) * 256;
  const WgradProb& p = wa.prob[pi];
  if (m0 >= p.M || k0 >= p.K) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, kh = lane >> 5;
  const int wm = w >> 2, wk = w & 3;
  // staging role: threads 0..255 transpose dY blocks, 256..511 X blocks; block = (pb: 8 positions, cb8: 8 channels of 256)
  const int isx = tid >> 8, blk = tid & 255, pb = blk >> 5, l32 = blk & 31, cb8 = 8 * (l32 & 3) + (l32 >> 2);
  const int nlc = (wa.L + 63) / 64, nall = wa.B * nlc;
  const int c_lo = (int)((long)nall * split / wa.nsplit), c_hi = (int)((long)nall * (split + 1) / wa.nsplit);
  const bf16_t* src;
  long sbs; int sld, srow0, ch0;
  if (!isx) {
    const bool second = p.dy1 && m0 >= p.msplit;
    src = second ? p.dy1 : p.dy0; sbs = p.dy_bs; sld = p.ldy; srow0 = p.dy_row0;
    ch0 = min((second ? m0 - p.msplit : m0) + 8 * cb8, (second ? p.M - p.msplit : (p.dy1 ? p.msplit : p.M)) - 8);
  } else {
    src = p.x; sbs = p.x_bs; sld = p.ldx; srow0 = p.x_row0;
    ch0 = min(k0 + 8 * cb8, p.K - 8);     // a tile that reaches past K (640 = 2.5 tiles) re-reads the last channels; those columns are never stored
  }
  // the operand side (dY / X) is the same for a whole wave: its base, strides and the chunk's row are kept in SGPRs (readfirstlane),
  // a request is scalar base + one loop-invariant 32-bit lane offset -- per-lane 64-bit row addresses cost the registers this
  // kernel does not have (acc 128 + two staged chunks 64 + fragments 24)
  const unsigned long long src_u = (unsigned long long)src;
  typedef __attribute__((address_space(1))) const char* gchar_p;          // (rebuilt from integers: say that it is GLOBAL memory, or the loads become flat_load)
  typedef __attribute__((address_space(1))) const u32x4* gu32x4_p;
  const gchar_p src_s = (gchar_p)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(src_u >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)src_u));
  const long sbs_s = ((long)__builtin_amdgcn_readfirstlane((int)(sbs >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sbs);
  const int sld_s = __builtin_amdgcn_readfirstlane(sld), srow0_s = __builtin_amdgcn_readfirstlane(srow0);
  const unsigned voff = (unsigned)((8 * pb * sld + ch0) * 2);
  auto stage_load = [&](int c, uint4 (&stg)[8]) {
    const int b = c / nlc, n = (c - b * nlc) * 64;
    const gchar_p s0 = src_s + ((size_t)b * sbs_s + (size_t)(srow0_s + n) * sld_s) * 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const u32x4 v = *(gu32x4_p)(s0 + (size_t)i * sld_s * 2 + voff);
      stg[i] = make_uint4(v.x, v.y, v.z, v.w);
    }
  };
  auto stage_write = [&](int buf, const uint4 (&stg)[8]) {
    uint4 tt[8];
    transpose8x8(stg, tt);
    bf16_t* d = &lds[buf][isx][(8 * cb8) * LDP + 8 * ((pb + (cb8 >> 2)) & 7)];
#pragma unroll
    for (int c = 0; c < 8; ++c) *reinterpret_cast<uint4*>(d + c * LDP) = tt[c];
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const int la_row = (128 * wm + li) * LDP, lb_row = (64 * wk + li) * LDP;
  // ONE staged chunk in registers (acc 128 + fragments 24 leave no room for two: a second set spilled LDS addresses into the
  // loop, each reload behind an s_waitcnt vmcnt(0)): chunk c+1 goes to the free LDS buffer at the TOP of iteration c (its last
  // readers left through the barrier), then chunk c+2 is requested into the same registers and has the iteration to arrive
  uint4 st[8];
  if (c_lo >= c_hi) return;
  const int c_last = c_hi - 1;
  stage_load(c_lo, st);
  stage_write(0, st);
  stage_load(min(c_lo + 1, c_last), st);
  __syncthreads();
  for (int c = c_lo; c < c_hi; ++c) {
    const int buf = (c - c_lo) & 1;
    stage_write(buf ^ 1, st);
    stage_load(min(c + 2, c_last), st);
    __builtin_amdgcn_sched_barrier(0);
    const bf16_t* la = &lds[buf][0][la_row];
    const bf16_t* lb = &lds[buf][1][lb_row];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      uint4 av[4], bv[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const uint4*>(la + i * 32 * LDP + 8 * ((2 * s + kh + 4 * wm + i) & 7));   // column block (pb + (row >> 5)) & 7, as written
#pragma unroll
      for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const uint4*>(lb + j * 32 * LDP + 8 * ((2 * s + kh + 2 * wk + j) & 7));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_bf16(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
  float* part = wa.nsplit > 1 ? wa.part + ((size_t)pi * wa.nsplit + split) * wa.pstride : nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + 64 * wk + 32 * j + li;
      if (k >= p.K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 128 * wm + 32 * i + 8 * (r >> 2) + (r & 3) + 4 * kh;
        if (m >= p.M) continue;
        if (part) part[(size_t)m * p.K + k] = acc[i][j][r];
        else p.out[m * p.o_sm + k * p.o_sk] = acc[i][j][r];
      }
    }
}

// out[m][k] = sum over the splits, in split order (bit-reproducible)
__global__ void k_wgrad_reduce(WgradArgs wa) {
  const WgradProb& p = wa.prob[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.M * p.K) return;
  const float* part = wa.part + (size_t)blockIdx.y * wa.nsplit * wa.pstride + i;
  float v = 0.0f;
  for (int s = 0; s < wa.nsplit; ++s) v += part[(size_t)s * wa.pstride];
  const int m = i / p.K, k = i - m * p.K;
  p.out[m * p.o_sm + k * p.o_sk] = v;
}

// The ordered sums of SEVERAL k_wgrad / k_wgrad2 launches' partials in one launch (a flow's tap, conditioning and res/skip products
// each leave theirs in a region of their own): same sums in the same order as k_wgrad_reduce per launch.
constexpr int MAXRED = 48;
struct ReduceProb { const float* part; float* out; size_t pstride; long o_sm, o_sk; int nsplit, M, K; };
struct ReduceSet {
  ReduceProb p[MAXRED];
  int n = 0, max_mk = 0;
  void add(const WgradArgs& wa, int nprob) {
    for (int i = 0; i < nprob && n < MAXRED; ++i) {
      const WgradProb& q = wa.prob[i];
      p[n++] = ReduceProb{wa.part + (size_t)i * wa.nsplit * wa.pstride, q.out, wa.pstride, q.o_sm, q.o_sk, wa.nsplit, q.M, q.K};
      max_mk = std::max(max_mk, q.M * q.K);
    }
  }
};
// blk0[i]: first workgroup of problem i (a workgroup = 1024 consecutive outputs, four per thread: 16-byte loads of the partials)
struct ReduceArgs { ReduceProb p[MAXRED]; int blk0[MAXRED + 1]; int n; };
__global__ __launch_bounds__(256) void k_wgrad_reduce_multi(ReduceArgs ra) {
  int pi = 0;
  while (pi + 1 < ra.n && (int)blockIdx.x >= ra.blk0[pi + 1]) ++pi;
  const ReduceProb& p = ra.p[pi];
  const int i = ((blockIdx.x - ra.blk0[pi]) * 256 + threadIdx.x) * 4;
  if (i >= p.M * p.K) return;
  const float* part = p.part + i;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < p.nsplit; ++s) {
    const float4 x = *reinterpret_cast<const float4*>(part + (size_t)s * p.pstride);
    v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
  }
  const int m = i / p.K, k = i - m * p.K;      // (K is a multiple of 4: the four outputs share the row)
  float* o = p.out + m * p.o_sm + k * p.o_sk;
  o[0] = v.x; o[p.o_sk] = v.y; o[2 * p.o_sk] = v.z; o[3 * p.o_sk] = v.w;
}
int reduce_launch(ReduceSet& rs, hipStream_t s) {
  if (!rs.n) return FACPPG_OK;
  ReduceArgs ra;
  memcpy(ra.p, rs.p, sizeof(ReduceProb) * rs.n);
  int blocks = 0;
  for (int i = 0; i < rs.n; ++i) {
    FACPPG_REQUIRE(rs.p[i].K % 4 == 0 && rs.p[i].pstride % 4 == 0, FACPPG_EINVAL, "weight-gradient reduce: K must be a multiple of 4");
    ra.blk0[i] = blocks;
    blocks += (rs.p[i].M * rs.p[i].K + 1023) / 1024;
  }
  ra.blk0[rs.n] = blocks; ra.n = rs.n;
  k_wgrad_reduce_multi<<<blocks, 256, 0, s>>>(ra);
  FACPPG_HIP_CHECK(hipGetLastError());
  rs.n = 0; rs.max_mk = 0;
  return FACPPG_OK;
}

constexpr int WG_MAXSPLIT = 8;
// launches one batch of problems that share (M, K) tile counts; partial buffer: nprob * nsplit * maxM * maxK floats
// 256 x 256 tiles (k_wgrad2) when they give >= 32 tiles and every workgroup then has >= 24 chunks of positions; FACPPG_WGRAD_TILE=128|256 forces
// the split count of a k_wgrad2 launch and whether the launch is the one to use (shared with facppg_wn_bf16_launch_plan)
bool wgrad2_wanted(int nprob, int maxM, int maxK, int B, int L, size_t part_bytes, int* ns_out = nullptr) {
  const int tiles = ((maxK + 255) / 256) * ((maxM + 255) / 256) * nprob;
  const int nall = B * ((L + 63) / 64);
  int ns = std::max(1, std::min(std::min(256 / std::max(tiles, 1), 8), nall));      // one workgroup per CU (144 KB of LDS each): never more than 256; every split gets a chunk
  while (ns > 1 && (size_t)nprob * ns * maxM * maxK * 4 > part_bytes) --ns;
  if (ns_out) *ns_out = ns;
  const char* e = getenv("FACPPG_WGRAD_TILE");
  return e ? atoi(e) == 256 : (tiles >= 32 && nall / ns >= 24);
}
int wgrad2_launch(WgradArgs& wa, int nprob, int maxM, int maxK, float* part, size_t part_bytes, hipStream_t s, bool* done, ReduceSet* defer = nullptr) {
  *done = false;
  const int tk = (maxK + 255) / 256, tm = (maxM + 255) / 256;
  int ns = 1;
  if (!wgrad2_wanted(nprob, maxM, maxK, wa.B, wa.L, part_bytes, &ns)) return FACPPG_OK;
  wa.nsplit = ns; wa.part = part; wa.pstride = (size_t)maxM * maxK;
  wa.tiles_k = tk; wa.tiles_m = tm; wa.ngroups = nprob * ns; wa.xcd_map = 1;
  constexpr size_t kLds = (size_t)2 * 2 * 256 * LDP * sizeof(bf16_t);   // 147 456 B
  {
    static std::atomic<unsigned long long> attr_devices{0};
    int dev = 0;
    FACPPG_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 64 || !((attr_devices.load(std::memory_order_relaxed) >> dev) & 1ull)) {
      FACPPG_HIP_CHECK(hipFuncSetAttribute((const void*)k_wgrad2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
      if (dev < 64) attr_devices.fetch_or(1ull << dev, std::memory_order_relaxed);
    }
  }
  k_wgrad2<<<dim3(8 * ((wa.ngroups + 7) / 8) * tk * tm), 512, kLds, s>>>(wa);
  if (ns > 1) {
    if (defer) defer->add(wa, nprob);
    else k_wgrad_reduce<<<dim3((maxM * maxK + 255) / 256, nprob), 256, 0, s>>>(wa);
  }
  FACPPG_HIP_CHECK(hipGetLastError());
  *done = true;
  return FACPPG_OK;
}

// defer: the partials' ordered sum is left to the caller's reduce_launch (one launch for several of these)
int wgrad_launch(WgradArgs& wa, int nprob, int maxM, int maxK, float* part, size_t part_bytes, hipStream_t s, ReduceSet* defer = nullptr) {
  FACPPG_REQUIRE(maxM % 128 == 0 && maxK % 128 == 0, FACPPG_EINVAL, "k_wgrad: M and K must be multiples of 128 (got %d, %d)", maxM, maxK);
  FACPPG_REQUIRE(!defer || defer->n + nprob <= MAXRED, FACPPG_EINVAL, "too many deferred weight-gradient problems");
  {
    bool done = false;
    if (int rc = wgrad2_launch(wa, nprob, maxM, maxK, part, part_bytes, s, &done, defer)) return rc;
    if (done) return FACPPG_OK;
  }
  const int tiles = ((maxK + 127) / 128) * ((maxM + 127) / 128) * nprob;
  const int nall = wa.B * ((wa.L + 63) / 64);
  // about ONE workgroup per CU (two fit, 73 KB of LDS each): measured at batch 3 / 12, whole step, aiming at 64 / 128 / 256 / 384 /
  // 512 / 768 / 1536 / 3072 workgroups: 11.48 / 11.23 / 11.09 / 11.18 / 11.31 / 11.53 / 11.99 / 12.93 ms and 24.42 / 23.43 / 22.46 /
  // 22.55 / 22.66 / 23.06 / 23.12 / 24.26 ms -- every further split adds a 64 KB partial tile per workgroup and a longer reduction
  int ns = (256 + tiles - 1) / tiles;
  ns = std::max(1, std::min(std::min(ns, WG_MAXSPLIT), nall));
  while (ns > 1 && (size_t)nprob * ns * maxM * maxK * 4 > part_bytes) --ns;
  wa.nsplit = ns; wa.part = part; wa.pstride = (size_t)maxM * maxK;
  constexpr size_t kWgradLds = (size_t)2 * 2 * 128 * LDP * sizeof(bf16_t);   // 73 728 B
  {
    // > 64 KB of dynamic LDS needs the attribute on EVERY device the kernel runs on (one bit per device; a second thread
    // setting it again is harmless)
    static std::atomic<unsigned long long> attr_devices{0};
    int dev = 0;
    FACPPG_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 64 || !((attr_devices.load(std::memory_order_relaxed) >> dev) & 1ull)) {
      FACPPG_HIP_CHECK(hipFuncSetAttribute((const void*)k_wgrad<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWgradLds));
      FACPPG_HIP_CHECK(hipFuncSetAttribute((const void*)k_wgrad<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWgradLds));
      if (dev < 64) attr_devices.fetch_or(1ull << dev, std::memory_order_relaxed);
    }
  }
  {
    const char* e_map = getenv("FACPPG_WGRAD_NO_XCD_MAP");
    const bool no_map = e_map && e_map[0] == '1';
    wa.tiles_k = (maxK + 127) / 128; wa.tiles_m = (maxM + 127) / 128; wa.ngroups = nprob * ns; wa.xcd_map = no_map ? 0 : 1;
    const char* spath = stamps_path("FACPPG_WGRAD_STAMPS", s);
    if (spath) {
      FACPPG_HIP_CHECK(hipMalloc(&wa.stamps, 64 * 24 * 8));
      FACPPG_HIP_CHECK(hipMemsetAsync(wa.stamps, 0, 64 * 24 * 8, s));
    }
    const dim3 grid = wa.xcd_map ? dim3(8 * ((wa.ngroups + 7) / 8) * wa.tiles_k * wa.tiles_m) : dim3(wa.tiles_k, wa.tiles_m, nprob * ns);
    if (spath) k_wgrad<true><<<grid, 256, kWgradLds, s>>>(wa);
    else k_wgrad<false><<<grid, 256, kWgradLds, s>>>(wa);
    if (spath) {
      unsigned long long h[64 * 24];
      FACPPG_HIP_CHECK(hipMemcpyAsync(h, wa.stamps, sizeof(h), hipMemcpyDeviceToHost, s));
      FACPPG_HIP_CHECK(hipStreamSynchronize(s));
      FACPPG_HIP_CHECK(hipFree(wa.stamps));
      wa.stamps = nullptr;
      if (FILE* f = fopen(spath, "a")) {
        fprintf(f, "launch wgrad nprob %d M %d K %d nsplit %d\n", nprob, maxM, maxK, ns);
        for (int t = 0; t < 64; ++t) {
          for (int j = 0; j < 24; ++j) fprintf(f, "%llu ", j == 23 ? h[t * 24 + j] : (h[t * 24 + j] ? h[t * 24 + j] - h[t * 24] : 0ull));
          fprintf(f, "\n");
        }
        fclose(f);
      }
    }
  }
  if (ns > 1) {
    if (defer) defer->add(wa, nprob);
    else k_wgrad_reduce<<<dim3((maxM * maxK + 255) / 256, nprob), 256, 0, s>>>(wa);
  }
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

// out[m] = sum_{b, n < L} y[b][row0 + n][m]  (bias gradients).  Stage 1: workgroup (channel tile, row slice, problem)
// sums its slice of the B*L rows (4 row lanes x 64 channels per wavefront-row); stage 2 adds the slices in index
// order -- a fixed summation order whatever the grid, so the result is bit-reproducible.
constexpr int CS_SLICES = 256, MAXCS = 16;
struct ColsumProb { const void* y; long bs; int ld, row0, M; float* out; float* out2; };   // out2: optional second copy of the sums
// part [prob][CS_SLICES][1024]; bc: further copies of problem bc_prob's sums (the skip half of every layer's res_skip bias gradient
// is the column sum of the same dskip)
struct ColsumArgs { ColsumProb prob[MAXCS]; int B, L; float* part; float* bc[8]; int nbc, bc_prob; };
template <bool F32>
__global__ __launch_bounds__(256) void k_colsum_part(ColsumArgs ca) {
  // a thread owns TWO adjacent channels (one 4-byte load of bf16, 8 bytes of fp32), the 256 threads cover M / 2 channel
  // pairs x 512 / M row lanes: every row is read as whole contiguous segments
  const ColsumProb& p = ca.prob[blockIdx.z];
  const int npair = p.M / 2, lanes = 256 / npair > 0 ? 256 / npair : 1;       // M <= 512, even
  const int pr = threadIdx.x % npair, rl = threadIdx.x / npair, sl = blockIdx.y;
  __shared__ float red[512];
  const long R = (long)ca.B * ca.L, r0 = R * sl / CS_SLICES, r1 = R * (sl + 1) / CS_SLICES;
  float v0 = 0.0f, v1 = 0.0f;
  if (rl < lanes)
#pragma unroll 4
    for (long r = r0 + rl; r < r1; r += lanes) {
      const int b = (int)(r / ca.L), n = (int)(r - (long)b * ca.L);
      const size_t o = (size_t)b * p.bs + (size_t)(p.row0 + n) * p.ld + 2 * pr;
      if (F32) {
        const float2 x = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(p.y) + o);
        v0 += x.x; v1 += x.y;
      } else {
        const unsigned x = *reinterpret_cast<const unsigned*>(reinterpret_cast<const bf16_t*>(p.y) + o);
        v0 += lo2f(x); v1 += hi2f(x);
      }
    }
  // row lanes meet in LDS in lane order
  float* part = ca.part + ((size_t)blockIdx.z * CS_SLICES + sl) * 1024;
  for (int l = 0; l < lanes; ++l) {
    if (rl == l) {
      if (l == 0) { red[2 * pr] = v0; red[2 * pr + 1] = v1; }
      else { red[2 * pr] += v0; red[2 * pr + 1] += v1; }
    }
    __syncthreads();
  }
  for (int m = threadIdx.x; m < p.M; m += 256) part[m] = red[m];
}
// bf16 rows of M = 256 or 512 channels: a thread owns EIGHT adjacent channels (one 16-byte load) of every `lanes`-th row,
// lanes = 256 / (M / 8) = 8 or 4 -- a quarter / an eighth of the sequential steps of the 4-byte version, which spent 25-32 us
// per launch at batch 12 on a chain of short loads.  Same slices, same partial layout, row lanes met in lane order.
__global__ __launch_bounds__(256) void k_colsum_part8(ColsumArgs ca) {
  const ColsumProb& p = ca.prob[blockIdx.z];
  const int tpr = p.M / 8, lanes = 256 / tpr;                   // threads per row, row lanes
  const int c8 = (threadIdx.x % tpr) * 8, rl = threadIdx.x / tpr, sl = blockIdx.y;
  __shared__ float red[7][512];
  const long R = (long)ca.B * ca.L, r0 = R * sl / CS_SLICES, r1 = R * (sl + 1) / CS_SLICES;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (long r = r0 + rl; r < r1; r += lanes) {
    const int b = (int)(r / ca.L), n = (int)(r - (long)b * ca.L);
    const uint4 x = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.y) + (size_t)b * p.bs + (size_t)(p.row0 + n) * p.ld + c8);
    v[0] += lo2f(x.x); v[1] += hi2f(x.x); v[2] += lo2f(x.y); v[3] += hi2f(x.y);
    v[4] += lo2f(x.z); v[5] += hi2f(x.z); v[6] += lo2f(x.w); v[7] += hi2f(x.w);
  }
  if (rl) {
#pragma unroll
    for (int t = 0; t < 8; ++t) red[rl - 1][c8 + t] = v[t];
  }
  __syncthreads();
  if (rl) return;
  for (int l = 0; l + 1 < lanes; ++l)
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] += red[l][c8 + t];
  float* part = ca.part + ((size_t)blockIdx.z * CS_SLICES + sl) * 1024 + c8;
  *reinterpret_cast<float4*>(part) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(part + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
// group > 1: out[m / group] also sums `group` adjacent channels (the 8 regrouped samples of one mel channel).
// 32 outputs x 8 slice groups per workgroup: a thread adds its 32 slices on four interleaved chains (the loads overlap),
// the eight groups meet in LDS in group order -- a fixed order whatever the grid.  (One thread per output walking all 256
// slices cost 13.7 us per launch, 25 launches per step.)
__global__ __launch_bounds__(256) void k_colsum_sum(ColsumArgs ca, int group) {
  const ColsumProb& p = ca.prob[blockIdx.y];
  __shared__ float red[8][32];
  const int o = threadIdx.x & 31, q = threadIdx.x >> 5, mo = blockIdx.x * 32 + o;
  const bool live = mo * group < p.M;
  float v4[4] = {0.f, 0.f, 0.f, 0.f};
  if (live)
    for (int g = 0; g < group; ++g) {
      const float* src = ca.part + ((size_t)blockIdx.y * CS_SLICES + q * (CS_SLICES / 8)) * 1024 + mo * group + g;
#pragma unroll
      for (int sl = 0; sl < CS_SLICES / 8; sl += 4)
#pragma unroll
        for (int u = 0; u < 4; ++u) v4[u] += src[(size_t)(sl + u) * 1024];
    }
  red[q][o] = (v4[0] + v4[1]) + (v4[2] + v4[3]);
  __syncthreads();
  if (q || !live) return;
  float v = red[0][o];
#pragma unroll
  for (int k = 1; k < 8; ++k) v += red[k][o];
  p.out[mo] = v;
  if (p.out2) p.out2[mo] = v;
  if (ca.nbc && (int)blockIdx.y == ca.bc_prob)
    for (int i = 0; i < ca.nbc; ++i) ca.bc[i][mo] = v;
}
template <bool F32>
int colsum_launch(ColsumArgs& ca, int nprob, int maxM, int group, hipStream_t s) {
  bool wide8 = !F32;    // every problem bf16 with 256 or 512 channels, rows 16-byte aligned: the 16-byte kernel
  for (int i = 0; i < nprob && wide8; ++i)
    wide8 = (ca.prob[i].M == 256 || ca.prob[i].M == 512) && ca.prob[i].ld % 8 == 0 && ca.prob[i].bs % 8 == 0 && ((size_t)ca.prob[i].y & 15) == 0;
  if (wide8) k_colsum_part8<<<dim3(1, CS_SLICES, nprob), 256, 0, s>>>(ca);
  else k_colsum_part<F32><<<dim3(1, CS_SLICES, nprob), 256, 0, s>>>(ca);
  k_colsum_sum<<<dim3((maxM / group + 31) / 32, nprob), 256, 0, s>>>(ca, group);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}


// ---- the <= 8-channel edges of the stack ----------------------------------------------------------------------
// start conv (glow.py:156): h0[b][HALO + n][c] = sum_j Ws[c][j] a0[b][j][n] + bs[c]
// a thread owns 4 channels (weights and bias in registers) and walks 8 positions; a workgroup covers 32 positions.  (One position per
// thread group -- 3 750 workgroups of ~40 instructions per thread at batch 12 -- took 20 us for 7.7 MB.)  Same FMA order per output.
__global__ __launch_bounds__(256) void k_t_start(const float* __restrict__ a0, const float* __restrict__ w, const float* __restrict__ bias,
                                                 bf16_t* __restrict__ h0, int nin, int L, int Lp) {
  const int c4 = (threadIdx.x & 63) * 4, b = blockIdx.y, n0 = blockIdx.x * 32 + (threadIdx.x >> 6) * 8;
  float wr[4][4], bv[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    bv[t] = bias[c4 + t];
#pragma unroll
    for (int j = 0; j < 4; ++j) wr[t][j] = j < nin ? w[(c4 + t) * nin + j] : 0.0f;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int n = n0 + i;
    if (n >= L) break;
    float v[4] = {bv[0], bv[1], bv[2], bv[3]};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < nin) {
        const float a = a0[((size_t)b * nin + j) * L + n];
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = fmaf(wr[t][j], a, v[t]);
      }
    *reinterpret_cast<uint2*>(h0 + ((size_t)b * Lp + HALO + n) * C + c4) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
  }
}

// end conv (glow.py:175): out[b][j][n] = sum_c We[j][c] skip[b][n][c] + be[j].  A wave takes 8 positions: lane = (position, slice of 32
// channels) reads 128 contiguous bytes of its fp32 row, forms its slice's <= 8 dot products with the weights from LDS ([c][j]), and the
// 8 slices of a position meet in three shuffle steps (a wave per position with a 6-step reduction per output cost 12 us at batch 12).
__global__ __launch_bounds__(256) void k_t_end(const float* __restrict__ skip, const float* __restrict__ w, const float* __restrict__ bias,
                                               float* __restrict__ out, int nout, int L, int Lr) {
  __shared__ __attribute__((aligned(16))) float sw[C * 8];
  for (int i = threadIdx.x; i < C * 8; i += 256) sw[i] = (i & 7) < nout ? w[(i & 7) * C + (i >> 3)] : 0.0f;
  __syncthreads();
  const int lane = threadIdx.x & 63, ps = lane >> 3, sl = lane & 7, b = blockIdx.y;
  const int n = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + ps;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (n < L) {
    const float4* row = reinterpret_cast<const float4*>(skip + ((size_t)b * Lr + n) * C + 32 * sl);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 x = row[q];
      const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float4 w0 = *reinterpret_cast<const float4*>(sw + (32 * sl + 4 * q + e) * 8), w1 = *reinterpret_cast<const float4*>(sw + (32 * sl + 4 * q + e) * 8 + 4);
        v[0] = fmaf(w0.x, xv[e], v[0]); v[1] = fmaf(w0.y, xv[e], v[1]); v[2] = fmaf(w0.z, xv[e], v[2]); v[3] = fmaf(w0.w, xv[e], v[3]);
        v[4] = fmaf(w1.x, xv[e], v[4]); v[5] = fmaf(w1.y, xv[e], v[5]); v[6] = fmaf(w1.z, xv[e], v[6]); v[7] = fmaf(w1.w, xv[e], v[7]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) v[j] += __shfl_xor(v[j], off, 64);
  if (sl == 0 && n < L)
    for (int j = 0; j < nout; ++j) out[((size_t)b * nout + j) * L + n] = v[j] + bias[j];
}

// backward of the end conv w.r.t. its input: dskip[b][n][c] = sum_j We[j][c] dout[b][j][n]  (bf16); a thread owns 4 channels (their
// weights in registers) and walks 8 positions, a workgroup covers 32 positions; same FMA order per output as one position per thread
__global__ __launch_bounds__(256) void k_t_end_bwd(const float* __restrict__ dout, const float* __restrict__ w, bf16_t* __restrict__ dskip,
                                                   int nout, int L, int Lr) {
  const int c4 = (threadIdx.x & 63) * 4, b = blockIdx.y, n0 = blockIdx.x * 32 + (threadIdx.x >> 6) * 8;
  float wr[8][4];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int t = 0; t < 4; ++t) wr[j][t] = j < nout ? w[j * C + c4 + t] : 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int n = n0 + i;
    if (n >= L) break;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < nout) {
        const float d = dout[((size_t)b * nout + j) * L + n];
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = fmaf(wr[j][t], d, v[t]);
      }
    *reinterpret_cast<uint2*>(dskip + ((size_t)b * Lr + n) * C + c4) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
  }
}

// small-channel weight gradients: out[j][c] = sum_{b,n} small[b][j][n] * wide[b][n][c]  (j < nj <= 8, c < 256),
// `wide` fp32 or bf16; plus optional column sums of small (bias of the end conv) and of wide (bias of the start conv).
// Partial sums per workgroup, then a fixed-order sum (deterministic).
template <bool WIDE_BF16>
__global__ __launch_bounds__(256) void k_small_wgrad_part(const float* __restrict__ small, const void* __restrict__ wide, long wide_bs,
                                                          int wide_row0, float* __restrict__ part, int nj, int B, int L, int nparts) {
  // workgroup = a contiguous range of the B*L rows; its slice of `small` (<= 8 x SW values) is staged in LDS first.  A thread
  // owns FOUR adjacent channels (one 16-byte load of fp32, 8 bytes of bf16) of every fourth row: four row lanes x 64 channel
  // quads, so a row range is walked in a quarter of the steps with 4x wider loads (one channel per thread walking every row
  // cost 33-45 us per launch at batch 12: a chain of dependent-latency loads); the row lanes meet in LDS in lane order.
  constexpr int SW = 128;
  __shared__ float ssm[8][SW];
  __shared__ float red[3][9][C];
  const int c4 = (threadIdx.x & 63) * 4, rl = threadIdx.x >> 6;
  const long total = (long)B * L, i0 = total * blockIdx.x / nparts, i1 = total * (blockIdx.x + 1) / nparts;
  float acc[9][4];
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[j][t] = 0.0f;
  for (long base = i0; base < i1; base += SW) {
    const int cnt = (int)((i1 - base) < SW ? (i1 - base) : SW);
    __syncthreads();
    for (int e = threadIdx.x; e < nj * cnt; e += 256) {
      const int j = e / cnt, r = e - j * cnt;
      const long i = base + r;
      const int b = (int)(i / L), n = (int)(i - (long)b * L);
      ssm[j][r] = small[((size_t)b * nj + j) * L + n];
    }
    __syncthreads();
#pragma unroll 4
    for (int r = rl; r < cnt; r += 4) {
      const long i = base + r;
      const int b = (int)(i / L), n = (int)(i - (long)b * L);
      const size_t o = (size_t)b * wide_bs + (size_t)(wide_row0 + n) * C + c4;
      float wv[4];
      if constexpr (WIDE_BF16) {
        const uint2 x = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(wide) + o);
        wv[0] = lo2f(x.x); wv[1] = hi2f(x.x); wv[2] = lo2f(x.y); wv[3] = hi2f(x.y);
      } else {
        const float4 x = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(wide) + o);
        wv[0] = x.x; wv[1] = x.y; wv[2] = x.z; wv[3] = x.w;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < nj) {
          const float sv = ssm[j][r];
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[j][t] = fmaf(sv, wv[t], acc[j][t]);
        }
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[8][t] += wv[t];
    }
  }
  // row lanes 1..3 hand their sums to lane 0, which adds them in lane order
  if (rl) {
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) red[rl - 1][j][c4 + t] = acc[j][t];
  }
  __syncthreads();
  if (rl) return;
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    float4 v = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
#pragma unroll
    for (int l = 0; l < 3; ++l) { v.x += red[l][j][c4]; v.y += red[l][j][c4 + 1]; v.z += red[l][j][c4 + 2]; v.w += red[l][j][c4 + 3]; }
    *reinterpret_cast<float4*>(part + ((size_t)blockIdx.x * 9 + j) * C + c4) = v;
  }
}
// out_w[(j, c)] laid out by (o_sj, o_sc); out_wsum[c] = column sums of wide (may be null).
// Workgroup (j, 16-channel block): 16 channels x 16 groups of the partials; a thread adds its nparts / 16 partials on four
// interleaved chains, the groups meet in LDS in group order (fixed order: bit-reproducible).  (Nine workgroups of 1024
// threads walking 64 partials each cost 21 us per launch, 24 launches per step.)
__global__ __launch_bounds__(256) void k_small_wgrad_sum(const float* __restrict__ part, int nparts, int nj, float* __restrict__ out_w, int o_sj,
                                                          int o_sc, float* __restrict__ out_wsum) {
  __shared__ float red[16][16];
  const int cl = threadIdx.x & 15, q = threadIdx.x >> 4, c = blockIdx.x * 16 + cl, j = blockIdx.y;
  const int p0 = nparts * q / 16, p1 = nparts * (q + 1) / 16;
  float v4[4] = {0.f, 0.f, 0.f, 0.f};
  for (int p = p0; p < p1; p += 4)
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (p + u < p1) v4[u] += part[((size_t)(p + u) * 9 + j) * C + c];
  red[q][cl] = (v4[0] + v4[1]) + (v4[2] + v4[3]);
  __syncthreads();
  if (q) return;
  float v = red[0][cl];
#pragma unroll
  for (int k = 1; k < 16; ++k) v += red[k][cl];
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
// backward of the start conv w.r.t. its input: da0[b][j][n] = sum_c Ws[c][j] dh0[b][n][c].  A wave takes 8 positions: lane = (position,
// slice of 32 channels) reads 64 contiguous bytes of its row, forms its slice's dot products with the weights from LDS, and the 8
// slices of a position meet in three shuffle steps (a wave per position with a 6-step reduction per input channel took 20 us).
__global__ __launch_bounds__(256) void k_t_start_bwd(const bf16_t* __restrict__ dh0, const float* __restrict__ w, float* __restrict__ da0,
                                                     int nin, int L, int Lr) {
  __shared__ float sw[C * 4];
  for (int i = threadIdx.x; i < C * 4; i += 256) sw[i] = (i & 3) < nin ? w[(i >> 2) * nin + (i & 3)] : 0.0f;
  __syncthreads();
  const int lane = threadIdx.x & 63, ps = lane >> 3, sl = lane & 7, b = blockIdx.y;
  const int n = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + ps;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (n < L) {
    const uint4* row = reinterpret_cast<const uint4*>(dh0 + ((size_t)b * Lr + n) * C + 32 * sl);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 d = row[q];
      const float dv[8] = {lo2f(d.x), hi2f(d.x), lo2f(d.y), hi2f(d.y), lo2f(d.z), hi2f(d.z), lo2f(d.w), hi2f(d.w)};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float4 ww = *reinterpret_cast<const float4*>(sw + (32 * sl + 8 * q + e) * 4);
        v[0] = fmaf(ww.x, dv[e], v[0]); v[1] = fmaf(ww.y, dv[e], v[1]); v[2] = fmaf(ww.z, dv[e], v[2]); v[3] = fmaf(ww.w, dv[e], v[3]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) v[j] += __shfl_xor(v[j], off, 64);
  if (sl == 0 && n < L)
    for (int j = 0; j < nin; ++j) da0[((size_t)b * nin + j) * L + n] = v[j];
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

// ---- upsampling ConvTranspose1d(80, 80, K, stride hop) + crop + 8-sample regroup (glow.py:184-186, 214-222) -----------
// forward, straight into the bf16 position-major conditioning operand:
//   spect[b][l][8m + g] = bu[m] + sum_{m'} sum_{j: 0 <= k = n - (q-j) hop < K} mel[b][m'][q - j] Wu[m'][m][k],  n = 8l + g = q hop + pp
// One workgroup = (batch, output channel m, block of UQ frames); thread pp owns the hop-phase pp and keeps UQ
// accumulators so every weight it loads is reused UQ times; the mel values are LDS broadcasts.
constexpr int UQ = 16, UMAXJ = 8;
__global__ __launch_bounds__(256) void k_up_fwd(const float* __restrict__ mel, const float* __restrict__ W, const float* __restrict__ bias,
                                                bf16_t* __restrict__ spect, int T, int nm, int hop, int ksize, int Lr, int n_limit) {
  extern __shared__ __attribute__((aligned(16))) float smel[];  // [nm][UQ + UMAXJ]
  const int b = blockIdx.z, m = blockIdx.y, q0 = blockIdx.x * UQ;
  const int nj = (ksize + hop - 1) / hop, SW = UQ + UMAXJ;
  for (int i = threadIdx.x; i < nm * SW; i += blockDim.x) {
    const int mp = i / SW, tt = i % SW, t = q0 - (UMAXJ - 1) + tt;
    smel[i] = (t >= 0 && t < T && tt < SW - 1) ? mel[((size_t)b * nm + mp) * T + t] : 0.0f;
  }
  __syncthreads();
  for (int pp = threadIdx.x; pp < hop; pp += blockDim.x) {
    float acc[UQ];
    const float bv = bias[m];
#pragma unroll
    for (int q = 0; q < UQ; ++q) acc[q] = bv;
    for (int j = 0; j < nj; ++j) {
      const int k = pp + j * hop;
      if (k < ksize)
#pragma unroll 8
        for (int mp = 0; mp < nm; ++mp) {      // (unrolled: eight weight loads in flight instead of one load -> 16 FMAs -> next load: 530 -> 357 us
                                               //  at batch 12; holding a channel's 24 staged frames in registers across the taps was slower: 522)
          const float wv = W[((size_t)mp * nm + m) * ksize + k];
          const float* sm = smel + mp * SW + (UMAXJ - 1) - j;
#pragma unroll
          for (int q = 0; q < UQ; ++q) acc[q] = fmaf(sm[q], wv, acc[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < UQ; ++q) {
      const int n = (q0 + q) * hop + pp;
      if (q0 + q < T && n < n_limit) spect[((size_t)b * Lr + (n >> 3)) * (nm * 8) + m * 8 + (n & 7)] = f2bf(acc[q]);
    }
  }
}
// backward w.r.t. the kernel: dWu[m'][m][k] = sum_{b,q} mel[b][m'][q] * dup[b][m][q hop + k], dup[b][m][n] = dspect[b][n/8][8m + n%8]
// (n < n_limit).  One workgroup = (output channel m, 256 taps k); a thread owns one k and all nm input channels m'.  The mel
// frames are staged UWQ at a time (one barrier pair per UWQ frames instead of per frame: the loop was 756 iterations of
// barrier -> 80 broadcasts -> barrier, 742 us at batch 12) and the gradient loads of the next frames are in flight under the FMAs.
constexpr int UWQ = 16;
template <int NM>
__global__ __launch_bounds__(256) void k_up_wgrad(const float* __restrict__ mel, const float* __restrict__ dspect, float* __restrict__ dW,
                                                  int B, int T, int hop, int ksize, int Lr, int n_limit) {
  __shared__ float smel[UWQ][NM];
  const int m = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
  float acc[NM];
#pragma unroll
  for (int i = 0; i < NM; ++i) acc[i] = 0.0f;
  for (int b = 0; b < B; ++b)
    for (int q0 = 0; q0 < T; q0 += UWQ) {
      __syncthreads();
      for (int e = threadIdx.x; e < UWQ * NM; e += 256) {
        const int i = e / UWQ, qq = e - i * UWQ;       // (frames fastest: consecutive threads read consecutive mel values)
        smel[qq][i] = q0 + qq < T ? mel[((size_t)b * NM + i) * T + q0 + qq] : 0.0f;
      }
      __syncthreads();
      float d[UWQ];
#pragma unroll
      for (int qq = 0; qq < UWQ; ++qq) {
        const int n = (q0 + qq) * hop + k;
        d[qq] = (k < ksize && q0 + qq < T && n < n_limit) ? dspect[((size_t)b * Lr + (n >> 3)) * (NM * 8) + m * 8 + (n & 7)] : 0.0f;
      }
#pragma unroll
      for (int qq = 0; qq < UWQ; ++qq)
#pragma unroll
        for (int i = 0; i < NM; ++i) acc[i] = fmaf(smel[qq][i], d[qq], acc[i]);
    }
  if (k < ksize)
#pragma unroll
    for (int i = 0; i < NM; ++i) dW[((size_t)i * NM + m) * ksize + k] = acc[i];
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
struct ScratchLayout { size_t w1, w2, rst, int_, condt, dpre, dh, dskip, part, cspart, wgpart, wgpart_bytes, total; size_t w1_one, w2_one, rst_one, int_one, dpre_one, dh_one; };
constexpr int K1 = 3 * C + NCOND;   // 1408
#ifndef FACPPG_SMALL_PARTS
#define FACPPG_SMALL_PARTS 256
#endif
constexpr int SMALL_PARTS = FACPPG_SMALL_PARTS;   // partial sums of the <= 8-channel weight gradients (one block each)
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
  s.cspart = take((size_t)MAXCS * CS_SLICES * 1024 * 4);
  s.wgpart_bytes = (size_t)3 * nl * 5 * (2 * C) * C * 4;      // 5 splits of the largest batch (3 taps x layers x [512 x 256]; k_wgrad2)
  s.wgpart = take(s.wgpart_bytes);
  s.total = off;
  return s;
}

// collects pack jobs and launches them as one grid (blockIdx.y = job)
struct Packer {
  PackBatch pb;
  int n = 0, max_total = 0;
  void add(const float* src, uint4* dst, int M, int KG, int k_base, int Cin, int taps, long sm, long sc, long st, long off, int gate_rows) {
    pb.e[n++] = PackArgs{src, dst, M, KG, k_base, Cin, taps, gate_rows, sm, sc, st, off};
    max_total = std::max(max_total, ((M + 31) / 32) * (Cin * taps / 16) * 64);
  }
  int launch(hipStream_t s) {
    if (!n) return FACPPG_OK;
    bool rows = true;       // every job row-contiguous along the reduction (16-byte aligned rows): the coalesced kernel
    for (int i = 0; i < n && rows; ++i) {
      const PackArgs& e = pb.e[i];
      rows = (e.taps == 1 || e.st == 1) && e.sc == e.taps && e.taps <= 3 && e.Cin % 8 == 0 && e.sm % 4 == 0 && e.off % 4 == 0 && ((size_t)e.src & 15) == 0;
    }
    if (rows) {
      int mt = 0;
      for (int i = 0; i < n; ++i) mt = std::max(mt, (pb.e[i].M + 31) / 32 * 32 * (pb.e[i].Cin / 8));
      k_pack_rows_bf16<<<dim3((mt + 255) / 256, n), 256, 0, s>>>(pb);
    } else {
      k_pack_bf16<<<dim3((max_total + 255) / 256, n), 256, 0, s>>>(pb);
    }
    n = 0; max_total = 0;
    FACPPG_HIP_CHECK(hipGetLastError());
    return FACPPG_OK;
  }
};

int check_wn(const facppg_wn_weights* w, int n_in, int nl, int B, int L) {
  FACPPG_REQUIRE(w && w->start_w && w->start_b && w->end_w && w->end_b, FACPPG_EINVAL, "NULL weight pointer");
  FACPPG_REQUIRE(n_in >= 1 && n_in <= 4 && nl >= 1 && nl <= 8 && B > 0 && B <= 65535 && L > 0, FACPPG_EINVAL, "bad n_in/n_layers/B/L");
  for (int i = 0; i < nl; ++i)
    FACPPG_REQUIRE(w->in_w[i] && w->in_b[i] && w->cond_w[i] && w->cond_b[i] && w->rs_w[i] && w->rs_b[i], FACPPG_EINVAL,
                   "NULL weight pointer (layer %d)", i);
  return FACPPG_OK;
}

struct AddBatch { const float* a[8]; const float* b[8]; };
__global__ void k_add2(AddBatch ab, float* out, int n) {   // out[y][i] = a[y][i] + b[y][i]
  const int i = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (i < n) out[(size_t)y * n + i] = ab.a[y][i] + ab.b[y][i];
}

}  // namespace
}  // namespace facppg

using namespace facppg;

extern "C" int facppg_wn_bf16_padded_len(int L) { return L > 0 ? pad_len(L) : 0; }

extern "C" size_t facppg_wn_bf16_state_bytes(int n_layers, int B, int L) {
  if (n_layers < 1 || n_layers > 8 || B <= 0 || L <= 0) return 0;
  return state_layout(n_layers, B, pad_len(L)).total;
}
extern "C" int facppg_wn_bf16_launch_plan(int n_layers, int B, int L) {
  if (n_layers < 1 || n_layers > 8 || B <= 0 || L <= 0) return FACPPG_EINVAL;
  int plan = tile_positions(B, L) << 8;
  if (fused_fwd_enabled(B, L)) plan |= 1;
  if (fused_bwd_enabled(B, L) && n_layers > 1) plan |= 2;
  if (wgrad2_wanted(n_layers * 3, 2 * C, C, B, L, scratch_layout(n_layers, B, pad_len(L)).wgpart_bytes)) plan |= 4;
  return plan;
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

// ---- the upsampler as exact-fp32 MFMA GEMMs (round 6) ----------------------------------------------------------------------------
// ConvTranspose1d(80, 80, K, stride hop): sample n = q hop + pp of channel m is sum_{j, m'} mel[m'][q - j] Wu[m'][m][pp + j hop], i.e.
//   out[(b, q)][(m, pp)] = sum_{k = (j, m')} melshift[(b, q)][k] * Wfold[k][(m, pp)],    nj = ceil(K / hop) taps,
// a [B Tq x 80 nj] x [80 nj x 80 hop] matrix product (k_gemm, facppg_gemm.h), and its weight gradient the product with the roles of
// the position and the tap axis exchanged.  k_up_fwd / k_up_wgrad (scalar fp32 FMA loops, 377 + 335 us at batch 12) stay as the
// FACPPG_UPSAMPLE_GEMM=0 path and the parity reference of the tests.
struct UpDims { int B, T, Tq, nm, hop, ksize, nj, K, N, rows, L, Lr; };
// Wfold[k = j nm + m'][n = m hop + pp] = Wu[m'][m][pp + j hop] (0 beyond the kernel)
__global__ void k_up_fold(const float* __restrict__ W, float* __restrict__ out, UpDims d) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)d.K * d.N) return;
  const int k = (int)(i / d.N), n = (int)(i - (size_t)k * d.N);
  const int j = k / d.nm, mp = k - j * d.nm, m = n / d.hop, pp = n - m * d.hop, t = pp + j * d.hop;
  out[i] = t < d.ksize ? W[((size_t)mp * d.nm + m) * d.ksize + t] : 0.0f;
}
// A-operand image (pack_a's layout) of melshift: rows (b, q), entries k = j nm + m': mel[b][m'][q - j]; TRANSPOSED: rows k, entries (b, q)
template <bool TRANSPOSED>
__global__ void k_up_pack_mel(const float* __restrict__ mel, float4* __restrict__ dst, UpDims d, int M, int K, int KG) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int MB = (M + 31) / 32;
  if (idx >= MB * (KG + 1) * 64) return;
  const int lane = idx & 63, g = (idx >> 6) % (KG + 1), mb = (idx >> 6) / (KG + 1);
  const int row = mb * 32 + (lane & 31);
  float v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int col = 8 * g + 4 * (lane >> 5) + t;
    const int r = TRANSPOSED ? col : row, k = TRANSPOSED ? row : col;      // r = (b, q), k = (j, m')
    float x = 0.0f;
    if (row < M && col < K && g < KG) {
      const int bb = r / d.Tq, q = r - bb * d.Tq, j = k / d.nm, mp = k - j * d.nm;
      if (q - j >= 0) x = mel[((size_t)bb * d.nm + mp) * d.T + q - j];
    }
    v[t] = x;
  }
  dst[idx] = make_float4(v[0], v[1], v[2], v[3]);
}
// out[(b, q)][m hop + pp] (+ bias[m]) -> spect bf16 [b][(q hop + pp) / 8][8 m + (q hop + pp) % 8]: 8-sample pieces (l', m) of frame q.
// One workgroup per (b, q): rows of `out` in, whole spect rows out (through LDS).
__global__ __launch_bounds__(256) void k_up_regroup(const float* __restrict__ out, const float* __restrict__ bias, bf16_t* __restrict__ spect, UpDims d) {
  extern __shared__ __attribute__((aligned(16))) float sm[];       // [N]
  const int r = blockIdx.x, b = r / d.Tq, q = r - b * d.Tq, P = d.hop / 8;
  const float4* src = reinterpret_cast<const float4*>(out + (size_t)r * d.N);
  for (int i = threadIdx.x; i < d.N / 4; i += blockDim.x) reinterpret_cast<float4*>(sm)[i] = src[i];
  __syncthreads();
  for (int i = threadIdx.x; i < P * d.nm; i += blockDim.x) {     // piece (l', m), m fastest: consecutive threads write consecutive 16 bytes
    const int lp = i / d.nm, m = i - lp * d.nm;
    const int l = q * P + lp;
    if (l >= d.L) continue;
    const float* x = sm + m * d.hop + 8 * lp;
    const float bv = bias[m];
    *reinterpret_cast<uint4*>(spect + ((size_t)b * d.Lr + l) * (d.nm * 8) + m * 8) =
        make_uint4(pack2(x[0] + bv, x[1] + bv), pack2(x[2] + bv, x[3] + bv), pack2(x[4] + bv, x[5] + bv), pack2(x[6] + bv, x[7] + bv));
  }
}
// dup[(b, q)][m hop + pp] = dspect[b][(q hop + pp) / 8][8 m + (q hop + pp) % 8] (0 for samples >= 8 L): the inverse regrouping
__global__ __launch_bounds__(256) void k_up_ungroup(const float* __restrict__ dspect, float* __restrict__ out, UpDims d) {
  extern __shared__ __attribute__((aligned(16))) float sm[];       // [N] as [m][pp]
  const int r = blockIdx.x, b = r / d.Tq, q = r - b * d.Tq, P = d.hop / 8;
  for (int i = threadIdx.x; i < P * d.nm * 2; i += blockDim.x) {  // half pieces (l', m, half): 16-byte loads along a dspect row
    const int hf = i & 1, pm = i >> 1, lp = pm / d.nm, m = pm - lp * d.nm;
    const int l = q * P + lp;
    const float4 v = l < d.L ? *reinterpret_cast<const float4*>(dspect + ((size_t)b * d.Lr + l) * (d.nm * 8) + m * 8 + 4 * hf) : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(sm + m * d.hop + 8 * lp + 4 * hf) = v;
  }
  __syncthreads();
  float4* dst = reinterpret_cast<float4*>(out + (size_t)r * d.N);
  for (int i = threadIdx.x; i < d.N / 4; i += blockDim.x) dst[i] = reinterpret_cast<const float4*>(sm)[i];
}
// dWu[m'][m][k] = dWfold[(k / hop) nm + m'][m hop + k % hop]
__global__ void k_up_unfold(const float* __restrict__ fold, float* __restrict__ dW, UpDims d) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)d.nm * d.nm * d.ksize) return;
  const int t = (int)(i % d.ksize), m = (int)((i / d.ksize) % d.nm), mp = (int)(i / ((size_t)d.ksize * d.nm));
  const int j = t / d.hop, pp = t - j * d.hop;
  dW[i] = fold[(size_t)(j * d.nm + mp) * d.N + m * d.hop + pp];
}
UpDims up_dims(int B, int T, int nm, int hop, int ksize, int L) {
  UpDims d;
  d.B = B; d.T = T; d.nm = nm; d.hop = hop; d.ksize = ksize; d.L = L; d.Lr = pad_len(L);
  d.Tq = std::min(T, (L * 8 + hop - 1) / hop);     // frames q with q*hop < N produce samples < N
  d.nj = (ksize + hop - 1) / hop; d.K = d.nj * nm; d.N = nm * hop; d.rows = B * d.Tq;
  return d;
}
struct UpWs { size_t fold, aimg, out, total; };
UpWs up_ws(const UpDims& d, bool backward) {   // forward: Wfold | A(melshift) | out;  backward: dWfold | A(melshift^T) | dup
  UpWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  w.fold = take((size_t)round_up(d.K, 32) * d.N * 4);
  w.aimg = take(packed_a_float4s(backward ? d.K : d.rows, backward ? d.rows : d.K) * 16);
  w.out = take((size_t)d.rows * d.N * 4);
  w.total = off;
  return w;
}
bool upsample_gemm() {
  const char* e = getenv("FACPPG_UPSAMPLE_GEMM");
  return !(e && e[0] == '0');
}

extern "C" size_t facppg_upsample_forward_workspace_bytes(int B, int T, int n_mel, int hop, int ksize, int L) {
  if (B <= 0 || T <= 0 || n_mel <= 0 || hop <= 0 || ksize <= 0 || L <= 0) return 0;
  return up_ws(up_dims(B, T, n_mel, hop, ksize, L), false).total;
}

extern "C" int facppg_upsample_regroup_bf16(const float* mel_dev, const float* up_w_dev, const float* up_b_dev, int B, int T, int n_mel,
                                            int hop, int ksize, int L, void* spect_pm_dev, void* ws_dev, size_t ws_bytes, void* stream_) {
  FACPPG_REQUIRE(mel_dev && up_w_dev && up_b_dev && spect_pm_dev, FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(B > 0 && T > 0 && L > 0 && n_mel == 80 && hop % 8 == 0 && hop > 0 && (ksize + hop - 1) / hop <= UMAXJ, FACPPG_EUNSUPPORTED,
                 "upsample: n_mel must be 80, hop a multiple of 8, kernel/hop <= %d", UMAXJ);
  FACPPG_REQUIRE((long)(T - 1) * hop + ksize >= (long)L * 8, FACPPG_EINVAL, "upsampled mel is shorter than the audio (glow.py:216)");
  hipStream_t s = (hipStream_t)stream_;
  const UpDims d = up_dims(B, T, n_mel, hop, ksize, L);
  FACPPG_HIP_CHECK(hipMemsetAsync(spect_pm_dev, 0, (size_t)B * d.Lr * n_mel * 8 * 2, s));   // rows >= L (and samples no frame reaches) are zero
  if (!upsample_gemm() || !ws_dev) {
    k_up_fwd<<<dim3((d.Tq + UQ - 1) / UQ, n_mel, B), 256, (size_t)n_mel * (UQ + UMAXJ) * 4, s>>>(mel_dev, up_w_dev, up_b_dev, (bf16_t*)spect_pm_dev, T,
                                                                                              n_mel, hop, ksize, d.Lr, L * 8);
    FACPPG_HIP_CHECK(hipGetLastError());
    return FACPPG_OK;
  }
  const UpWs w = up_ws(d, false);
  FACPPG_REQUIRE(ws_bytes >= w.total, FACPPG_EWORKSPACE, "upsample workspace has %zu bytes, need %zu", ws_bytes, w.total);
  char* Wk = (char*)ws_dev;
  float* fold = (float*)(Wk + w.fold);
  float4* aimg = (float4*)(Wk + w.aimg);
  float* out = (float*)(Wk + w.out);
  k_up_fold<<<(unsigned)(((size_t)d.K * d.N + 255) / 256), 256, 0, s>>>(up_w_dev, fold, d);
  const int KG = gemm_kpad(d.K) / 8;
  k_up_pack_mel<false><<<((d.rows + 31) / 32 * (KG + 1) * 64 + 255) / 256, 256, 0, s>>>(mel_dev, aimg, d, d.rows, d.K, KG);
  GemmArgs g;
  g.A = aimg; g.M = d.rows; g.Cin = d.K; g.taps = 1; g.X = fold; g.x_bs = 0; g.ldx = d.N; g.N = d.N; g.C = out; g.c_bs = 0; g.ldc = d.N; g.B = 1;
  if (int rc = gemm_launch(g, s)) return rc;
  k_up_regroup<<<d.rows, 256, (size_t)d.N * 4, s>>>(out, up_b_dev, (bf16_t*)spect_pm_dev, d);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

extern "C" size_t facppg_upsample_backward_workspace_bytes(int B, int T, int n_mel, int hop, int ksize, int L) {
  const size_t cs = (size_t)MAXCS * CS_SLICES * 1024 * 4;
  if (B <= 0 || T <= 0 || n_mel <= 0 || hop <= 0 || ksize <= 0 || L <= 0) return cs;
  return cs + up_ws(up_dims(B, T, n_mel, hop, ksize, L), true).total;
}

extern "C" int facppg_upsample_regroup_backward(const float* mel_dev, const float* dspect_pm_dev, int B, int T, int n_mel, int hop, int ksize,
                                                int L, float* d_up_w_dev, float* d_up_b_dev, void* ws_dev, size_t ws_bytes, void* stream_) {
  FACPPG_REQUIRE(mel_dev && dspect_pm_dev && d_up_w_dev && d_up_b_dev && ws_dev, FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(B > 0 && T > 0 && L > 0 && n_mel == 80 && hop % 8 == 0 && hop > 0, FACPPG_EUNSUPPORTED, "upsample backward: n_mel must be 80, hop a multiple of 8");
  const size_t cs = (size_t)MAXCS * CS_SLICES * 1024 * 4;
  FACPPG_REQUIRE(ws_bytes >= cs, FACPPG_EWORKSPACE, "workspace too small");
  hipStream_t s = (hipStream_t)stream_;
  const UpDims d = up_dims(B, T, n_mel, hop, ksize, L);
  const UpWs w = up_ws(d, true);
  if (upsample_gemm() && ws_bytes >= cs + w.total && (ksize + hop - 1) / hop <= UMAXJ) {
    char* Wk = (char*)ws_dev + cs;
    float* fold = (float*)(Wk + w.fold);
    float4* aimg = (float4*)(Wk + w.aimg);
    float* dup = (float*)(Wk + w.out);
    k_up_ungroup<<<d.rows, 256, (size_t)d.N * 4, s>>>(dspect_pm_dev, dup, d);
    const int KG = gemm_kpad(d.rows) / 8;
    k_up_pack_mel<true><<<((d.K + 31) / 32 * (KG + 1) * 64 + 255) / 256, 256, 0, s>>>(mel_dev, aimg, d, d.K, d.rows, KG);
    GemmArgs g;
    g.A = aimg; g.M = d.K; g.Cin = d.rows; g.taps = 1; g.X = dup; g.x_bs = 0; g.ldx = d.N; g.N = d.N; g.C = fold; g.c_bs = 0; g.ldc = d.N; g.B = 1;
    if (int rc = gemm_launch(g, s)) return rc;
    k_up_unfold<<<(unsigned)(((size_t)n_mel * n_mel * ksize + 255) / 256), 256, 0, s>>>(fold, d_up_w_dev, d);
  } else {
    k_up_wgrad<80><<<dim3((ksize + 255) / 256, n_mel), 256, 0, s>>>(mel_dev, dspect_pm_dev, d_up_w_dev, B, T, hop, ksize, d.Lr, L * 8);
  }
  // d bias[m] = sum over every produced sample of channel m = column sums of dspect over its 8 regrouped channels
  ColsumArgs ca;
  memset(&ca, 0, sizeof(ca));
  ca.B = B; ca.L = L; ca.part = (float*)ws_dev;
  // 640 channels as two problems of 320 (k_colsum_part covers <= 512 channels per problem)
  const int half = n_mel * 4;
  ca.prob[0] = ColsumProb{dspect_pm_dev, (long)d.Lr * n_mel * 8, n_mel * 8, 0, half, d_up_b_dev};
  ca.prob[1] = ColsumProb{dspect_pm_dev + half, (long)d.Lr * n_mel * 8, n_mel * 8, 0, half, d_up_b_dev + half / 8};
  return colsum_launch<true>(ca, 2, half, 8, s);
}

// Zero rows [0, r0) and [r1, rows) of every image of a [images][rows][row_bytes] buffer: the conv's zero padding (the
// 128-row margins) and the rows between L and the 128-padded length, which the GEMM tiles read.  (Zeroing the whole
// buffers instead cost 0.5 GB of memset per flow and direction at batch 12: 1.5 ms of a 27 ms step.)
struct ZeroJob { char* base; long images, image_bytes, per_group, group_stride; int row_bytes, r0, r1, rows; };
struct ZeroBatch { ZeroJob job[3]; int n; };
// up to three such buffers per launch (blockIdx.z = buffer): the forward zeroes h / ts / acts, the backward dpre / dh / dskip.
// A buffer may be a strided set of `groups` such arrays (the same tensor of every flow's saved state: image y = group y / per_group).
__global__ void k_zero_rows(ZeroBatch zb) {
  const ZeroJob& j = zb.job[blockIdx.z];
  if ((long)blockIdx.y >= j.images) return;
  const long head = (long)j.r0 * j.row_bytes, n16 = (head + (long)(j.rows - j.r1) * j.row_bytes) / 16;
  const long g = (long)blockIdx.y / j.per_group, im = (long)blockIdx.y - g * j.per_group;
  char* img = j.base + g * j.group_stride + im * j.image_bytes;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
    const long byte = i * 16;
    *reinterpret_cast<uint4*>(byte < head ? img + byte : img + (long)j.r1 * j.row_bytes + (byte - head)) = make_uint4(0, 0, 0, 0);
  }
}

struct ZeroRows {
  ZeroBatch zb;
  long max_images = 0, max_n16 = 0;
  ZeroRows() { zb.n = 0; }
  void add(void* base, long images, long image_bytes, int row_bytes, int r0, int r1, int rows, long groups = 1, long group_stride = 0) {
    const long n16 = ((long)r0 * row_bytes + (long)(rows - r1) * row_bytes) / 16;
    if (n16 <= 0 || images <= 0 || groups <= 0) return;
    zb.job[zb.n++] = ZeroJob{(char*)base, images * groups, image_bytes, images, group_stride, row_bytes, r0, r1, rows};
    max_images = std::max(max_images, images * groups);
    max_n16 = std::max(max_n16, n16);
  }
  void launch(hipStream_t s) {
    if (!zb.n) return;
    const unsigned gx = (unsigned)((max_n16 + 255) / 256 > 64 ? 64 : (max_n16 + 255) / 256);
    k_zero_rows<<<dim3(gx, (unsigned)max_images, (unsigned)zb.n), 256, 0, s>>>(zb);
  }
};

// out[flow][layer][i] = a[flow][layer][i] + b[flow][layer][i]: the summed gate biases (in + cond) of several flows' stacks
constexpr int ADD_FLOWS = 12;
struct AddBatchN { const float* a[ADD_FLOWS * 8]; const float* b[ADD_FLOWS * 8]; float* out; long out_stride; };
__global__ void k_add2_flows(AddBatchN ab, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, f = blockIdx.z;
  if (i < n) ab.out[(size_t)f * ab.out_stride + (size_t)y * n + i] = ab.a[f * 8 + y][i] + ab.b[f * 8 + y][i];
}

// ---- the stack's launches, shared by the stand-alone WN entry points (facppg_wn_forward_bf16 / _backward_bf16) and the whole-model
// group entry points (facppg_glow_bf16_*): where the packed operand images, the saved state and the gradients in flight live is
// the caller's business.
struct WnPacked {   // bf16 A-operand images of one flow's stack (+ the summed gate biases), see pack_flow_images
  const uint4* w1; size_t w1_one; const uint4* w2; size_t w2_one; const uint4* rst; size_t rst_one; const uint4* int_; size_t int_one;
  const uint4* condt; const float* b1;
};
struct WnWork {     // gradients in flight of one flow's backward + the partial-sum buffers of its reductions
  char* dpre; size_t dpre_one; char* dh; size_t dh_one; bf16_t* dskip; float* cspart; float* wgpart; size_t wgpart_bytes;
  int wgpart_regions;    // 3: a region of wgpart_bytes per weight-gradient launch of a flow -> ONE ordered reduce for the three; 1: one shared region
};

// forward image jobs of one flow: K order tap 0 | tap 1 | tap 2 | cond, gate-interleaved rows; res/skip rows as they are
void add_forward_images(Packer& pk, const facppg_wn_weights* wts, int nl, char* w1, size_t w1_one, char* w2, size_t w2_one) {
  for (int i = 0; i < nl; ++i) {
    const int last = i == nl - 1;
    uint4* d1 = (uint4*)(w1 + w1_one * i);
    pk.add(wts->in_w[i], d1, 2 * C, K1 / 16, 0, C, 3, (long)C * 3, 3, 1, 0, 1);
    pk.add(wts->cond_w[i], d1, 2 * C, K1 / 16, 3 * C, NCOND, 1, NCOND, 1, 0, 0, 1);
    pk.add(wts->rs_w[i], (uint4*)(w2 + w2_one * i), last ? C : 2 * C, C / 16, 0, C, 1, C, 1, 0, 0, 0);
  }
}
// transposed images for the backward
void add_backward_images(Packer& pk, const facppg_wn_weights* wts, int nl, char* rst, size_t rst_one, char* int_, size_t int_one, uint4* condt) {
  for (int i = 0; i < nl; ++i) {
    const int last = i == nl - 1;
    // dacts[c] = sum_r Wrs[r][c] * [dh_next (res rows) | dskip (skip rows)][r]; last layer: skip rows only
    pk.add(wts->rs_w[i], (uint4*)(rst + rst_one * i), C, (last ? C : 2 * C) / 16, 0, last ? C : 2 * C, 1, 1, C, 0, 0, 0);
    // dh[m] += sum_{tap,o} Win[o][m][tap] * dpre[n - (tap-1) d][o]
    pk.add(wts->in_w[i], (uint4*)(int_ + int_one * i), C, 3 * 2 * C / 16, 0, 2 * C, 3, 3, (long)C * 3, 1, 0, 0);
    // dspect[j] = sum_{i,o} Wcond_i[o][j] * dpre_i[o]: one image, K = nl * 512
    pk.add(wts->cond_w[i], condt, NCOND, nl * 2 * C / 16, i * 2 * C, 2 * C, 1, 1, NCOND, 0, 0, 0);
  }
}

// the nl layers of one stack from h_0 (already in `S`) to the skip sum (glow.py:158-174)
int wn_layers_forward(const facppg_wn_weights* wts, int nl, const WnPacked& pw, const void* spect_pm_dev, int B, int L, char* S, hipStream_t s) {
  const int Lr = pad_len(L), Lp = HALO + Lr + HALO;
  const StateLayout st = state_layout(nl, B, Lr);
  const bool fused = fused_fwd_enabled(B, L);
  for (int i = 0; i < nl; ++i) {
    const int last = i == nl - 1, d = 1 << i;
    const bf16_t* h_in = (const bf16_t*)(S + st.h + st.h_one * i);
    const uint4* A1 = (const uint4*)((const char*)pw.w1 + pw.w1_one * i);
    const uint4* A2 = (const uint4*)((const char*)pw.w2 + pw.w2_one * i);
    if (fused) {
      WnFwdArgs f;
      memset(&f, 0, sizeof(f));
      f.A1 = A1; f.A2 = A2;
      f.h_in = h_in; f.h_out = (bf16_t*)(S + st.h + st.h_one * (i + 1)); f.h_bs = (long)Lp * C;
      f.spect = (const bf16_t*)spect_pm_dev; f.sp_bs = (long)Lr * NCOND;
      f.b1 = pw.b1 + 2 * C * i; f.b2 = wts->rs_b[i];
      f.acts = (bf16_t*)(S + st.acts + st.acts_one * i); f.ts = (bf16_t*)(S + st.ts + st.ts_one * i); f.skip = (float*)(S + st.skip);
      f.L = L; f.Lr = Lr; f.B = B; f.d = d; f.first = i == 0; f.last = last;
      if (int rc = wn_fwd_launch(f, s)) return rc;
      continue;
    }
    BGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A1; g.KG = K1 / 16; g.M = 2 * C; g.N = L; g.B = B; g.nseg = 4;
    for (int t = 0; t < 3; ++t) g.seg[t] = Seg{h_in, (long)Lp * C, C, HALO + (t - 1) * d, C};
    g.seg[3] = Seg{(const bf16_t*)spect_pm_dev, (long)Lr * NCOND, NCOND, 0, NCOND};
    g.mode = EP_GATE; g.bias = pw.b1 + 2 * C * i; g.Lr = Lr;
    g.acts = (bf16_t*)(S + st.acts + st.acts_one * i); g.ts = (bf16_t*)(S + st.ts + st.ts_one * i);
    if (int rc = bgemm_launch(g, s)) return rc;
    BGemmArgs r;
    memset(&r, 0, sizeof(r));
    r.A = A2; r.KG = C / 16; r.M = last ? C : 2 * C; r.N = L; r.B = B; r.nseg = 1;
    r.seg[0] = Seg{g.acts, (long)Lr * C, C, 0, C};
    r.mode = EP_RESSKIP; r.bias = wts->rs_b[i]; r.Lr = Lr; r.h_in = h_in; r.h_out = (bf16_t*)(S + st.h + st.h_one * (i + 1));
    r.h_bs = (long)Lp * C; r.h_row0 = HALO; r.skip = (float*)(S + st.skip); r.first = i == 0; r.last = last;
    if (int rc = bgemm_launch(r, s)) return rc;
  }
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
  // h: zero margins (the conv padding) and rows >= L; ts, acts: rows >= L meet zero gradients in k_wgrad, but 0 * NaN = NaN
  {
    ZeroRows z;
    z.add(S + st.h, (long)(nl + 1) * B, (long)Lp * C * 2, C * 2, HALO, HALO + L, Lp);
    z.add(S + st.ts, (long)nl * B, (long)Lr * 2 * C * 2, 2 * C * 2, 0, L, Lr);
    z.add(S + st.acts, (long)nl * B, (long)Lr * C * 2, C * 2, 0, L, Lr);
    z.launch(s);
  }
  {
    Packer pk;
    add_forward_images(pk, wts, nl, W + sc.w1, sc.w1_one, W + sc.w2, sc.w2_one);
    if (int rc = pk.launch(s)) return rc;
    AddBatch ab;
    for (int i = 0; i < nl; ++i) { ab.a[i] = wts->in_b[i]; ab.b[i] = wts->cond_b[i]; }
    k_add2<<<dim3(2, nl), 256, 0, s>>>(ab, b1, 2 * C);
  }
  k_t_start<<<dim3((L + 31) / 32, B), 256, 0, s>>>(a0_dev, wts->start_w, wts->start_b, (bf16_t*)(S + st.h), n_in, L, Lp);
  WnPacked pw;
  memset(&pw, 0, sizeof(pw));
  pw.w1 = (const uint4*)(W + sc.w1); pw.w1_one = sc.w1_one; pw.w2 = (const uint4*)(W + sc.w2); pw.w2_one = sc.w2_one; pw.b1 = b1;
  if (int rc = wn_layers_forward(wts, nl, pw, spect_pm_dev, B, L, S, s)) return rc;
  k_t_end<<<dim3((L + 31) / 32, B), 256, 0, s>>>((const float*)(S + st.skip), wts->end_w, wts->end_b, out_dev, 2 * n_in, L, Lr);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

// The data-gradient chain of one stack, from dskip (already formed from d(out) by the end conv's backward) down to dh_0.
int wn_layers_backward_data(int nl, const WnPacked& pw, const char* S, const WnWork& wk, int B, int L, hipStream_t s) {
  const int Lr = pad_len(L), Lp = HALO + Lr + HALO;
  const StateLayout st = state_layout(nl, B, Lr);
  bf16_t* dskip = wk.dskip;
  const bool fused = fused_bwd_enabled(B, L);
  auto rst = [&](int i) { return (const uint4*)((const char*)pw.rst + pw.rst_one * i); };
  auto int_ = [&](int i) { return (const uint4*)((const char*)pw.int_ + pw.int_one * i); };
  auto gate_bwd_launch = [&](int i) -> int {      // dpre_i = gate'(ts_i) * Wrs_i^T [dh_{i+1} ; dskip]
    const int last = i == nl - 1;
    const bf16_t* dh_next = (const bf16_t*)(wk.dh + wk.dh_one * (i + 1));
    BGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = rst(i); g.KG = (last ? C : 2 * C) / 16; g.M = C; g.N = L; g.B = B;
    if (last) { g.nseg = 1; g.seg[0] = Seg{dskip, (long)Lr * C, C, 0, C}; }
    else { g.nseg = 2; g.seg[0] = Seg{dh_next, (long)Lr * C, C, 0, C}; g.seg[1] = Seg{dskip, (long)Lr * C, C, 0, C}; }
    g.mode = EP_BWD_GATE; g.Lr = Lr; g.ts = (bf16_t*)(S + st.ts + st.ts_one * i); g.dpre = (bf16_t*)(wk.dpre + wk.dpre_one * i);
    g.dpre_bs = (long)Lp * 2 * C;
    return bgemm_launch(g, s);
  };
  auto conv_bwd_launch = [&](int i) -> int {      // dh_i = dh_{i+1} + sum_tap Win_i^T dpre_i(shifted)
    const int last = i == nl - 1, d = 1 << i;
    const bf16_t* dpre = (const bf16_t*)(wk.dpre + wk.dpre_one * i);
    BGemmArgs t;
    memset(&t, 0, sizeof(t));
    t.A = int_(i); t.KG = 3 * 2 * C / 16; t.M = C; t.N = L; t.B = B; t.nseg = 3;
    for (int tp = 0; tp < 3; ++tp) t.seg[tp] = Seg{dpre, (long)Lp * 2 * C, 2 * C, HALO - (tp - 1) * d, 2 * C};
    t.mode = EP_BWD_CONV; t.Lr = Lr; t.dh_next = last ? nullptr : (const bf16_t*)(wk.dh + wk.dh_one * (i + 1));
    t.dh_out = (bf16_t*)(wk.dh + wk.dh_one * i);
    return bgemm_launch(t, s);
  };
  if (fused && nl > 1) {
    if (int rc = gate_bwd_launch(nl - 1)) return rc;
    for (int i = nl - 1; i >= 1; --i) {           // conv backward of layer i + gate backward of layer i-1 in one launch
      WnBwdArgs f;
      memset(&f, 0, sizeof(f));
      f.A1 = int_(i); f.A2 = rst(i - 1);
      f.dpre_i = (const bf16_t*)(wk.dpre + wk.dpre_one * i); f.dpre_bs = (long)Lp * 2 * C;
      f.dh_next = i == nl - 1 ? nullptr : (const bf16_t*)(wk.dh + wk.dh_one * (i + 1));
      f.dh_out = (bf16_t*)(wk.dh + wk.dh_one * i); f.dskip = dskip;
      f.ts = (const bf16_t*)(S + st.ts + st.ts_one * (i - 1)); f.dpre_out = (bf16_t*)(wk.dpre + wk.dpre_one * (i - 1));
      f.L = L; f.Lr = Lr; f.B = B; f.d = 1 << i;
      if (int rc = wn_bwd_launch(f, s)) return rc;
    }
    if (int rc = conv_bwd_launch(0)) return rc;
  } else {
    for (int i = nl - 1; i >= 0; --i) {
      if (int rc = gate_bwd_launch(i)) return rc;
      if (int rc = conv_bwd_launch(i)) return rc;
    }
  }
  return FACPPG_OK;
}

// the conditioning gradient of all layers of one stack: dspect (+)= [Wcond_0^T | ... ] [dpre_0; ...]
int wn_layers_backward_dspect(int nl, const WnPacked& pw, const WnWork& wk, float* dspect_pm_dev, int accumulate_dspect, int B, int L,
                              hipStream_t s) {
  const int Lr = pad_len(L), Lp = HALO + Lr + HALO;
  const char* e_ds = getenv("FACPPG_TRAIN_DSPECT_BGEMM");     // =1: the k_bgemm<EP_ACC_F32> launch (bit-equality test, A/B timing)
  if (!(e_ds && e_ds[0] == '1')) {  // dspect over all layers at once
    DspectArgs c;
    memset(&c, 0, sizeof(c));
    c.A = pw.condt; c.dpre = (const bf16_t*)wk.dpre; c.dpre_one = (long)(wk.dpre_one / 2); c.dpre_bs = (long)Lp * 2 * C;
    c.out = dspect_pm_dev; c.nl = nl; c.L = L; c.Lr = Lr; c.B = B; c.accumulate = accumulate_dspect != 0;
    if (int rc = dspect_launch(c, s)) return rc;
  } else {
    BGemmArgs c;
    memset(&c, 0, sizeof(c));
    c.A = pw.condt; c.KG = nl * 2 * C / 16; c.M = NCOND; c.N = L; c.B = B; c.nseg = nl;
    for (int i = 0; i < nl; ++i) c.seg[i] = Seg{(const bf16_t*)(wk.dpre + wk.dpre_one * i), (long)Lp * 2 * C, 2 * C, HALO, 2 * C};
    c.mode = EP_ACC_F32; c.Lr = Lr; c.outf = dspect_pm_dev; c.ldo = NCOND; c.accumulate = accumulate_dspect != 0;
    if (int rc = bgemm_launch(c, s)) return rc;
  }
  return FACPPG_OK;
}

// Weight and bias gradients of the three convs of every layer: NT products over positions, batched over layers (x taps); bias
// gradients as column sums.
// (s: the NT products and their ordered reduce; s_bias: the column sums -- the same stream, or a parallel branch)
int wn_layers_backward_weights(const facppg_wn_grads* gr, int nl, const char* S, const WnWork& wk, const void* spect_pm_dev, int B, int L,
                               hipStream_t s, hipStream_t s_bias) {
  const int Lr = pad_len(L), Lp = HALO + Lr + HALO;
  const StateLayout st = state_layout(nl, B, Lr);
  bf16_t* dskip = wk.dskip;
  {
    WgradArgs wa;
    memset(&wa, 0, sizeof(wa));
    wa.B = B; wa.L = L; wa.Lr = Lr;
    for (int i = 0; i < nl; ++i)
      for (int tp = 0; tp < 3; ++tp) {
        WgradProb& p = wa.prob[i * 3 + tp];
        p.dy0 = (const bf16_t*)(wk.dpre + wk.dpre_one * i); p.dy1 = nullptr; p.dy_bs = (long)Lp * 2 * C; p.ldy = 2 * C; p.dy_row0 = HALO; p.msplit = 0;
        p.x = (const bf16_t*)(S + st.h + st.h_one * i); p.x_bs = (long)Lp * C; p.ldx = C; p.x_row0 = HALO + (tp - 1) * (1 << i);
        p.out = gr->in_w[i] + tp; p.o_sm = (long)C * 3; p.o_sk = 3; p.M = 2 * C; p.K = C;
      }
    ReduceSet rs;
    ReduceSet* defer = wk.wgpart_regions >= 3 ? &rs : nullptr;
    float* region[3];
    for (int r = 0; r < 3; ++r) region[r] = defer ? (float*)((char*)wk.wgpart + wk.wgpart_bytes * r) : wk.wgpart;
    if (int rc = wgrad_launch(wa, nl * 3, 2 * C, C, region[0], wk.wgpart_bytes, s, defer)) return rc;
    memset(&wa.prob, 0, sizeof(wa.prob));
    for (int i = 0; i < nl; ++i) {
      WgradProb& p = wa.prob[i];
      p.dy0 = (const bf16_t*)(wk.dpre + wk.dpre_one * i); p.dy_bs = (long)Lp * 2 * C; p.ldy = 2 * C; p.dy_row0 = HALO;
      p.x = (const bf16_t*)spect_pm_dev; p.x_bs = (long)Lr * NCOND; p.ldx = NCOND; p.x_row0 = 0;
      p.out = gr->cond_w[i]; p.o_sm = NCOND; p.o_sk = 1; p.M = 2 * C; p.K = NCOND;
    }
    if (int rc = wgrad_launch(wa, nl, 2 * C, NCOND, region[1], wk.wgpart_bytes, s, defer)) return rc;
    memset(&wa.prob, 0, sizeof(wa.prob));
    for (int i = 0; i < nl; ++i) {
      const int last = i == nl - 1;
      WgradProb& p = wa.prob[i];
      if (last) { p.dy0 = dskip; p.dy1 = nullptr; p.msplit = 0; p.M = C; }
      else { p.dy0 = (const bf16_t*)(wk.dh + wk.dh_one * (i + 1)); p.dy1 = dskip; p.msplit = C; p.M = 2 * C; }
      p.dy_bs = (long)Lr * C; p.ldy = C; p.dy_row0 = 0;
      p.x = (const bf16_t*)(S + st.acts + st.acts_one * i); p.x_bs = (long)Lr * C; p.ldx = C; p.x_row0 = 0;
      p.out = gr->rs_w[i]; p.o_sm = C; p.o_sk = 1; p.K = C;
    }
    if (int rc = wgrad_launch(wa, nl, 2 * C, C, region[2], wk.wgpart_bytes, s, defer)) return rc;
    if (defer)
      if (int rc = reduce_launch(rs, s)) return rc;
  }
  {  // bias gradients: column sums of dpre_i (in + cond biases share them) and of [dh_{i+1} | dskip], ONE pair of launches for the 2 nl
     // problems.  The skip half of every rs_b[i] is the column sum of the SAME dskip: summed once, written to every layer.  (Round 3
     // also tried leaving per-tile partial sums behind in the two backward GEMMs' epilogues instead of re-reading dpre / dh: the
     // 32-lane reductions cost those latency-bound launches 3-4 us each, more than the second pass they saved.)
    ColsumArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.B = B; ca.L = L; ca.part = wk.cspart;
    int np = 0;
    // (the conditioning conv's bias sees the same pre-activations: its gradient is the same sums, written by the same launch)
    for (int i = 0; i < nl; ++i) ca.prob[np++] = ColsumProb{wk.dpre + wk.dpre_one * i, (long)Lp * 2 * C, 2 * C, HALO, 2 * C, gr->in_b[i], gr->cond_b[i]};
    for (int i = 0; i + 1 < nl; ++i) ca.prob[np++] = ColsumProb{wk.dh + wk.dh_one * (i + 1), (long)Lr * C, C, 0, C, gr->rs_b[i], nullptr};
    ca.bc_prob = np;
    ca.prob[np++] = ColsumProb{dskip, (long)Lr * C, C, 0, C, gr->rs_b[nl - 1], nullptr};
    for (int i = 0; i + 1 < nl; ++i) ca.bc[ca.nbc++] = gr->rs_b[i] + C;
    if (int rc = colsum_launch<false>(ca, np, 2 * C, 1, s_bias)) return rc;
  }
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

int check_wn_grads(const facppg_wn_grads* gr, int nl) {
  FACPPG_REQUIRE(gr && gr->start_w && gr->start_b && gr->end_w && gr->end_b, FACPPG_EINVAL, "NULL gradient pointer");
  for (int i = 0; i < nl; ++i)
    FACPPG_REQUIRE(gr->in_w[i] && gr->in_b[i] && gr->cond_w[i] && gr->cond_b[i] && gr->rs_w[i] && gr->rs_b[i], FACPPG_EINVAL,
                   "NULL gradient pointer (layer %d)", i);
  return FACPPG_OK;
}

// Complete backward of the stack: da0, dspect (position-major fp32; overwritten, or added to when accumulate_dspect --
// the 12 flows of a step share one buffer) and every weight / bias gradient.
extern "C" int facppg_wn_backward_bf16(const facppg_wn_weights* wts, const facppg_wn_grads* gr, int n_in, int nl, const float* a0_dev,
                                       const void* spect_pm_dev, const float* dout_dev, int B, int L, const void* state_dev,
                                       size_t state_bytes, float* da0_dev, float* dspect_pm_dev, int accumulate_dspect, void* scratch_dev,
                                       size_t scratch_bytes, void* stream_) {
  if (int rc = check_wn(wts, n_in, nl, B, L)) return rc;
  FACPPG_REQUIRE(gr && a0_dev && spect_pm_dev && dout_dev && state_dev && da0_dev && dspect_pm_dev && scratch_dev, FACPPG_EINVAL, "NULL argument");
  if (int rc = check_wn_grads(gr, nl)) return rc;
  const int Lr = pad_len(L), Lp = HALO + Lr + HALO;
  const StateLayout st = state_layout(nl, B, Lr);
  const ScratchLayout sc = scratch_layout(nl, B, Lr);
  FACPPG_REQUIRE(state_bytes >= st.total, FACPPG_EWORKSPACE, "state has %zu bytes, need %zu", state_bytes, st.total);
  FACPPG_REQUIRE(scratch_bytes >= sc.total, FACPPG_EWORKSPACE, "scratch has %zu bytes, need %zu", scratch_bytes, sc.total);
  hipStream_t s = (hipStream_t)stream_;
  const char* S = (const char*)state_dev;
  char* W = (char*)scratch_dev;
  const int nout = 2 * n_in;
  uint4* condt = (uint4*)(W + sc.condt);
  {
    Packer pk;
    add_backward_images(pk, wts, nl, W + sc.rst, sc.rst_one, W + sc.int_, sc.int_one, condt);
    if (int rc = pk.launch(s)) return rc;
  }
  bf16_t* dskip = (bf16_t*)(W + sc.dskip);
  {
    ZeroRows z;
    z.add(W + sc.dpre, (long)nl * B, (long)Lp * 2 * C * 2, 2 * C * 2, HALO, HALO + L, Lp);
    z.add(W + sc.dh, (long)(nl + 1) * B, (long)Lr * C * 2, C * 2, 0, L, Lr);
    z.add(W + sc.dskip, B, (long)Lr * C * 2, C * 2, 0, L, Lr);
    z.launch(s);
  }
  k_t_end_bwd<<<dim3((L + 31) / 32, B), 256, 0, s>>>(dout_dev, wts->end_w, dskip, nout, L, Lr);
  {  // end conv: weight [nout][256] and bias gradients
    float* part = (float*)(W + sc.part);
    k_small_wgrad_part<false><<<SMALL_PARTS, 256, 0, s>>>(dout_dev, S + st.skip, (long)Lr * C, 0, part, nout, B, L, SMALL_PARTS);
    k_small_wgrad_sum<<<dim3(C / 16, 9), 256, 0, s>>>(part, SMALL_PARTS, nout, gr->end_w, C, 1, nullptr);
    k_small_rowsum<<<nout, 256, 0, s>>>(dout_dev, gr->end_b, nout, B, L);
  }
  WnPacked pw;
  memset(&pw, 0, sizeof(pw));
  pw.rst = (const uint4*)(W + sc.rst); pw.rst_one = sc.rst_one; pw.int_ = (const uint4*)(W + sc.int_); pw.int_one = sc.int_one; pw.condt = condt;
  WnWork wk;
  wk.dpre = W + sc.dpre; wk.dpre_one = sc.dpre_one; wk.dh = W + sc.dh; wk.dh_one = sc.dh_one; wk.dskip = dskip;
  wk.cspart = (float*)(W + sc.cspart); wk.wgpart = (float*)(W + sc.wgpart); wk.wgpart_bytes = sc.wgpart_bytes; wk.wgpart_regions = 1;
  if (int rc = wn_layers_backward_data(nl, pw, S, wk, B, L, s)) return rc;
  if (int rc = wn_layers_backward_dspect(nl, pw, wk, dspect_pm_dev, accumulate_dspect, B, L, s)) return rc;
  const bf16_t* dh0 = (const bf16_t*)(W + sc.dh);
  k_t_start_bwd<<<dim3((L + 31) / 32, B), 256, 0, s>>>(dh0, wts->start_w, da0_dev, n_in, L, Lr);
  {  // start conv: weight [256][n_in] and bias [256] gradients
    float* part = (float*)(W + sc.part);
    k_small_wgrad_part<true><<<SMALL_PARTS, 256, 0, s>>>(a0_dev, dh0, (long)Lr * C, 0, part, n_in, B, L, SMALL_PARTS);
    k_small_wgrad_sum<<<dim3(C / 16, 9), 256, 0, s>>>(part, SMALL_PARTS, n_in, gr->start_w, 1, n_in, gr->start_b);
  }
  return wn_layers_backward_weights(gr, nl, S, wk, spect_pm_dev, B, L, s, s);
}

// ================================================================================================================
// The whole model's training direction (glow.py:208-250) on the bf16 stack: GROUPS of consecutive flows per call.
//
// Between two stacks everything is per-position arithmetic on <= 8 channels: flow k-1's end conv + affine coupling
// (glow.py:175, 240-245), the early-output split (glow.py:231-233), flow k's 1x1 mixing conv (glow.py:98-102) and the start
// conv of its stack (glow.py:156).  Until round 5 that was ~13 launches per flow forward and ~25 backward (HIP and torch
// kernels of 5-15 us each: a third of the step at batch 12, most of it at batch 3).  k_edge_fwd / k_edge_bwd do one flow
// boundary per launch -- tail of the flow below, head of the flow above -- and the backward one also leaves per-workgroup
// partial sums of the <= 8-channel weight / bias gradients (end conv, start conv, mixing matrix), which one k_edge_sum
// launch per group adds up in a fixed order.  The packed bf16 weight images, the zeroed margins of every flow's saved
// state and the summed gate biases are prepared ONCE per step for all flows (facppg_glow_bf16_begin).
// ================================================================================================================
namespace {

constexpr int MAXGF = FACPPG_GLOW_MAX_FLOWS;
constexpr int EPART = FACPPG_GLOW_PART_FLOATS;          // end_w [8][256] | start_w^T [4][256] | start_b [256] | dW [8][8] | end_b [8]
constexpr int EP_START = 8 * C, EP_W = EP_START + 5 * C, EP_EB = EP_W + 64;
static_assert(EPART == EP_EB + 8, "FACPPG_GLOW_PART_FLOATS");

struct EdgeFlow {
  const float* conv_w;                            // [c][c]
  const float* start_w; const float* start_b;     // [256][c/2], [256]
  const float* end_w; const float* end_b;         // [c][256], [c]
  float* u; float* z; float* wn; float* dzp;      // [B][c][L]: conv input, conv output, stack output (b | log_s), backward temp (dy0 | dx1)
  float* early; long early_bs;                    // [B][early_n][L] with batch stride early_bs: early output (forward) / its gradient (backward)
  float* part;                                    // [nparts][EPART]
  bf16_t* h0;                                     // [B][Lp][256]: h_0 of the stack
  const float* skip;                              // [B][Lr][256] fp32: the stack's skip sum
  const float* dlogs; long dl_b, dl_j, dl_n;      // gradient of log_s (element strides; NULL: none)
  int c, early_n;
};
struct EdgeFwdArgs { EdgeFlow lo, hi; const float* audio_in; float* audio_out; long in_bs, out_bs; int has_lo, has_hi, L, Lr, Lp; };   // (batch strides in floats)
struct EdgeBwdArgs {
  EdgeFlow lo, hi;
  const bf16_t* dh0;      // [B][Lr][256]: gradient w.r.t. hi's h_0 (its stack's backward has run)
  bf16_t* dskip;          // [B][Lr][256]: gradient w.r.t. lo's skip sum (its stack's backward runs next)
  const float* d_out;     // !has_hi: gradient of the group's output audio [B][lo.c][L], batch stride d_out_bs
  float* d_in;            // !has_lo: gradient of the group's input audio [B][hi.c + hi.early_n][L], batch stride d_in_bs
  long d_out_bs, d_in_bs;
  int has_lo, has_hi, L, Lr, nchunk;
};

// lo: end conv of the stack that just ran + affine coupling; hi: early split + mixing conv + start conv of the next stack.
// A workgroup takes 32 positions of one batch item.  The <= 8-channel tensors are [B][c][L]: thread (j = tid / 32, r = tid % 32) moves
// channel j of position r between HBM and LDS (32 consecutive floats per channel), the per-position arithmetic -- one thread per
// position -- reads and writes LDS only.
__global__ __launch_bounds__(256) void k_edge_fwd(EdgeFwdArgs a) {
  __shared__ __attribute__((aligned(16))) float sw[C * 8];
  __shared__ float s_wn[32][9], s_y[32][9], s_in[32][9], s_z[32][9];
  const int tid = threadIdx.x, b = blockIdx.y, n0 = blockIdx.x * 32, L = a.L;
  const int cj = tid >> 5, cr = tid & 31, cn = n0 + cr;      // the cooperative (channel, position) of this thread
  const int cl = a.has_lo ? a.lo.c : 0, hl = cl >> 1;
  const int ch = a.has_hi ? a.hi.c : 0, hh = ch >> 1, e = a.has_hi ? a.hi.early_n : 0;
  const int cy = a.has_lo ? cl : ch + e;
  if (cn < L && cj < cy) s_in[cr][cj] = a.has_lo ? a.lo.z[((size_t)b * cl + cj) * L + cn] : a.audio_in[(size_t)b * a.in_bs + (size_t)cj * L + cn];
  if (a.has_lo) {
    // end conv (glow.py:175) as in k_t_end: a wave takes 8 positions, lane = (position, slice of 32 channels)
    for (int i = tid; i < C * 8; i += 256) sw[i] = (i & 7) < cl ? a.lo.end_w[(i & 7) * C + (i >> 3)] : 0.0f;
    __syncthreads();
    const int lane = tid & 63, ps = lane >> 3, sl = lane & 7, r = (tid >> 6) * 8 + ps, n = n0 + r;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (n < L) {
      const float4* row = reinterpret_cast<const float4*>(a.lo.skip + ((size_t)b * a.Lr + n) * C + 32 * sl);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 x = row[q];
        const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float4 w0 = *reinterpret_cast<const float4*>(sw + (32 * sl + 4 * q + t) * 8), w1 = *reinterpret_cast<const float4*>(sw + (32 * sl + 4 * q + t) * 8 + 4);
          v[0] = fmaf(w0.x, xv[t], v[0]); v[1] = fmaf(w0.y, xv[t], v[1]); v[2] = fmaf(w0.z, xv[t], v[2]); v[3] = fmaf(w0.w, xv[t], v[3]);
          v[4] = fmaf(w1.x, xv[t], v[4]); v[5] = fmaf(w1.y, xv[t], v[5]); v[6] = fmaf(w1.z, xv[t], v[6]); v[7] = fmaf(w1.w, xv[t], v[7]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int off = 1; off < 8; off <<= 1) v[j] += __shfl_xor(v[j], off, 64);
    if (sl == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) s_wn[r][j] = j < cl ? v[j] + a.lo.end_b[j] : 0.0f;
    }
  }
  __syncthreads();
  if (tid < 32 && n0 + tid < L) {
    float* y = s_y[tid];
    if (a.has_lo) {       // affine coupling (glow.py:240-245): y = [x0 | exp(log_s) * x1 + b], (b | log_s) = the stack's output
      for (int j = 0; j < hl; ++j) {
        y[j] = s_in[tid][j];
        y[hl + j] = fmaf(expf(s_wn[tid][hl + j]), s_in[tid][hl + j], s_wn[tid][j]);
      }
    } else {
      for (int j = 0; j < cy; ++j) y[j] = s_in[tid][j];
    }
    if (a.has_hi)         // mixing conv (glow.py:98-102) of what the early split (glow.py:231-233) leaves; a0 = the first half of its output
      for (int i = 0; i < ch; ++i) {
        float zv = 0.0f;
        for (int j = 0; j < ch; ++j) zv = fmaf(a.hi.conv_w[i * ch + j], y[e + j], zv);
        s_z[tid][i] = zv;
      }
  }
  __syncthreads();
  if (cn < L) {           // cooperative stores: lo's stack output, hi's early output / conv input / conv output, or the group's output
    if (a.has_lo && cj < cl) a.lo.wn[((size_t)b * cl + cj) * L + cn] = s_wn[cr][cj];
    if (a.has_hi) {
      if (cj < e) a.hi.early[(size_t)b * a.hi.early_bs + (size_t)cj * L + cn] = s_y[cr][cj];
      if (cj < ch) {
        a.hi.u[((size_t)b * ch + cj) * L + cn] = s_y[cr][e + cj];
        a.hi.z[((size_t)b * ch + cj) * L + cn] = s_z[cr][cj];
      }
    } else if (cj < cy) {
      a.audio_out[(size_t)b * a.out_bs + (size_t)cj * L + cn] = s_y[cr][cj];
    }
  }
  if (!a.has_hi) return;
  {   // start conv (glow.py:156) as in k_t_start: a thread owns 4 channels and walks 8 positions
    const int c4 = (tid & 63) * 4, r0 = (tid >> 6) * 8;
    float wr[4][4], bv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      bv[t] = a.hi.start_b[c4 + t];
#pragma unroll
      for (int j = 0; j < 4; ++j) wr[t][j] = j < hh ? a.hi.start_w[(c4 + t) * hh + j] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int n = n0 + r0 + i;
      if (n >= L) break;
      float v[4] = {bv[0], bv[1], bv[2], bv[3]};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < hh) {
          const float av = s_z[r0 + i][j];
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = fmaf(wr[t][j], av, v[t]);
        }
      *reinterpret_cast<uint2*>(a.hi.h0 + ((size_t)b * a.Lp + HALO + n) * C + c4) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
    }
  }
}

// Backward of one flow boundary.  hi: the flow whose stack's backward has just produced dh_0 -- start conv backward (da0), the
// mixing conv's backward (du = W^T dz) and the early split's; lo: the flow below -- affine coupling backward and the end conv's
// (dskip), after which lo's stack can run backward.  Also per-workgroup partial sums (fixed order, no atomics) of hi's start conv
// weight / bias and mixing-matrix gradients and of lo's end conv weight / bias gradients.  A workgroup walks `nchunk` chunks of 32
// positions of one batch item; HBM <-> LDS traffic of the <= 8-channel tensors is cooperative as in k_edge_fwd.
__global__ __launch_bounds__(256) void k_edge_bwd(EdgeBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float sw_s[C * 4];
  __shared__ float s_da0[32][5], s_a0[32][5], s_dwn[32][9], s_v[32][9], s_dz[32][9];
  __shared__ float s_u[32][9], s_dzph[32][9], s_zl[32][9], s_wnl[32][9], s_dl[32][5], s_out[32][9];
  __shared__ float red[3][13][C];
  const int tid = threadIdx.x, b = blockIdx.y, L = a.L;
  const int c4 = (tid & 63) * 4, rl = tid >> 6;
  const int cj = tid >> 5, cr = tid & 31;
  const size_t wg = (size_t)b * gridDim.x + blockIdx.x;
  const int chh = a.has_hi ? a.hi.c : 0, hh = chh >> 1, eh = a.has_hi ? a.hi.early_n : 0;
  const int cl = a.has_lo ? a.lo.c : 0, hl = cl >> 1;
  const int cy = a.has_hi ? chh + eh : cl;
  for (int i = tid; i < C * 4; i += 256) sw_s[i] = (i & 3) < hh ? a.hi.start_w[(i >> 2) * hh + (i & 3)] : 0.0f;
  for (int i = tid; i < 32 * 9; i += 256) (&s_dwn[0][0])[i] = 0.0f;
  for (int i = tid; i < 32 * 5; i += 256) (&s_a0[0][0])[i] = 0.0f;
  float we[8][4];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int t = 0; t < 4; ++t) we[j][t] = j < cl ? a.lo.end_w[j * C + c4 + t] : 0.0f;
  float acc_e[8][4], acc_s[5][4], accW[64], acc_eb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    acc_eb[j] = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc_e[j][t] = 0.0f;
  }
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc_s[j][t] = 0.0f;
#pragma unroll
  for (int k = 0; k < 64; ++k) accW[k] = 0.0f;
  __syncthreads();
  for (int ck = 0; ck < a.nchunk; ++ck) {
    const int n0 = (blockIdx.x * a.nchunk + ck) * 32;
    if (n0 >= L) break;
    const int cn = n0 + cr;
    if (cn < L) {   // cooperative loads of the <= 8-channel tensors of this chunk
      if (a.has_hi) {
        if (cj < chh) {
          s_u[cr][cj] = a.hi.u[((size_t)b * chh + cj) * L + cn];
          s_dzph[cr][cj] = a.hi.dzp[((size_t)b * chh + cj) * L + cn];
          if (cj < hh) s_a0[cr][cj] = a.hi.z[((size_t)b * chh + cj) * L + cn];
        }
        if (cj < eh) s_v[cr][cj] = a.hi.early[(size_t)b * a.hi.early_bs + (size_t)cj * L + cn];
      } else if (cj < cl) {
        s_v[cr][cj] = a.d_out[(size_t)b * a.d_out_bs + (size_t)cj * L + cn];
      }
      if (a.has_lo && cj < cl) {
        s_zl[cr][cj] = a.lo.z[((size_t)b * cl + cj) * L + cn];
        s_wnl[cr][cj] = a.lo.wn[((size_t)b * cl + cj) * L + cn];
        if (cj < hl) s_dl[cr][cj] = a.lo.dlogs ? a.lo.dlogs[(size_t)b * a.lo.dl_b + (size_t)cj * a.lo.dl_j + (size_t)cn * a.lo.dl_n] : 0.0f;
      }
    }
    if (a.has_hi) {   // da0[j] = sum_c Ws[c][j] dh0[c] as in k_t_start_bwd: a wave takes 8 positions, lane = (position, 32 channels)
      const int lane = tid & 63, ps = lane >> 3, sl = lane & 7, r = (tid >> 6) * 8 + ps, n = n0 + r;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (n < L) {
        const uint4* row = reinterpret_cast<const uint4*>(a.dh0 + ((size_t)b * a.Lr + n) * C + 32 * sl);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint4 d = row[q];
          const float dv[8] = {lo2f(d.x), hi2f(d.x), lo2f(d.y), hi2f(d.y), lo2f(d.z), hi2f(d.z), lo2f(d.w), hi2f(d.w)};
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const float4 ww = *reinterpret_cast<const float4*>(sw_s + (32 * sl + 8 * q + t) * 4);
            v[0] = fmaf(ww.x, dv[t], v[0]); v[1] = fmaf(ww.y, dv[t], v[1]); v[2] = fmaf(ww.z, dv[t], v[2]); v[3] = fmaf(ww.w, dv[t], v[3]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) v[j] += __shfl_xor(v[j], off, 64);
      if (sl == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s_da0[r][j] = v[j];
      }
    }
    __syncthreads();
    if (tid < 32 && n0 + tid < L) {   // one thread per position, LDS operands only
      float* dy = s_v[tid];
      if (a.has_hi) {
        float* dz = s_dz[tid];
        for (int j = 0; j < hh; ++j) {
          dz[j] = s_dzph[tid][j] + s_da0[tid][j];
          dz[hh + j] = s_dzph[tid][hh + j];
        }
        for (int j = chh; j < 8; ++j) dz[j] = 0.0f;
        float uu[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) uu[j] = j < chh ? s_u[tid][j] : 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {   // dW[i][j] += dz[i] u[j]
          const float dzi = dz[i];
#pragma unroll
          for (int j = 0; j < 8; ++j) accW[i * 8 + j] = fmaf(dzi, uu[j], accW[i * 8 + j]);
        }
        for (int j = 0; j < chh; ++j) {  // du[j] = sum_i W[i][j] dz[i]
          float v = 0.0f;
          for (int i = 0; i < chh; ++i) v = fmaf(a.hi.conv_w[i * chh + j], dz[i], v);
          dy[eh + j] = v;
        }
      }
      if (a.has_lo) {   // y = [x0 | exp(log_s) x1 + b]: d b = dy1, d log_s = dy1 x1 exp(log_s) (+ the loss's), dx1 = dy1 exp(log_s), dx0 = dy0 (+ da0 later)
        for (int j = 0; j < hl; ++j) {
          const float x1 = s_zl[tid][hl + j], ev = expf(s_wnl[tid][hl + j]);
          const float d1 = dy[hl + j];
          s_dwn[tid][j] = d1;
          s_dwn[tid][hl + j] = d1 * x1 * ev + s_dl[tid][j];
          s_out[tid][j] = dy[j];
          s_out[tid][hl + j] = d1 * ev;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc_eb[j] += j < cl ? s_dwn[tid][j] : 0.0f;
      } else {
        for (int j = 0; j < cy; ++j) s_out[tid][j] = dy[j];
      }
    }
    __syncthreads();
    if (cn < L) {   // cooperative stores: lo's (dy0 | dx1), or the gradient of the group's input
      if (a.has_lo) {
        if (cj < cl) a.lo.dzp[((size_t)b * cl + cj) * L + cn] = s_out[cr][cj];
      } else if (cj < cy) {
        a.d_in[(size_t)b * a.d_in_bs + (size_t)cj * L + cn] = s_out[cr][cj];
      }
    }
    {   // a thread owns 4 of the 256 channels and walks 8 positions; every row's loads are issued before the first FMA (rows >= L read
        // row L - 1 and contribute nothing)
      float4 sk[8];
      uint2 dd[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int n = min(n0 + rl * 8 + i, L - 1);
        if (a.has_lo) sk[i] = *reinterpret_cast<const float4*>(a.lo.skip + ((size_t)b * a.Lr + n) * C + c4);
        if (a.has_hi) dd[i] = *reinterpret_cast<const uint2*>(a.dh0 + ((size_t)b * a.Lr + n) * C + c4);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = rl * 8 + i, n = n0 + r;
        const bool live = n < L;
        if (a.has_lo) {   // dskip = We^T dwn (bf16); end conv weight gradient += dwn skip^T
          float dwn[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) dwn[j] = live ? s_dwn[r][j] : 0.0f;
          float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = fmaf(we[j][t], dwn[j], v[t]);
          if (live) *reinterpret_cast<uint2*>(a.dskip + ((size_t)b * a.Lr + n) * C + c4) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
          const float sv[4] = {sk[i].x, sk[i].y, sk[i].z, sk[i].w};
#pragma unroll
          for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc_e[j][t] = fmaf(dwn[j], sv[t], acc_e[j][t]);
        }
        if (a.has_hi) {   // start conv weight gradient += dh0 a0^T, bias gradient += dh0
          const float dv[4] = {live ? lo2f(dd[i].x) : 0.0f, live ? hi2f(dd[i].x) : 0.0f, live ? lo2f(dd[i].y) : 0.0f, live ? hi2f(dd[i].y) : 0.0f};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float av = s_a0[r][j];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc_s[j][t] = fmaf(av, dv[t], acc_s[j][t]);
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) acc_s[4][t] += dv[t];
        }
      }
    }
    __syncthreads();
  }
  // the four row lanes meet in lane order
  if (rl) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) red[rl - 1][j][c4 + t] = acc_e[j][t];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) red[rl - 1][8 + j][c4 + t] = acc_s[j][t];
  }
  __syncthreads();
  if (rl == 0) {
#pragma unroll
    for (int j = 0; j < 13; ++j) {
      float4 v = j < 8 ? make_float4(acc_e[j][0], acc_e[j][1], acc_e[j][2], acc_e[j][3])
                       : make_float4(acc_s[j - 8 < 0 ? 0 : j - 8][0], acc_s[j - 8 < 0 ? 0 : j - 8][1], acc_s[j - 8 < 0 ? 0 : j - 8][2], acc_s[j - 8 < 0 ? 0 : j - 8][3]);
#pragma unroll
      for (int l = 0; l < 3; ++l) { v.x += red[l][j][c4]; v.y += red[l][j][c4 + 1]; v.z += red[l][j][c4 + 2]; v.w += red[l][j][c4 + 3]; }
      if (j < 8) { if (a.has_lo) *reinterpret_cast<float4*>(a.lo.part + wg * EPART + j * C + c4) = v; }
      else if (a.has_hi) *reinterpret_cast<float4*>(a.hi.part + wg * EPART + EP_START + (j - 8) * C + c4) = v;
    }
  }
  __syncthreads();
  float* sW = &red[0][0][0];     // [32][65] mixing-matrix sums | [32][9] end-bias sums of the position threads
  if (tid < 32) {
#pragma unroll
    for (int k = 0; k < 64; ++k) sW[tid * 65 + k] = accW[k];
#pragma unroll
    for (int j = 0; j < 8; ++j) sW[32 * 65 + tid * 9 + j] = acc_eb[j];
  }
  __syncthreads();
  if (tid < 64 && a.has_hi) {
    float v = 0.0f;
    for (int t = 0; t < 32; ++t) v += sW[t * 65 + tid];
    a.hi.part[wg * EPART + EP_W + tid] = v;
  } else if (tid >= 64 && tid < 72 && a.has_lo) {
    float v = 0.0f;
    for (int t = 0; t < 32; ++t) v += sW[32 * 65 + t * 9 + (tid - 64)];
    a.lo.part[wg * EPART + EP_EB + (tid - 64)] = v;
  }
}

// Fixed-order sums of the per-workgroup partials of a group's flows -> the <= 8-channel gradients; the mixing matrix also gets its
// log-determinant term g * B * L * W^-T (glow.py:100).  Workgroup = 16 outputs x 16 groups of partials.
struct EdgeSumFlow {
  const float* part; float* end_w; float* end_b; float* start_w; float* start_b; float* d_conv_w;
  const float* winv_t; const float* g_logdet; float ld_scale; int c;
};
struct EdgeSumArgs { EdgeSumFlow f[MAXGF]; int nparts; };
__global__ __launch_bounds__(256) void k_edge_sum(EdgeSumArgs a) {
  const EdgeSumFlow& f = a.f[blockIdx.y];
  __shared__ float red[16][16];
  const int el = threadIdx.x & 15, q = threadIdx.x >> 4, e = blockIdx.x * 16 + el;
  const int p0 = a.nparts * q / 16, p1 = a.nparts * (q + 1) / 16;
  float v4[4] = {0.f, 0.f, 0.f, 0.f};
  if (e < EPART)
    for (int p = p0; p < p1; p += 4)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (p + u < p1) v4[u] += f.part[(size_t)(p + u) * EPART + e];
  red[q][el] = (v4[0] + v4[1]) + (v4[2] + v4[3]);
  __syncthreads();
  if (q || e >= EPART) return;
  float v = red[0][el];
#pragma unroll
  for (int k = 1; k < 16; ++k) v += red[k][el];
  const int c = f.c, h = c >> 1;
  if (e < EP_START) {
    const int j = e >> 8, cc = e & 255;
    if (j < c) f.end_w[j * C + cc] = v;
  } else if (e < EP_W) {
    const int j = (e - EP_START) >> 8, cc = (e - EP_START) & 255;
    if (j < h) f.start_w[cc * h + j] = v;
    else if (j == 4) f.start_b[cc] = v;
  } else if (e < EP_EB) {
    const int i = (e - EP_W) >> 3, j = (e - EP_W) & 7;
    if (i < c && j < c) f.d_conv_w[i * c + j] = v + (f.g_logdet ? f.g_logdet[0] * f.ld_scale * f.winv_t[i * c + j] : 0.0f);
  } else if (e - EP_EB < c) {
    f.end_b[e - EP_EB] = v;
  }
}

// out[0] = scale * log|det W| (glow.py:100: log_det_W = B * L * logdet(W)), out[1..] = W^-T
struct LogdetBatch { const float* w[MAXGF]; float* out[MAXGF]; int c[MAXGF]; float scale[MAXGF]; };
__global__ __launch_bounds__(64) void k_logdet_batch(LogdetBatch lb) {
  logdet_wave(lb.w[blockIdx.x], lb.c[blockIdx.x], lb.out[blockIdx.x], lb.out[blockIdx.x] + 1);
  if (threadIdx.x == 0) lb.out[blockIdx.x][0] *= lb.scale[blockIdx.x];
}

// what one flow's stack keeps packed for a whole step: both directions' bf16 operand images and the summed gate biases
struct PackedLayout { size_t w1, w2, rst, int_, condt, b1, total; size_t w1_one, w2_one, rst_one, int_one; };
PackedLayout packed_layout(int nl) {
  PackedLayout s;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  s.w1_one = (size_t)16 * (K1 / 16) * 64 * 16; s.w2_one = (size_t)16 * (C / 16) * 64 * 16;
  s.rst_one = (size_t)8 * (2 * C / 16) * 64 * 16; s.int_one = (size_t)8 * (3 * 2 * C / 16) * 64 * 16;
  s.w1 = take(s.w1_one * nl); s.w2 = take(s.w2_one * nl); s.rst = take(s.rst_one * nl); s.int_ = take(s.int_one * nl);
  s.condt = take((size_t)(NCOND / 32) * (nl * 2 * C / 16) * 64 * 16);
  s.b1 = take((size_t)nl * 2 * C * 4);
  s.total = off;
  return s;
}
// gradients in flight of ONE flow's backward (the flows of a step take turns on it) + the reductions' partial buffers
struct WorkLayout { size_t dpre, dh, dskip, cspart, wgpart, wgpart_bytes, total; size_t dpre_one, dh_one; };
WorkLayout work_layout(int nl, int B, int Lr) {
  WorkLayout s;
  const int Lp = HALO + Lr + HALO;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  s.dpre_one = (size_t)B * Lp * 2 * C * 2; s.dh_one = (size_t)B * Lr * C * 2;
  s.dpre = take(s.dpre_one * nl); s.dh = take(s.dh_one * (nl + 1)); s.dskip = take(s.dh_one);
  s.cspart = take((size_t)MAXCS * CS_SLICES * 1024 * 4);
  s.wgpart_bytes = (size_t)3 * nl * 5 * (2 * C) * C * 4;      // per weight-gradient launch of a flow (three regions: one ordered reduce)
  s.wgpart = take(s.wgpart_bytes * 3);
  s.total = off;
  return s;
}
// When the bf16 operand images of a flow's stack are formed.  Measured (round 6): all flows at the start of the step (one launch per
// flow, 12 instead of 24) leaves every layer launch fetching its 2 - 3 MB of weights from HBM milliseconds after they were written
// (k_wn_fwd 38.6 -> 42.7 us, k_wn_bwd 30.1 -> 32.4 at batch 12; 18.1 -> 22.6 at batch 3: +0.5 ms per step); packed right in front of
// the flow's stack, per direction, they are still in the L2s / the Infinity Cache when the layers read them.  FACPPG_TRAIN_PACK=step: the former.
bool pack_per_step() {
  const char* e = getenv("FACPPG_TRAIN_PACK");
  return e && !strcmp(e, "step");
}
int edge_chunks(int B, int L) {   // chunks of 32 positions a backward edge workgroup walks: at most ~512 workgroups (two per CU)
  const int nblk = (L + 31) / 32;
  const char* e = getenv("FACPPG_EDGE_WGS");
  const int target = e && atoi(e) > 0 ? atoi(e) : 512;
  return std::max(1, (B * nblk + target - 1) / target);
}
WnPacked packed_view(const char* P, const PackedLayout& pl) {
  WnPacked pw;
  pw.w1 = (const uint4*)(P + pl.w1); pw.w1_one = pl.w1_one; pw.w2 = (const uint4*)(P + pl.w2); pw.w2_one = pl.w2_one;
  pw.rst = (const uint4*)(P + pl.rst); pw.rst_one = pl.rst_one; pw.int_ = (const uint4*)(P + pl.int_); pw.int_one = pl.int_one;
  pw.condt = (const uint4*)(P + pl.condt); pw.b1 = (const float*)(P + pl.b1);
  return pw;
}
int check_glow_flows(const facppg_glow_flow* f, int n, int nl, int B, int L, bool backward) {
  FACPPG_REQUIRE(f && n >= 1 && n <= MAXGF, FACPPG_EINVAL, "1..%d flows per group call (got %d)", MAXGF, n);
  for (int k = 0; k < n; ++k) {
    FACPPG_REQUIRE(f[k].c >= 2 && f[k].c <= 8 && f[k].c % 2 == 0 && f[k].early >= 0 && f[k].c + f[k].early <= 8, FACPPG_EUNSUPPORTED,
                   "flow %d: %d channels after an early split of %d (even, <= 8 in all)", k, f[k].c, f[k].early);
    if (int rc = check_wn(f[k].w, f[k].c / 2, nl, B, L)) return rc;
    FACPPG_REQUIRE(f[k].conv_w && f[k].logdet && f[k].u && f[k].z && f[k].wn_out && f[k].packed && f[k].state && (f[k].early == 0 || f[k].early_io),
                   FACPPG_EINVAL, "flow %d: NULL buffer", k);
    if (k) FACPPG_REQUIRE(f[k].c + f[k].early == f[k - 1].c, FACPPG_EINVAL, "flow %d takes %d + %d channels, flow %d leaves %d", k, f[k].c,
                          f[k].early, k - 1, f[k - 1].c);
    if (backward) {
      if (int rc = check_wn_grads(f[k].g, nl)) return rc;
      FACPPG_REQUIRE(f[k].d_conv_w && f[k].dzp && f[k].part, FACPPG_EINVAL, "flow %d: NULL gradient buffer", k);
    }
  }
  return FACPPG_OK;
}
EdgeFlow edge_flow(const facppg_glow_flow& f, int nl, int B, int Lr) {
  const StateLayout st = state_layout(nl, B, Lr);
  EdgeFlow e;
  memset(&e, 0, sizeof(e));
  e.conv_w = f.conv_w; e.start_w = f.w->start_w; e.start_b = f.w->start_b; e.end_w = f.w->end_w; e.end_b = f.w->end_b;
  e.u = f.u; e.z = f.z; e.wn = f.wn_out; e.dzp = f.dzp; e.early = f.early_io; e.early_bs = f.early_bs; e.part = f.part;
  e.h0 = (bf16_t*)((char*)f.state + st.h); e.skip = (const float*)((const char*)f.state + st.skip);
  e.dlogs = f.dlog_s; e.dl_b = f.dls_b; e.dl_j = f.dls_j; e.dl_n = f.dls_n;
  e.c = f.c; e.early_n = f.early;
  return e;
}

}  // namespace

extern "C" int facppg_glow_bf16_layout(int n_layers, int B, int L, facppg_glow_bf16_sizes* out) {
  FACPPG_REQUIRE(out && n_layers >= 1 && n_layers <= 8 && B > 0 && B <= 65535 && L > 0, FACPPG_EINVAL, "bad argument");
  const int Lr = pad_len(L);
  out->packed_bytes_per_flow = packed_layout(n_layers).total;
  out->state_bytes_per_flow = (state_layout(n_layers, B, Lr).total + 255) / 256 * 256;
  out->work_bytes = work_layout(n_layers, B, Lr).total;
  const int nchunk = edge_chunks(B, L);
  out->n_parts = B * (((L + 31) / 32 + nchunk - 1) / nchunk);
  out->part_floats = EPART;
  return FACPPG_OK;
}

// Once per step, for ALL flows: both directions' bf16 operand images of every stack, the summed gate biases, and the zero rows of
// every flow's saved state (the conv padding margins, the rows between L and the padded length) and of the backward's buffers.
extern "C" int facppg_glow_bf16_begin(const facppg_wn_weights* wts, int n_flows, int nl, int B, int L, void* packed_dev, void* states_dev,
                                      void* work_dev, void* stream_) {
  FACPPG_REQUIRE(wts && packed_dev && states_dev && work_dev && n_flows >= 1 && n_flows <= 64, FACPPG_EINVAL, "bad argument");
  FACPPG_REQUIRE((long)n_flows * (nl + 1) * B <= 65535, FACPPG_EUNSUPPORTED, "%d flows x %d layers x batch %d: more state images than one launch zeroes", n_flows, nl, B);
  for (int k = 0; k < n_flows; ++k)
    if (int rc = check_wn(&wts[k], 1, nl, B, L)) return rc;
  hipStream_t s = (hipStream_t)stream_;
  const int Lr = pad_len(L), Lp = HALO + Lr + HALO;
  const PackedLayout pl = packed_layout(nl);
  const StateLayout st = state_layout(nl, B, Lr);
  const size_t st_stride = (st.total + 255) / 256 * 256;
  const WorkLayout wl = work_layout(nl, B, Lr);
  if (pack_per_step())
    for (int k = 0; k < n_flows; ++k) {
      char* P = (char*)packed_dev + pl.total * k;
      Packer pk;
      add_forward_images(pk, &wts[k], nl, P + pl.w1, pl.w1_one, P + pl.w2, pl.w2_one);
      add_backward_images(pk, &wts[k], nl, P + pl.rst, pl.rst_one, P + pl.int_, pl.int_one, (uint4*)(P + pl.condt));
      if (int rc = pk.launch(s)) return rc;
    }
  for (int k0 = 0; k0 < n_flows; k0 += ADD_FLOWS) {
    AddBatchN ab;
    const int nf = std::min(ADD_FLOWS, n_flows - k0);
    for (int k = 0; k < nf; ++k)
      for (int i = 0; i < nl; ++i) { ab.a[k * 8 + i] = wts[k0 + k].in_b[i]; ab.b[k * 8 + i] = wts[k0 + k].cond_b[i]; }
    ab.out = (float*)((char*)packed_dev + pl.total * k0 + pl.b1); ab.out_stride = (long)(pl.total / 4);
    k_add2_flows<<<dim3(2, nl, nf), 256, 0, s>>>(ab, 2 * C);
  }
  {
    char* S = (char*)states_dev;
    ZeroRows z;
    z.add(S + st.h, (long)(nl + 1) * B, (long)Lp * C * 2, C * 2, HALO, HALO + L, Lp, n_flows, (long)st_stride);
    z.add(S + st.ts, (long)nl * B, (long)Lr * 2 * C * 2, 2 * C * 2, 0, L, Lr, n_flows, (long)st_stride);
    z.add(S + st.acts, (long)nl * B, (long)Lr * C * 2, C * 2, 0, L, Lr, n_flows, (long)st_stride);
    z.launch(s);
    char* W = (char*)work_dev;
    ZeroRows zb;
    zb.add(W + wl.dpre, (long)nl * B, (long)Lp * 2 * C * 2, 2 * C * 2, HALO, HALO + L, Lp);
    zb.add(W + wl.dh, (long)(nl + 1) * B, (long)Lr * C * 2, C * 2, 0, L, Lr);
    zb.add(W + wl.dskip, B, (long)Lr * C * 2, C * 2, 0, L, Lr);
    zb.launch(s);
  }
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

// Forward of n consecutive flows (glow.py:228-247): audio_in [B][flows[0].c + flows[0].early][L] -> audio_out [B][flows[n-1].c][L];
// per flow its early output (if it splits one off), u / z / wn_out (kept for the backward; log_s = wn_out[:, c/2:]) and
// logdet[0] = log|det W|, logdet[1..] = W^-T.
extern "C" int facppg_glow_bf16_group_forward(const facppg_glow_flow* flows, int n, int nl, const float* audio_in_dev, long in_bs,
                                              float* audio_out_dev, long out_bs, const void* spect_pm_dev, int B, int L, void* stream_) {
  if (int rc = check_glow_flows(flows, n, nl, B, L, false)) return rc;
  FACPPG_REQUIRE(audio_in_dev && audio_out_dev && spect_pm_dev, FACPPG_EINVAL, "NULL argument");
  hipStream_t s = (hipStream_t)stream_;
  const int Lr = pad_len(L), Lp = HALO + Lr + HALO;
  const PackedLayout pl = packed_layout(nl);
  {
    LogdetBatch lb;
    for (int k = 0; k < n; ++k) { lb.w[k] = flows[k].conv_w; lb.out[k] = flows[k].logdet; lb.c[k] = flows[k].c; lb.scale[k] = flows[k].ld_scale; }
    k_logdet_batch<<<n, 64, 0, s>>>(lb);
  }
  for (int k = 0; k <= n; ++k) {
    EdgeFwdArgs a;
    memset(&a, 0, sizeof(a));
    a.has_lo = k > 0; a.has_hi = k < n; a.L = L; a.Lr = Lr; a.Lp = Lp;
    if (a.has_lo) a.lo = edge_flow(flows[k - 1], nl, B, Lr);
    if (a.has_hi) a.hi = edge_flow(flows[k], nl, B, Lr);
    a.audio_in = audio_in_dev; a.audio_out = audio_out_dev; a.in_bs = in_bs; a.out_bs = out_bs;
    k_edge_fwd<<<dim3((L + 31) / 32, B), 256, 0, s>>>(a);
    if (k < n) {
      if (!pack_per_step()) {
        char* P = (char*)flows[k].packed;
        Packer pk;
        add_forward_images(pk, flows[k].w, nl, P + pl.w1, pl.w1_one, P + pl.w2, pl.w2_one);
        if (int rc = pk.launch(s)) return rc;
      }
      if (int rc = wn_layers_forward(flows[k].w, nl, packed_view((const char*)flows[k].packed, pl), spect_pm_dev, B, L, (char*)flows[k].state, s))
        return rc;
    }
  }
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

// Backward of the same n flows: d_audio_out -> d_audio_in, every weight / bias gradient of the stacks (flows[k].g), the mixing
// matrices' gradients incl. the log-determinant term (d_conv_w), the conditioning gradient added to dspect_pm (overwritten by the
// first flow that runs when !accumulate_dspect).  flows[k].early_io = gradient of the early output, dlog_s = gradient of log_s.
extern "C" int facppg_glow_bf16_group_backward(const facppg_glow_flow* flows, int n, int nl, const float* d_audio_out_dev, long d_out_bs,
                                               float* d_audio_in_dev, long d_in_bs, const void* spect_pm_dev, float* dspect_pm_dev, int accumulate_dspect, void* work_dev, int B,
                                               int L, void* stream_) {
  if (int rc = check_glow_flows(flows, n, nl, B, L, true)) return rc;
  FACPPG_REQUIRE(d_audio_out_dev && d_audio_in_dev && spect_pm_dev && dspect_pm_dev && work_dev, FACPPG_EINVAL, "NULL argument");
  hipStream_t s = (hipStream_t)stream_;
  const int Lr = pad_len(L);
  const PackedLayout pl = packed_layout(nl);
  const WorkLayout wl = work_layout(nl, B, Lr);
  char* W = (char*)work_dev;
  WnWork wk;
  wk.dpre = W + wl.dpre; wk.dpre_one = wl.dpre_one; wk.dh = W + wl.dh; wk.dh_one = wl.dh_one; wk.dskip = (bf16_t*)(W + wl.dskip);
  wk.cspart = (float*)(W + wl.cspart); wk.wgpart = (float*)(W + wl.wgpart); wk.wgpart_bytes = wl.wgpart_bytes; wk.wgpart_regions = 3;
  const int nchunk = edge_chunks(B, L), gx = ((L + 31) / 32 + nchunk - 1) / nchunk;
  for (int k = n; k >= 0; --k) {
    EdgeBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.has_hi = k < n; a.has_lo = k > 0; a.L = L; a.Lr = Lr; a.nchunk = nchunk;
    if (a.has_hi) a.hi = edge_flow(flows[k], nl, B, Lr);
    if (a.has_lo) a.lo = edge_flow(flows[k - 1], nl, B, Lr);
    a.dh0 = (const bf16_t*)(W + wl.dh); a.dskip = wk.dskip; a.d_out = d_audio_out_dev; a.d_in = d_audio_in_dev; a.d_out_bs = d_out_bs; a.d_in_bs = d_in_bs;
    k_edge_bwd<<<dim3(gx, B), 256, 0, s>>>(a);
    if (k > 0) {
      const facppg_glow_flow& f = flows[k - 1];
      if (!pack_per_step()) {
        char* P = (char*)f.packed;
        Packer pk;
        add_backward_images(pk, f.w, nl, P + pl.rst, pl.rst_one, P + pl.int_, pl.int_one, (uint4*)(P + pl.condt));
        if (int rc = pk.launch(s)) return rc;
      }
      const WnPacked pw = packed_view((const char*)f.packed, pl);
      if (int rc = wn_layers_backward_data(nl, pw, (const char*)f.state, wk, B, L, s)) return rc;
      // (Round 6 experiment: the conditioning gradient, the weight-gradient products and the bias column sums as three parallel
      // branches -- side streams forked / joined by events, parallel branches of the captured graph -- measured SLOWER: 8.63 -> 9.50 ms
      // at batch 3, 15.0 -> 15.6 at batch 12; a fork + two joins per flow cost more cross-queue latency than the overlap returns.)
      if (int rc = wn_layers_backward_dspect(nl, pw, wk, dspect_pm_dev, accumulate_dspect || k < n, B, L, s)) return rc;
      if (int rc = wn_layers_backward_weights(f.g, nl, (const char*)f.state, wk, spect_pm_dev, B, L, s, s)) return rc;
    }
  }
  {
    EdgeSumArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.nparts = B * gx;
    for (int k = 0; k < n; ++k) {
      const facppg_glow_flow& f = flows[k];
      sa.f[k] = EdgeSumFlow{f.part, f.g->end_w, f.g->end_b, f.g->start_w, f.g->start_b, f.d_conv_w, f.logdet + 1, f.g_logdet, f.ld_scale, f.c};
    }
    k_edge_sum<<<dim3((EPART + 15) / 16, n), 256, 0, s>>>(sa);
  }
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}
