// WaveGlow.infer of ONE short utterance as ONE persistent launch (MI355X / gfx950).
//
// Replaces, for launches smaller than the chip, the launch-per-layer sequence of facppg_wg.hip (src/waveglow/glow.py:252-293
// WaveGlow.infer, :154-175 WN.forward, :88-97 Invertible1x1Conv reverse): the metric's "real-time factor at batch = 1" case.
// One 200-frame utterance at hop 256 is 32 phases x 200 group positions per WaveNet layer: cut into 32-column MFMA tiles
// that is 224 tiles of which a quarter of the last one per phase is real, one tile per CU, 96 + 13 dependent launches of
// ~85 us.  Here instead
//
//   * the work is split over OUTPUT CHANNELS, not columns: workgroup (phase ph, slice j) owns 32 of the 256 channels of
//     phase ph for ALL its columns -- 32 phases x 8 slices = 256 workgroups = one per CU.  Its gate GEMM is [64 x 1088] x
//     [1088 x T] on v_mfma_f32_16x16x4_f32, columns in blocks of 16 (T = 200: 13 blocks, 4 % padding instead of 28 %), each
//     weight fragment feeding 13 MFMAs; the weights a CU streams per layer shrink from the whole 2.2 MB matrix to its own
//     278 KB slice, the activations it streams (1088 x T x 4 B per layer) arrive by LDS-DMA (global_load_lds_dwordx4);
//   * the 8 slices of a phase exchange their gated activations (all-gather through L2, sc1 stores / sc1 DMA loads, one flag
//     per producer) for the res GEMM, and a layer's output h is read as the dilated taps of the next layer by the
//     workgroups of phases ph - d, ph, ph + d: producer -> consumer flags, no grid-wide barrier per layer;
//   * the K order of every inference kernel is [conditioning | tap -d | tap 0 | tap +d]: the conditioning rows depend on
//     the mel frames only, so a workgroup computes them (and those of the NEXT layer) while the exchanges are in flight;
//   * the residual stream h of a workgroup's own channels never leaves its registers between layers; the end rows
//     (facppg_wg.hip, "folded flow edges") accumulate in registers over the 8 layers of a flow, and the affine coupling /
//     inverse 1x1 conv / early-z concat of the flow end run in the same launch (one grid-wide arrival counter per flow
//     guards the re-use of the exchange buffers).
//
// Every sum is formed in the order the per-layer kernels form it (same k16() K order, same fmaf chains at the flow ends):
// the audio equals facppg_wg_infer's launch-per-layer result BIT FOR BIT (tests/test_gpu_waveglow.py).
//
// Layouts ("quad" = 4 channels of one 16-group that one lane feeds to the four MFMAs of a K group, see k16()):
//   channel c of a 256-channel operand -> quad Q = 4*(c/16) + kq, slot s with c % 16 = k16(s, kq);
//   HB[i]  [64 quads][P][Tq][4]   h after layer i (i = 0..n_layers-2), Tq = 16 + PITCH + 16 frames (zero margins = conv padding)
//   AB[i]  [64 quads][P][PITCH][4] gated activations of layer i (PITCH = 16 NB rounded up to 64: every B image has such rows)
//   XA[f%2][4 quads][P][Tq][4]    a flow's first-layer operand: (ch k16(0,kq), ch k16(1,kq), 0, 0) of the <= 5 conditioning
//                                 channels (+ the in-utterance indicator), the same quads serving all three taps
//   MELQ   [kcp/4 quads][PITCH][4] folded conditioning operand, row r = j*80 + m' <- mel[m'][q - j]
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "facppg_wg_internal.h"

namespace facppg {

namespace {

constexpr int NSL = 8;                 // channel slices per phase = workgroups per phase (32 channels each)
constexpr int NWV = 8;                 // waves per workgroup: one hand-off flag each
constexpr int WKCH = 64;               // K rows per chunk (one LDS buffer)
constexpr int KG = WKCH / 16;          // K groups of 16 per chunk
constexpr int QC = WKCH / 4;           // quads per chunk
constexpr int NGT = 3 * C / 16;        // 48 K groups of the three taps
constexpr int NTAPC = 3 * C / WKCH;    // 12 tap chunks
constexpr int NRESC = C / WKCH;        // 4 chunks of the res GEMM / of the end rows
constexpr int CPT = C / WKCH;          // chunks per tap
#ifndef FACPPG_WGP_HOIST
#define FACPPG_WGP_HOIST 2             // conditioning chunks of the NEXT layer computed between a layer's gate and its res GEMM
#endif
constexpr int NH = FACPPG_WGP_HOIST;
#ifndef WGP_EARLY_EXPR
#define WGP_EARLY_EXPR (w < 4)
#endif

// device tables (read with scalar loads)
struct WgpLayer {
  const float4* w1;   // taps image [NSL][48][4 waves][64 lanes], or [NSL][3][4][64] for a flow's first layer (folded through the start conv)
  const float4* wc;   // conditioning image [P][NSL][ngc][4][64]
  const float4* w2;   // res image [NSL][16][2][64]; null for a flow's last layer
  const float* we;    // end-row image of this layer (k_fold_end_rows)
  const float* b1;    // [NSL][4][16] gate bias in this kernel's row order
  const float* b2;    // [NSL][2][16] res bias
  int flow, i, first, last, dil, pad0;
};
struct WgpFlow {
  const float* endb;     // [8] folded end bias
  const float* winv;     // [cc][cc] inverse 1x1 conv
  const float* start_w;  // [256][n_half]
  const float* start_b;  // [256]
  int n_half, early, swap, swap_next, final_flow, early_index;
};

struct WgpArgs {
  const WgpLayer* layers;
  const WgpFlow* flows;
  int n_total, wn_layers;
  int T, P, Tq, ncc, ngc;
  float* hb;             // + i * hb_stride (floats)
  float* ab;             // + i * ab_stride
  float* xa;             // + (f & 1) * xa_stride
  const float* melq;
  float* aud;            // [2][8][La]
  const float* z;
  float* audio;
  unsigned* flag_h;
  unsigned* flag_a;
  unsigned* bar;
  long long hb_stride, ab_stride, xa_stride;
  float sigma;
  int La, L, n_rem_first, swap_begin, xcd_map;
  unsigned long long poll_limit;   // wall-clock ticks a blocking wait may take before it traps (0: unbounded)
  const unsigned* prog;            // chunk descriptors, n_chunks of them + three invalid ones
  int n_chunks;
};

enum { CT_COND = 0, CT_TAP = 1, CT_FIRST = 2, CT_RES = 3, CT_END = 4 };
enum { PRE_NONE = 0, PRE_H = 1, PRE_ACT = 2 };
enum { POST_NONE = 0, POST_GATE = 1, POST_EPI = 2 };
// One 32-bit descriptor per chunk of the program (build_program, host): what the kernel computes, in order.
#define D_TYPE(d) ((int)((d) & 7u))
#define D_IDX(d) ((int)(((d) >> 3) & 31u))
#define D_PRE(d) ((int)(((d) >> 8) & 3u))
#define D_POST(d) ((int)(((d) >> 10) & 3u))
#define D_I(d) ((int)(((d) >> 12) & 15u))      /* the layer's index in its flow: dilation 1 << i */
#define D_LWN(d) ((int)(((d) >> 16) & 1u))     /* a conditioning chunk of the NEXT layer, hoisted into this layer's program */
#define D_VALID(d) ((int)(((d) >> 17) & 1u))
#define D_NEWL(d) ((int)(((d) >> 18) & 1u))    /* first chunk of its layer */
#define D_TAP(d) ((int)(((d) >> 19) & 3u))     /* CT_TAP: which tap (0, 1, 2 = -d, 0, +d) ... */
#define D_CQ(d) ((int)(((d) >> 21) & 3u))      /* ... and which 64-channel chunk of it */
#define D_L(d) ((int)((d) >> 24))              /* layer, processing order */

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void touch4(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void touchu(unsigned& v) { asm volatile("" : "+v"(v)); }

__device__ __forceinline__ unsigned load_flag(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Hand-off payloads are stored WRITE-THROUGH (sc1) and read with sc1 loads / sc1 LDS-DMA, which bypass the reader's L1 (guide,
// Guideline 16 R1): no release / acquire fences on either side.  Buffer descriptors: wave-uniform base, per-lane byte offset.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void store16_sc1(rsrc_t r, int byte_off, float a, float b, float c, float d) {
  const u32x4 v = {__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, 16);
}
__device__ __forceinline__ void store8_sc1(rsrc_t r, int byte_off, float a, float b) {
  const u32x2 v = {__float_as_uint(a), __float_as_uint(b)};
  __builtin_amdgcn_raw_buffer_store_b64(v, r, byte_off, 0, 16);
}
__device__ __forceinline__ float2 load8_sc1(rsrc_t r, int byte_off) {
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 16);
  return make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));
}

// phase row and frame shift of tap t (0, 1, 2 = -d, 0, +d) of a layer with dilation d, seen from phase ph
__device__ __forceinline__ void tap_row(int ph, int d, int t, int P, int& pht, int& qsh) {
  const int pp = ph + (t - 1) * d;
  qsh = pp >= 0 ? pp / P : -((P - 1 - pp) / P);
  pht = pp - qsh * P;
}

// a flow's first-layer operand at one position: rows 0..hn-1 = the conditioning channels, row hn = 1 (inside the utterance),
// zeros above, stored as the four quads (ch k16(0,kq), ch k16(1,kq), 0, 0)
__device__ __forceinline__ void store_xa(rsrc_t xr, const float (&x)[8], int P, int Tq, int ph, int col) {
#pragma unroll
  for (int kq = 0; kq < 4; ++kq)
    store16_sc1(xr, ((kq * P + ph) * Tq + HQ + col) * 16, x[k16(0, kq)], x[k16(1, kq)], 0.0f, 0.0f);
}

// ---- the flow end of one position (k_flow_end's folded path, facppg_wg.hip: same operations in the same order)
struct FlowEndArgs {
  const float* z;
  float* audio;
  float sigma;
  int La, L, n_rem_first, P, Tq;
};
template <int H, bool EARLY>
__device__ __forceinline__ void flow_end_position(const FlowEndArgs& p, const WgpFlow& f, const float (&o)[8], int pos, int ph, int col,
                                                  const float* aud_in, float* aud_out, rsrc_t xa_out) {
  constexpr int CC = 2 * H, CN = EARLY ? CC + 2 : CC, HN = CN / 2;
  float a[CC];
#pragma unroll
  for (int j = 0; j < CC; ++j) a[j] = aud_in[(size_t)j * p.La + pos];
  if (f.swap) {
#pragma unroll
    for (int j = 0; j < H; ++j) a[j] = (a[j] - o[j]) / expf(o[H + j]);
  } else {
#pragma unroll
    for (int j = 0; j < H; ++j) a[H + j] = (a[H + j] - o[j]) / expf(o[H + j]);
  }
  float y[CN];
  if (EARLY) {
    const float* ze = p.z + (size_t)(p.n_rem_first + 2 * f.early_index) * p.L;
#pragma unroll
    for (int j = 0; j < 2; ++j) y[j] = p.sigma * ze[(size_t)j * p.L + pos];
  }
#pragma unroll
  for (int i = 0; i < CC; ++i) {
    float v = 0.0f;
#pragma unroll
    for (int j = 0; j < CC; ++j) v = fmaf(f.winv[i * CC + j], a[j], v);
    y[(EARLY ? 2 : 0) + i] = v;
  }
  if (f.final_flow) {
    float* dst = p.audio + (size_t)pos * CN;   // glow.py:292: sample n = 8*pos + channel
#pragma unroll
    for (int j = 0; j < CN; ++j) dst[j] = y[j];
    return;
  }
#pragma unroll
  for (int j = 0; j < CN; ++j) aud_out[(size_t)j * p.La + pos] = y[j];
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = j < HN ? (f.swap_next ? y[(HN + j) % CN] : y[j % CN]) : j == HN ? 1.0f : 0.0f;
  store_xa(xa_out, x, p.P, p.Tq, ph, col);
}

// B operand by LDS-DMA.  A wave moves 4 consecutive quad rows of a chunk (row0 + qi * rowstride -> lrow0 + qi * PITCH
// columns), each row SEG = PITCH / 64 pieces of 64 columns (1 KiB) that share the row's base address and LDS address and
// differ by the instruction's immediate offset (it applies to both sides).  Piece k = (row k / SEG, piece k % SEG); the
// pieces of the NEXT chunk are issued one at a time between the MFMA groups of the current one (an LDS-DMA instruction
// costs the issuing wave 60-180 cycles: in one burst in front of the MFMAs that is 1-3 us of a 2.8 us step).  Every source
// is read with sc1 (hand-off buffers must bypass L1; for the mel operand it is merely harmless).
template <int NBc, int PITCH, int K>
__device__ __forceinline__ void dma_piece(const char* row0, unsigned rowstride, float* lrow0, unsigned lane16, int lane) {
  constexpr int SEG = PITCH / 64, qi = K / SEG, sg = K % SEG;
  constexpr bool full = sg * 64 + 64 <= NBc;
  if constexpr (sg * 64 < NBc) {
    const char* row = row0 + (size_t)qi * rowstride;
    float* lrow = lrow0 + qi * PITCH * 4;
    if (full || lane < NBc - sg * 64)        // the row's last piece may be partial: only the lanes with live columns
      __builtin_amdgcn_global_load_lds((gptr_t)(row + lane16), (lptr_t)lrow, 16, sg * 1024, 16);
  }
}

// ------------------------------------------------------------------------------------------
// The persistent kernel.  NB = column blocks of 16 (T <= 16 NB frames).  512 threads = 8 waves, TWO per SIMD (<= 256
// registers each): wave w and wave w + 4 share a SIMD and its matrix pipe.
//   gate GEMM  wave w owns the 16-row block w & 3 -- 8 channels, tanh and sigmoid rows interleaved so that a lane holds both
//              pre-activations of its channels -- for one half of the column blocks (w >> 2);
//   res GEMM   wave w owns 16 of the slice's 32 res rows (w & 1) for a quarter of the column blocks (w >> 1);
//   end rows   waves 0 and 1 own the column blocks j and j + 8 of the phase (also at the flow ends).
//
// One STEP of the pipeline = one chunk (64 K rows) of the host-built program (build_program): [wait: everything this wave
// issued in the previous step has landed | barrier | MFMAs of this chunk, and in between: the PREP of the next chunk
// (decode, addresses, its A operand as plain loads into the other register set) and then its B operand, one LDS-DMA piece
// per MFMA group | post event].  Everything that is not an MFMA blocks the issuing wave (a DMA piece 60-180 cycles, the
// prep ~1500), so the two waves of a SIMD take turns: waves 0-3 prepare after their first MFMA group, waves 4-7 half-way
// through, and each issues its pieces under the other's MFMAs.  A step's plain loads are first used one step later, behind
// that step's own vmcnt(0) (they are "touched" there, which is where hipcc places its wait).
// ------------------------------------------------------------------------------------------
// -DFACPPG_WGP_PROF: cycle stamps of workgroup 0 / wave 0 per pipeline phase, printed at the end of the launch (tools/wgp_debug.py)
#ifdef FACPPG_WGP_PROF
#define WGP_STAMP(i) do { const long long n__ = clock64(); prof[i] += n__ - pt; pt = n__; } while (0)
#else
#define WGP_STAMP(i)
#endif

template <int NB>
__global__ __launch_bounds__(512, 2) void k_wg_persist(const WgpArgs kargs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // The kernel arguments are read where they are needed, through the kernarg segment (constant address space: scalar loads
  // the register allocator can re-issue instead of spilling ~60 scalars; and no by-value struct whose address reaches a
  // lambda, which would be copied to scratch)
  typedef const __attribute__((address_space(4))) WgpArgs* kargs_t;
  const kargs_t ka = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
  (void)kargs;
#define a_layers (ka->layers)
#define a_flows (ka->flows)
#define a_n_total (ka->n_total)
#define a_wn_layers (ka->wn_layers)
#define a_T (ka->T)
#define a_P (ka->P)
#define a_Tq (ka->Tq)
#define a_ngc (ka->ngc)
#define a_hb (ka->hb)
#define a_ab (ka->ab)
#define a_xa (ka->xa)
#define a_melq (ka->melq)
#define a_aud (ka->aud)
#define a_z (ka->z)
#define a_audio (ka->audio)
#define a_flag_h (ka->flag_h)
#define a_flag_a (ka->flag_a)
#define a_bar (ka->bar)
#define a_hb_stride (ka->hb_stride)
#define a_ab_stride (ka->ab_stride)
#define a_xa_stride (ka->xa_stride)
#define a_sigma (ka->sigma)
#define a_La (ka->La)
#define a_L (ka->L)
#define a_n_rem_first (ka->n_rem_first)
#define a_swap_begin (ka->swap_begin)
#define a_xcd_map (ka->xcd_map)
#define a_poll_limit (ka->poll_limit)
#define a_n_chunks (ka->n_chunks)
#ifdef FACPPG_WGP_PROF
  long long prof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt = clock64();
  const long long pt0 = pt;
#endif
  constexpr int NBc = 16 * NB;
  constexpr int NBH = (NB + 1) / 2;                                // gate GEMM: column blocks of the first half (the second has NB - NBH)
  constexpr int NBQ = (NB + 3) / 4;                                // res GEMM: column blocks per wave, at most
  constexpr int PITCH = (NBc + 63) / 64 * 64, SEG = PITCH / 64;    // columns per quad row of every B image (global and LDS)
  constexpr int BUF = QC * PITCH * 4;                              // floats per LDS B buffer
  constexpr int NPIECE = 2 * SEG;                                  // DMA pieces per wave per chunk (2 quad rows)
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);          // (provably wave-uniform: scalar branches, scalar addresses)
  const bool early = WGP_EARLY_EXPR;                                        // which of a SIMD's two waves prepares first (see above)
  int ph, j;
  {
    const int lin = blockIdx.x;
    if (a_xcd_map) { const int r = lin >> 3; ph = (lin & 7) + 8 * (r / NSL); j = r % NSL; }   // workgroup i runs on XCD i % 8: a phase's slices share an L2
    else { ph = lin / NSL; j = lin % NSL; }
  }
  const int P = a_P, Tq = a_Tq, T = a_T;
  const int rb = w & 3, gh = w >> 2;                        // gate GEMM: row block, column half
  const int Gw = 2 * j + (rb >> 1), hw = rb & 1;            //   its 16-group and which half of each quad it produces
  const int gcb0 = gh ? NBH : 0, gn = gh ? NB - NBH : NBH;  //   its column blocks
  const int rbk = w & 1, rq = w >> 1, G2 = 2 * j + rbk;     // res GEMM: row block, column quarter
  const int rcb0 = rq * (NB / 4) + min(rq, NB % 4), rn = NB / 4 + (rq < NB % 4 ? 1 : 0);
  const int ecb = j + 8 * w;                                // end rows / flow end: this wave's column block (waves 0, 1)
  const bool evalid = w < 2 && ecb < NB;
  float* const escr = smem + 2 * BUF + w * 128;             // [8 rows][16 columns] exchange of the end rows at a flow end
  float* const wel = smem + 2 * BUF + 1024;                 // the layer's end-row image (16 KiB), staged with its first res chunk
  unsigned* const prog = reinterpret_cast<unsigned*>(smem + 2 * BUF + 1024 + 4096);   // the program (<= 2304 descriptors)
  const unsigned lane16 = (unsigned)lane * 16u;

  f32x4 acc[NBH];       // gate pre-activations: rows 4*kq + {0, 1, 2, 3} = tanh(ca), tanh(cb), sigmoid(ca), sigmoid(cb)
  f32x4 acc2[NBQ];      // res rows 4*kq + r = channels 16*G2 + k16(r, kq)
  f32x4 hres[NBQ];      // the residual stream of those channels (the layer's input h), resident across layers
  f32x4 eacc = {0.f, 0.f, 0.f, 0.f}, etot = {0.f, 0.f, 0.f, 0.f};   // end rows 4*kq + r (kq < 2) of column block ecb
  f32x4 areg[2][KG];    // A operand of the chunk computed in this step / loaded for the next one
#pragma unroll
  for (int c = 0; c < NBQ; ++c) { acc2[c] = eacc; hres[c] = eacc; }
#pragma unroll
  for (int q = 0; q < KG; ++q) areg[0][q] = areg[1][q] = eacc;

  // ---- the program (into LDS: a descriptor is one ds_read away) and the per-layer scalar state -------------------
  for (int i = tid; i < a_n_chunks + 3; i += 512) prog[i] = i < a_n_chunks ? ka->prog[i] : 0u;
  __syncthreads();
  auto prog_at = [&](int i) __attribute__((always_inline)) -> unsigned { return __builtin_amdgcn_readfirstlane(prog[i]); };
  int n = 0;
  unsigned d_cur = prog_at(0), d_nxt = prog_at(1), d_nn = prog_at(2);
  if (!D_VALID(d_cur)) return;
  WgpLayer Lc;                          // table entry of the layer being computed
  const float* b1_n;                    // gate bias of the NEXT layer (its accumulators are seeded at this layer's gate)
  // Operand bases of the current layer with this wave's / workgroup's share already added, so that the prep of a chunk is a
  // select and a multiply-add (scalar registers; recomputed once per layer):
  //   B rows (LDS-DMA sources, 2 quad rows per wave): conditioning image, the three taps, the folded first layer, the res operand
  const char* const bs_cond = (const char*)a_melq + (size_t)(2 * w * PITCH) * 16;
  const char *bs_tap0, *bs_tap1, *bs_tap2, *bs_first, *bs_res;
  const unsigned st_tap = (unsigned)(P * Tq) * 16u, st_res = (unsigned)(P * PITCH) * 16u;     // quad row -> quad row
  const unsigned mu_tap = (unsigned)(QC * P * Tq) * 16u, mu_res = (unsigned)(QC * P * PITCH) * 16u;   // chunk -> chunk
  //   A operand (plain loads, float4 index): conditioning of this / the next layer, taps, res
  const f32x4 *as_cond, *as_cond_n, *as_w1, *as_w2;
  int cur_l = 0;
  auto enter_layer = [&](int l) __attribute__((always_inline)) {
    Lc = a_layers[l];
    cur_l = l;
    const int l2 = min(l + 1, a_n_total - 1);
    b1_n = a_layers[l2].b1;
    const size_t cond_off = (size_t)((ph * NSL + j) * a_ngc * 4 + rb) * 64;
    as_cond = reinterpret_cast<const f32x4*>(Lc.wc) + cond_off;
    as_cond_n = reinterpret_cast<const f32x4*>(a_layers[l2].wc) + cond_off;
    as_w1 = reinterpret_cast<const f32x4*>(Lc.w1) + (size_t)(j * (Lc.first ? 3 : NGT) * 4 + rb) * 64;
    as_w2 = reinterpret_cast<const f32x4*>(Lc.w2) + (size_t)(j * 16 * 2 + rbk) * 64;
    const char* hb_in = (const char*)(a_hb + (size_t)max(Lc.i - 1, 0) * a_hb_stride);
    const char* xa_cur = (const char*)(a_xa + (size_t)(Lc.flow & 1) * a_xa_stride);
    int pht, qsh, t0, t1, t2;
    tap_row(ph, Lc.dil, 0, P, pht, qsh); t0 = pht * Tq + qsh;
    tap_row(ph, Lc.dil, 1, P, pht, qsh); t1 = pht * Tq + qsh;
    tap_row(ph, Lc.dil, 2, P, pht, qsh); t2 = pht * Tq + qsh;
    const size_t wrow = (size_t)(2 * w * P) * Tq;
    bs_tap0 = hb_in + (wrow + t0 + HQ) * 16; bs_tap1 = hb_in + (wrow + t1 + HQ) * 16; bs_tap2 = hb_in + (wrow + t2 + HQ) * 16;
    // folded first layer: quads 4 tap + kq; wave w < 6 moves quads kq = 2 (w & 1), + 1 of tap w >> 1 (dilation 1: t0..t2 above)
    const int tw = w >> 1, tf = tw == 0 ? t0 : tw == 1 ? t1 : t2;
    bs_first = xa_cur + ((size_t)(2 * (w & 1) * P) * Tq + tf + HQ) * 16;
    bs_res = (const char*)(a_ab + (size_t)Lc.i * a_ab_stride) + (size_t)(2 * w * P + ph) * PITCH * 16;
  };
  enter_layer(0);

  // ---- hand-offs -----------------------------------------------------------------------------------------------
  // One flag per producing WAVE (8 per workgroup): a wave publishes what it stored as soon as ITS stores have drained, with
  // no workgroup barrier in the protocol (guide, Guideline 16 R1 per wave: write-through stores, s_waitcnt vmcnt(0), one
  // lane's relaxed agent-scope flag store).  Consumers: PRE_ACT = the 64 waves of this phase's 8 slices (one flag per
  // lane); PRE_H = the 64 waves of each of the phases ph - d, ph, ph + d.  Epoch l + 1 = "what layer l reads is there".
  // Publishing is DEFERRED to the top of the next step, where the wave waits for vmcnt(0) anyway: draining the stores right
  // behind the post event would idle the wave for the 1-2 us a write-through store takes.
  auto flags_ready = [&](unsigned d) __attribute__((always_inline)) -> bool {
    const unsigned need = (unsigned)(D_L(d) + 1);
    const int li = (lane >> 3) * NWV + (lane & 7);      // lane -> (slice, wave) of a phase
    bool ok;
    if (D_PRE(d) == PRE_ACT) {
      ok = load_flag(a_flag_a + ph * NSL * NWV + li) >= need;
    } else {
      int p0, p1, p2, qsh;
      tap_row(ph, 1 << D_I(d), 0, P, p0, qsh);
      tap_row(ph, 1 << D_I(d), 1, P, p1, qsh);
      tap_row(ph, 1 << D_I(d), 2, P, p2, qsh);
      const unsigned v0 = load_flag(a_flag_h + p0 * NSL * NWV + li), v1 = load_flag(a_flag_h + p1 * NSL * NWV + li),
                     v2 = load_flag(a_flag_h + p2 * NSL * NWV + li);
      ok = v0 >= need && v1 >= need && v2 >= need;
    }
    return __all(ok);
  };
  unsigned* pub_flag = nullptr;
  unsigned pub_epoch = 0;
  auto publish = [&](unsigned* flags, unsigned epoch) __attribute__((always_inline)) { pub_flag = flags + w; pub_epoch = epoch; };
  auto publish_now = [&]() __attribute__((always_inline)) {      // (behind an s_waitcnt vmcnt(0) of this wave)
    if (pub_flag) {
      if (lane == 0) __hip_atomic_store(pub_flag, pub_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pub_flag = nullptr;
    }
  };
  auto blocking_wait = [&](unsigned d) __attribute__((always_inline)) {
    unsigned long long t0 = 0;
    unsigned spins = 0;
    while (!flags_ready(d)) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 255u) == 0 && a_poll_limit) {
        const unsigned long long now = wall_clock64();
        if (!t0) t0 = now;
        else if (now - t0 > a_poll_limit) __builtin_trap();
      }
    }
  };
  auto wait_counter = [&](const unsigned* cp, unsigned target) __attribute__((always_inline)) {
    unsigned long long t0 = 0;
    unsigned spins = 0;
    while (load_flag(cp) < target) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 255u) == 0 && a_poll_limit) {
        const unsigned long long now = wall_clock64();
        if (!t0) t0 = now;
        else if (now - t0 > a_poll_limit) __builtin_trap();
      }
    }
  };

  // ---- operand movement ----------------------------------------------------------------------------------------
  // B operand of the NEXT chunk: dma_prep() names the 2 quad rows this wave moves (source row 0, row stride, LDS row 0);
  // the chunk being computed issues the pieces between its MFMA groups (dma_one), dma_rest() whatever is left.
  const char* dma_row0 = nullptr;
  unsigned dma_stride = 0;
  float* dma_l0 = smem;
  bool dma_on = false;
  auto dma_prep = [&](unsigned d, int buf) __attribute__((always_inline)) {
    dma_l0 = smem + buf * BUF + (2 * w) * PITCH * 4;
    const int type = D_TYPE(d), idx = D_IDX(d);
    dma_on = true;
    if (type == CT_COND) {
      dma_row0 = bs_cond + (unsigned)idx * (unsigned)(QC * PITCH * 16); dma_stride = PITCH * 16u;
    } else if (type == CT_TAP) {
      // (selects over VALUES the lambda copies below, not a ?: over the captured variables themselves: hipcc turns the latter
      //  into an indexed load from the closure, which then has to live in scratch memory -- with every scalar it captures)
      const int t = D_TAP(d);
      const unsigned long long b0 = (unsigned long long)bs_tap0, b1 = (unsigned long long)bs_tap1, b2 = (unsigned long long)bs_tap2;
      dma_row0 = (const char*)(b0 + (unsigned long long)(t >= 1) * (b1 - b0) + (unsigned long long)(t >= 2) * (b2 - b1) + (unsigned)D_CQ(d) * mu_tap);
      dma_stride = st_tap;
    } else if (type == CT_FIRST) {
      dma_row0 = bs_first; dma_stride = st_tap;
      dma_on = w < 6;
    } else {
      dma_row0 = bs_res + (unsigned)idx * mu_res; dma_stride = st_res;
      if (idx == 0) {      // the layer's end-row image: 16 KiB = 16 pieces, 2 per wave (once per layer: issued here)
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
          __builtin_amdgcn_global_load_lds((gptr_t)((const char*)Lc.we + (w * 2 + ii) * 1024 + lane16), (lptr_t)(wel + (w * 2 + ii) * 256), 16, 0, 0);
      }
    }
  };
  auto dma_one = [&](auto k_c) __attribute__((always_inline)) {
    constexpr int K = decltype(k_c)::value;
    if constexpr (K >= 0 && K < NPIECE) {
      if (dma_on) dma_piece<NBc, PITCH, K>(dma_row0, dma_stride, dma_l0, lane16, lane);
    }
  };
  auto dma_from = [&](auto from_c) __attribute__((always_inline)) {      // pieces FROM .. NPIECE - 1
    constexpr int FROM = decltype(from_c)::value;
    if (dma_on) {
#define WGP_PIECE(K) if constexpr (K >= FROM && K < NPIECE) dma_piece<NBc, PITCH, K>(dma_row0, dma_stride, dma_l0, lane16, lane);
      WGP_PIECE(0) WGP_PIECE(1) WGP_PIECE(2) WGP_PIECE(3) WGP_PIECE(4) WGP_PIECE(5) WGP_PIECE(6) WGP_PIECE(7)
#undef WGP_PIECE
    }
  };
  // A operand of a chunk: plain loads into the register set of the NEXT step
  auto load_a = [&](unsigned d, f32x4 (&a)[KG]) __attribute__((always_inline)) {
    const int type = D_TYPE(d), idx = D_IDX(d);
    const f32x4* src = nullptr;
    int stride = 256;
    if (type == CT_COND) {
      const bool next_layer = D_LWN(d) || D_L(d) != cur_l;   // a hoisted chunk of the next layer, or the next layer's own first chunk
      const unsigned long long c0 = (unsigned long long)as_cond, c1 = (unsigned long long)as_cond_n;
      src = reinterpret_cast<const f32x4*>(c0 + (unsigned long long)next_layer * (c1 - c0)) + idx * (KG * 256);
    } else if (type == CT_TAP) src = as_w1 + idx * (KG * 256);
    else if (type == CT_FIRST) src = as_w1;
    else if (type == CT_RES) { src = as_w2 + idx * (KG * 128); stride = 128; }
    if (src) {
      src += lane;
#pragma unroll
      for (int kg = 0; kg < KG; ++kg)
        if (kg < 3 || type != CT_FIRST) a[kg] = src[kg * stride];
    }
  };
  // The PREP of the next chunk, somewhere inside the current chunk's MFMAs: is it ready (nothing to wait for, or the flag
  // snapshot taken a step ago says so)?  Then name its B rows and load its A operand; take the snapshot for the chunk after.
  bool prepped = false;
  auto prep_next = [&](auto par_c) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;
    WGP_STAMP(3);
    prepped = D_VALID(d_nxt);
    if (prepped && D_PRE(d_nxt) != PRE_NONE) prepped = flags_ready(d_nxt);   // one poll (this wave alone waits for it: its SIMD partner keeps the matrix pipe busy)
    dma_on = false;
    if (prepped) {
      dma_prep(d_nxt, PAR ^ 1);
      load_a(d_nxt, areg[PAR ^ 1]);
    }
    WGP_STAMP(2);
  };

  // ---- compute -------------------------------------------------------------------------------------------------
  // Gate GEMM of one chunk: K group -> GROUPS of 3-4 column blocks -> the four K steps of the group -> its column blocks.
  // Every accumulator meets its K entries in order (s = 0..3 within a K group, K groups in order); 3-4 independent
  // accumulators per MFMA chain cover the 40-cycle dependent latency; two groups' B values (<= 32 registers) are live.
  // Inside a group: the MFMAs of K step 0, THEN the next group's ds_reads, the prep or a DMA piece of the next chunk, then
  // K steps 1-3 -- the wait for this group's operands has nothing younger in front of it (hipcc waits lgkmcnt(0)).
  constexpr int NGRP = (NBH + 3) / 4, GBASE = NBH / NGRP, GREM = NBH % NGRP;   // groups per K group; the first GREM have GBASE + 1 blocks
  auto gate_chunk = [&](const float* bbuf, const f32x4 (&a)[KG], auto nkg_c, auto par_c) __attribute__((always_inline)) {
    constexpr int NKG = decltype(nkg_c)::value;
    constexpr int NIT = NKG * NGRP;
#ifdef WGP_LATE_IT
    constexpr int LATE = WGP_LATE_IT < NIT - 1 ? WGP_LATE_IT : NIT - 2;
#else
    constexpr int LATE = NIT / 2 - 1 > 0 ? NIT / 2 - 1 : 0;     // the iteration after which waves 4-7 prepare (waves 0-3: after iteration 0)
#endif
    const float* bb = bbuf + (kq * PITCH + gcb0 * 16 + l15) * 4;
    f32x4 b[2][4];
    auto grp_first = [](int g) { return g * GBASE + (g < GREM ? g : GREM); };
    auto grp_size = [](int g) { return GBASE + (g < GREM ? 1 : 0); };
    auto load_group = [&](f32x4 (&dst)[4], int it) __attribute__((always_inline)) {
      const int kg = it / NGRP, g = it % NGRP;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (u < grp_size(g)) dst[u] = *reinterpret_cast<const f32x4*>(bb + (kg * 4 * PITCH + min(grp_first(g) + u, gn - 1) * 16) * 4);
    };
    load_group(b[0], 0);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int kg = it / NGRP, g = it % NGRP;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (u < grp_size(g) && grp_first(g) + u < gn) acc[grp_first(g) + u] = mfma16x16x4(a[kg][0], b[it & 1][u][0], acc[grp_first(g) + u]);
      __builtin_amdgcn_sched_barrier(0);
      if (it + 1 < NIT) load_group(b[(it + 1) & 1], it + 1);
      if (it == 0 && early) prep_next(par_c);
      if (it == LATE && !early) prep_next(par_c);
#ifdef WGP_BURST
      if ((it == 0 && early) || (it == LATE && !early)) dma_from(std::integral_constant<int, 0>{});
#else
      // DMA pieces: waves 0-3 one per iteration from iteration 1 on; waves 4-7 theirs in the iterations after LATE
      switch (it) {      // (it is a compile-time constant after unrolling)
#define WGP_IT(I)                                                                                   \
        case I:                                                                                     \
          if (early) {                                                                              \
            if constexpr (I >= 1) dma_one(std::integral_constant<int, I - 1>{});                    \
            if constexpr (I == NIT - 1) dma_from(std::integral_constant<int, NIT - 1>{});           \
          } else if constexpr (I > LATE) {                                                          \
            constexpr int per = (NPIECE + (NIT - 1 - LATE) - 1) / (NIT - 1 - LATE);                 \
            dma_one(std::integral_constant<int, (I - LATE - 1) * per>{});                           \
            if constexpr (per > 1) dma_one(std::integral_constant<int, (I - LATE - 1) * per + 1>{}); \
            if constexpr (per > 2) dma_one(std::integral_constant<int, (I - LATE - 1) * per + 2>{}); \
            if constexpr (I == NIT - 1) dma_from(std::integral_constant<int, (NIT - 1 - LATE) * (per > 3 ? 3 : per)>{}); \
          }                                                                                         \
          break;
        WGP_IT(0) WGP_IT(1) WGP_IT(2) WGP_IT(3) WGP_IT(4) WGP_IT(5) WGP_IT(6) WGP_IT(7)
#undef WGP_IT
        default: break;
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 1; s < 4; ++s) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (u < grp_size(g) && grp_first(g) + u < gn) acc[grp_first(g) + u] = mfma16x16x4(a[kg][s], b[it & 1][u][s], acc[grp_first(g) + u]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // Res GEMM of one chunk: <= 4 column blocks per wave, one K group per iteration; prep after K group 0 (waves 0-3) or 1
  // (waves 4-7), the DMA pieces in the K steps that follow.
  auto res_chunk = [&](const float* bbuf, const f32x4 (&a)[KG], auto par_c) __attribute__((always_inline)) {
    const float* bb = bbuf + (kq * PITCH + rcb0 * 16 + l15) * 4;
    f32x4 b[2][NBQ];
#pragma unroll
    for (int c = 0; c < NBQ; ++c) b[0][c] = *reinterpret_cast<const f32x4*>(bb + (min(c, rn - 1) * 16) * 4);
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < NBQ; ++c)
        if (c < rn) acc2[c] = mfma16x16x4(a[kg][0], b[kg & 1][c][0], acc2[c]);
      __builtin_amdgcn_sched_barrier(0);
      if (kg + 1 < KG) {
#pragma unroll
        for (int c = 0; c < NBQ; ++c)
          b[(kg + 1) & 1][c] = *reinterpret_cast<const f32x4*>(bb + ((kg + 1) * 4 * PITCH + min(c, rn - 1) * 16) * 4);
      }
      if (kg == 0 && early) prep_next(par_c);
      if (kg == 1 && !early) prep_next(par_c);
#pragma unroll
      for (int s = 1; s < 4; ++s) {
        __builtin_amdgcn_sched_barrier(0);
        if (early) {       // 9 slots (K groups 1-3) for <= 8 pieces
          if (kg >= 1) {
            switch ((kg - 1) * 3 + s - 1) {
#define WGP_SLOT(S) case S: dma_one(std::integral_constant<int, S>{}); break;
              WGP_SLOT(0) WGP_SLOT(1) WGP_SLOT(2) WGP_SLOT(3) WGP_SLOT(4) WGP_SLOT(5) WGP_SLOT(6) WGP_SLOT(7)
#undef WGP_SLOT
              default: break;
            }
          }
        } else if (kg >= 2) {   // 6 slots (K groups 2, 3): two pieces in the first two
          switch ((kg - 2) * 3 + s - 1) {
#define WGP_SLOT(S) case S: dma_one(std::integral_constant<int, S>{}); if (S < 2) dma_one(std::integral_constant<int, 6 + S>{}); break;
            WGP_SLOT(0) WGP_SLOT(1) WGP_SLOT(2) WGP_SLOT(3) WGP_SLOT(4) WGP_SLOT(5)
#undef WGP_SLOT
            default: break;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NBQ; ++c)
          if (c < rn) acc2[c] = mfma16x16x4(a[kg][s], b[kg & 1][c][s], acc2[c]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // end rows of column block ecb from this chunk's 64 gated channels = slices 2 idx, 2 idx + 1: per slice a chain of 8 MFMAs
  // from zero over k = 32 s + 4 g + kq, the slices summed in order (k_wn_layer's end rows: the same sums in the same order)
  auto end_chunk = [&](const float* bbuf, int idx) __attribute__((always_inline)) {
    if (!evalid) return;
#pragma unroll
    for (int sl2 = 0; sl2 < 2; ++sl2) {
      const f32x4* wl = reinterpret_cast<const f32x4*>(wel + ((2 * idx + sl2) * 64 + lane) * 8);
      const f32x4 w0 = wl[0], w1 = wl[1];
      f32x4 e = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        // channel 32 sl2 + 4 g + kq of the chunk -> (16-group, quad, slot) of the image: the inverse of k16()
        const int gl = 2 * sl2 + (g >> 2), sp = 2 * ((g & 3) >> 1) + (kq >> 1), kqp = ((kq & 1) << 1) | (g & 1);
        e = mfma16x16x4(g < 4 ? w0[g & 3] : w1[g & 3], bbuf[((gl * 4 + kqp) * PITCH + 16 * ecb + l15) * 4 + sp], e);
      }
      if (idx == 0 && sl2 == 0) etot = e;
      else { etot[0] += e[0]; etot[1] += e[1]; etot[2] += e[2]; etot[3] += e[3]; }
    }
  };

  auto load_bias16 = [&](const float* b) __attribute__((always_inline)) -> f32x4 {   // rows 4 kq .. 4 kq + 3 of a 16-row block
    return *reinterpret_cast<const f32x4*>(b + 4 * kq);
  };
  auto init_gate_acc = [&](const float* b1) __attribute__((always_inline)) {
    const f32x4 b = load_bias16(b1 + (j * 4 + rb) * 16);
#pragma unroll
    for (int cb = 0; cb < NBH; ++cb) acc[cb] = b;
  };

  // ---- post events ---------------------------------------------------------------------------------------------
  auto post_gate = [&](unsigned d) __attribute__((always_inline)) {
    const WgpFlow& F = a_flows[Lc.flow];
    const int l = D_L(d);
    if (Lc.first) {
      // the exchange buffers are re-used by every flow: nobody may still be reading the previous flow's
      if (Lc.flow > 0) wait_counter(a_bar, (unsigned)Lc.flow * gridDim.x);
      // h_0 = start(x_a) of this wave's res channels (glow.py:156), the residual of the first layer
      const rsrc_t xr = make_rsrc(a_xa + (size_t)(Lc.flow & 1) * a_xa_stride);
#pragma unroll
      for (int c = 0; c < NBQ; ++c) {
        const int col = (rcb0 + min(c, rn - 1)) * 16 + l15;
        const float2 q0 = load8_sc1(xr, ((0 * P + ph) * Tq + HQ + col) * 16);   // channels 0, 2
        const float2 q2 = load8_sc1(xr, ((2 * P + ph) * Tq + HQ + col) * 16);   // channels 1, 3
        const float a0[4] = {q0.x, q2.x, q0.y, q2.y};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ch = 16 * G2 + k16(r, kq);
          float v = F.start_b[ch];
#pragma unroll
          for (int jn = 0; jn < 4; ++jn)
            if (jn < F.n_half) v = fmaf(F.start_w[ch * F.n_half + jn], a0[jn], v);
          hres[c][r] = v;
        }
      }
    }
    // gated activations -> AB[i] (write-through), quad Gw*4 + kq, slots 2 hw, 2 hw + 1
    const rsrc_t ar = make_rsrc(a_ab + (size_t)Lc.i * a_ab_stride);
    const int ab_off = (((Gw * 4 + kq) * P + ph) * PITCH + gcb0 * 16 + l15) * 16 + 8 * hw;
#pragma unroll
    for (int cb = 0; cb < NBH; ++cb)
      if (cb < gn) store8_sc1(ar, ab_off + cb * 256, gate_tanh_sigmoid(acc[cb][0], acc[cb][2]), gate_tanh_sigmoid(acc[cb][1], acc[cb][3]));
    if (l + 1 < a_n_total) init_gate_acc(b1_n);
    if (!Lc.last) {
      const f32x4 b = load_bias16(Lc.b2 + (j * 2 + rbk) * 16);
#pragma unroll
      for (int c = 0; c < NBQ; ++c) acc2[c] = b;
    }
    publish(a_flag_a + (ph * NSL + j) * NWV, (unsigned)(l + 1));
  };

  auto post_epi = [&](unsigned d) __attribute__((always_inline)) {
    const WgpFlow& F = a_flows[Lc.flow];
    const int l = D_L(d);
    if (evalid) {
      if (Lc.first) {
        const f32x4 eb = *reinterpret_cast<const f32x4*>(F.endb + 4 * (kq & 1));
#pragma unroll
        for (int r = 0; r < 4; ++r) eacc[r] = eb[r] + etot[r];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) eacc[r] += etot[r];
      }
    }
    if (!Lc.last) {
      const rsrc_t hr = make_rsrc(a_hb + (size_t)Lc.i * a_hb_stride);
      const int hb_off = (((G2 * 4 + kq) * P + ph) * Tq + HQ + rcb0 * 16 + l15) * 16;
#pragma unroll
      for (int c = 0; c < NBQ; ++c) {
        if (c < rn) {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc2[c][r] + hres[c][r];
          hres[c] = v;
          if ((rcb0 + c) * 16 + l15 < T) store16_sc1(hr, hb_off + c * 256, v[0], v[1], v[2], v[3]);
        }
      }
      publish(a_flag_h + (ph * NSL + j) * NWV, (unsigned)(l + 2));
      return;
    }
    // ---- flow end: affine coupling inverse, inverse 1x1 conv, early-z concat, next flow's operand (glow.py:273-290)
    __syncthreads();                                   // every wave is through with this flow's exchange buffers ...
    if (tid == 0) atomicAdd(a_bar, 1u);                // ... arrive; the next flow's first write waits for all arrivals
    if (evalid) {
      if (kq < 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) escr[(4 * kq + r) * 16 + l15] = eacc[r];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (one wave: its LDS operations execute in order)
      __builtin_amdgcn_wave_barrier();
      const int col = ecb * 16 + l15;
      if (kq == 0 && col < T) {
        float o[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) o[r] = escr[r * 16 + l15];
        const int pos = col * P + ph, ai = Lc.flow & 1;
        const float* aud_in = a_aud + (size_t)ai * 8 * a_La;
        float* aud_out = a_aud + (size_t)(ai ^ 1) * 8 * a_La;
        const rsrc_t xo = make_rsrc(a_xa + (size_t)((Lc.flow + 1) & 1) * a_xa_stride);
        const FlowEndArgs fe = {a_z, a_audio, a_sigma, a_La, a_L, a_n_rem_first, P, Tq};
        switch (F.n_half * 2 + F.early) {
          case 2: flow_end_position<1, false>(fe, F, o, pos, ph, col, aud_in, aud_out, xo); break;
          case 3: flow_end_position<1, true>(fe, F, o, pos, ph, col, aud_in, aud_out, xo); break;
          case 4: flow_end_position<2, false>(fe, F, o, pos, ph, col, aud_in, aud_out, xo); break;
          case 5: flow_end_position<2, true>(fe, F, o, pos, ph, col, aud_in, aud_out, xo); break;
          case 6: flow_end_position<3, false>(fe, F, o, pos, ph, col, aud_in, aud_out, xo); break;
          case 7: flow_end_position<3, true>(fe, F, o, pos, ph, col, aud_in, aud_out, xo); break;
          case 8: flow_end_position<4, false>(fe, F, o, pos, ph, col, aud_in, aud_out, xo); break;
          default: flow_end_position<4, true>(fe, F, o, pos, ph, col, aud_in, aud_out, xo); break;
        }
      }
    }
    publish(a_flag_h + (ph * NSL + j) * NWV, (unsigned)(l + 2));
  };

  // ---- prologue: audio = sigma * z (glow.py:261-270) and the first flow's operand (k_begin) --------------------
  {
    const int hn = a_flows[0].n_half;
    if (evalid && kq == 0) {
      const int col = ecb * 16 + l15;
      if (col < T) {
        const int pos = col * P + ph;
        float a[8];
#pragma unroll
        for (int jn = 0; jn < 8; ++jn) {
          a[jn] = jn < 2 * hn ? a_sigma * a_z[(size_t)jn * a_L + pos] : 0.0f;
          if (jn < 2 * hn) a_aud[(size_t)jn * a_La + pos] = a[jn];
        }
        float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int hc = 1; hc <= 4; ++hc) {      // (static register indices only: one arm per possible n_half)
          if (hn == hc) {
#pragma unroll
            for (int jn = 0; jn < hc; ++jn) x[jn] = a_swap_begin ? a[hc + jn] : a[jn];
            x[hc] = 1.0f;
          }
        }
        store_xa(make_rsrc(a_xa), x, P, Tq, ph, col);
      }
    }
    publish(a_flag_h + (ph * NSL + j) * NWV, 1u);
  }

  // ---- the chunk pipeline --------------------------------------------------------------------------------------
  init_gate_acc(Lc.b1);
  dma_prep(d_cur, 0);
  dma_from(std::integral_constant<int, 0>{});
  load_a(d_cur, areg[0]);

  auto step = [&](auto par_c) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;
    WGP_STAMP(8);
    // everything this wave issued during the previous step has landed; after the barrier: everybody's, and every wave is
    // through with the other buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    publish_now();
    WGP_STAMP(0);
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) asm volatile("" : "+v"(areg[PAR][kg]));
    __builtin_amdgcn_s_barrier();
    WGP_STAMP(1);
    const unsigned d_n3 = prog_at(n + 3);
    prepped = false;
    const float* bbuf = smem + PAR * BUF;
    switch (D_TYPE(d_cur)) {
      case CT_COND:
      case CT_TAP: gate_chunk(bbuf, areg[PAR], std::integral_constant<int, KG>{}, par_c); WGP_STAMP(3); break;
      case CT_FIRST: gate_chunk(bbuf, areg[PAR], std::integral_constant<int, 3>{}, par_c); WGP_STAMP(3); break;
      case CT_RES: res_chunk(bbuf, areg[PAR], par_c); __builtin_amdgcn_sched_barrier(0); end_chunk(bbuf, D_IDX(d_cur)); WGP_STAMP(4); break;
      default: prep_next(par_c); dma_from(std::integral_constant<int, 0>{}); end_chunk(bbuf, D_IDX(d_cur)); WGP_STAMP(4); break;
    }
    if (D_POST(d_cur) == POST_GATE) { post_gate(d_cur); WGP_STAMP(5); }
    else if (D_POST(d_cur) == POST_EPI) { post_epi(d_cur); WGP_STAMP(6); }
    if (D_VALID(d_nxt) && D_NEWL(d_nxt)) enter_layer(D_L(d_nxt));   // (a layer's first chunk is never issued late: it waits for nothing)
    if (D_VALID(d_nxt) && !prepped) {
      if (D_PRE(d_nxt) != PRE_NONE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // what this wave owes its consumers first: it may be waiting for itself
        publish_now();
        blocking_wait(d_nxt);
      }
      dma_prep(d_nxt, PAR ^ 1);
      dma_from(std::integral_constant<int, 0>{});
      load_a(d_nxt, areg[PAR ^ 1]);
      WGP_STAMP(7);
    }
    d_cur = d_nxt; d_nxt = d_nn; d_nn = d_n3; ++n;
  };
  while (true) {
    step(std::integral_constant<int, 0>{});
    if (!D_VALID(d_cur)) break;
    step(std::integral_constant<int, 1>{});
    if (!D_VALID(d_cur)) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  publish_now();
#ifdef FACPPG_WGP_PROF
  if (blockIdx.x == 0 && tid == 0)
    printf("wgp prof (cycles, workgroup 0 wave 0): total %lld | vmcnt-wait %lld barrier %lld prep %lld mfma-gate(+dma) %lld mfma-res %lld "
           "post-gate %lld post-epi %lld late-issue(+blocking wait) %lld loop-tail %lld\n",
           clock64() - pt0, prof[0], prof[1], prof[2], prof[3], prof[4], prof[5], prof[6], prof[7], prof[8]);
#endif
}

#undef a_layers
#undef a_flows
#undef a_n_total
#undef a_wn_layers
#undef a_T
#undef a_P
#undef a_Tq
#undef a_ngc
#undef a_hb
#undef a_ab
#undef a_xa
#undef a_melq
#undef a_aud
#undef a_z
#undef a_audio
#undef a_flag_h
#undef a_flag_a
#undef a_bar
#undef a_hb_stride
#undef a_ab_stride
#undef a_xa_stride
#undef a_sigma
#undef a_La
#undef a_L
#undef a_n_rem_first
#undef a_swap_begin
#undef a_xcd_map
#undef a_poll_limit
#undef a_n_chunks

// ------------------------------------------------------------------------------------------
// weight images (once per handle)
// ------------------------------------------------------------------------------------------
// gate row of (slice j, wave w, row m of the wave's 16-row block): see k_wg_persist
__device__ __forceinline__ int gate_row(int j, int w, int m) {
  const int G = 2 * j + (w >> 1), hw = w & 1, g = m >> 2, r = m & 3;
  return (r >> 1) * C + 16 * G + k16(2 * hw + (r & 1), g);
}
__global__ void k_wgp_pack_w1(const float* __restrict__ in_w /*[512][256][3]*/, float4* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // ((j*NGT + g16)*4 + w)*64 + lane
  if (idx >= NSL * NGT * 256) return;
  const int lane = idx & 63, w = (idx >> 6) & 3, g16 = (idx >> 8) % NGT, j = (idx >> 8) / NGT;
  const int row = gate_row(j, w, lane & 15);
  float v[4];
  for (int s = 0; s < 4; ++s) {
    const int kk = 16 * g16 + k16(s, lane >> 4);
    v[s] = in_w[((size_t)row * C + (kk % C)) * 3 + kk / C];
  }
  out[idx] = make_float4(v[0], v[1], v[2], v[3]);
}
__global__ void k_wgp_pack_wc(const float* __restrict__ F /*[512][P*kcp]*/, float4* __restrict__ out, int P, int kcp) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // (((ph*NSL + j)*ngc + g16)*4 + w)*64 + lane
  const int ngc = kcp / 16;
  if (idx >= (size_t)P * NSL * ngc * 256) return;
  const int lane = idx & 63, w = (idx >> 6) & 3, g16 = (idx >> 8) % ngc, j = ((idx >> 8) / ngc) % NSL, ph = (idx >> 8) / ngc / NSL;
  const float* src = F + (size_t)gate_row(j, w, lane & 15) * P * kcp + (size_t)ph * kcp + 16 * g16;
  const int q = lane >> 4;
  out[idx] = make_float4(src[k16(0, q)], src[k16(1, q)], src[k16(2, q)], src[k16(3, q)]);
}
// a flow's first layer: F0 [512][64], column 8*tap + ch -> K' = 16*tap + k16(s, kq) with ch = k16(s, kq) < 8 (else zero)
__global__ void k_wgp_pack_w1f(const float* __restrict__ F0, float4* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // ((j*3 + tap)*4 + w)*64 + lane
  if (idx >= NSL * 3 * 256) return;
  const int lane = idx & 63, w = (idx >> 6) & 3, tap = (idx >> 8) % 3, j = (idx >> 8) / 3;
  const float* src = F0 + (size_t)gate_row(j, w, lane & 15) * 64 + 8 * tap;
  float v[4];
  for (int s = 0; s < 4; ++s) {
    const int ch = k16(s, lane >> 4);
    v[s] = ch < 8 ? src[ch] : 0.0f;
  }
  out[idx] = make_float4(v[0], v[1], v[2], v[3]);
}
__global__ void k_wgp_pack_w2(const float* __restrict__ rs_w /*[>=256][256], res rows first*/, float4* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // ((j*16 + g16)*2 + rbk)*64 + lane
  if (idx >= NSL * 16 * 128) return;
  const int lane = idx & 63, rbk = (idx >> 6) & 1, g16 = (idx >> 7) & 15, j = idx >> 11;
  const int m = lane & 15, row = 16 * (2 * j + rbk) + k16(m & 3, m >> 2);
  float v[4];
  for (int s = 0; s < 4; ++s) v[s] = rs_w[(size_t)row * C + 16 * g16 + k16(s, lane >> 4)];
  out[idx] = make_float4(v[0], v[1], v[2], v[3]);
}
__global__ void k_wgp_pack_bias(const float* __restrict__ b1pm /*[512]*/, const float* __restrict__ b2 /*[>=256] or null*/,
                                float* __restrict__ b1m /*[NSL][4][16]*/, float* __restrict__ b2m /*[NSL][2][16]*/) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < NSL * 64) b1m[idx] = b1pm[gate_row(idx >> 6, (idx >> 4) & 3, idx & 15)];
  if (idx < NSL * 32 && b2) {
    const int m = idx & 15, rbk = (idx >> 4) & 1, j = idx >> 5;
    b2m[idx] = b2[16 * (2 * j + rbk) + k16(m & 3, m >> 2)];
  }
}
// MELQ[Qc][q][s] = mel[m'][q - jj] for row r = 16*(Qc/4) + k16(s, Qc%4) = jj*80 + m' (zero outside the utterance / past kc);
// rows of `pitch` columns
__global__ void k_wgp_melq(const float* __restrict__ mel /*[80][T]*/, float* __restrict__ melq, int T, int pitch, int kc, int kcp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // (Qc*pitch + q)*4 + s
  if (idx >= kcp * pitch) return;
  const int s = idx & 3, q = (idx >> 2) % pitch, Qc = (idx >> 2) / pitch;
  const int r = 16 * (Qc >> 2) + k16(s, Qc & 3), jj = r / NMEL, m = r - jj * NMEL, qq = q - jj;
  melq[idx] = (r < kc && qq >= 0 && qq < T) ? mel[(size_t)m * T + qq] : 0.0f;
}

// The program: one descriptor per chunk, in the order the kernel computes them (see the file header for why this order).
// Per layer l = (flow, i):
//   S1  own conditioning chunks that the previous layer's program has not already computed
//   S2  the three taps (12 chunks) or, for a flow's first layer, the folded taps (1 chunk); waits for the producers of
//       its operand (PRE_H) before the first; GATE after the last
//   S3  the first NH conditioning chunks of layer l + 1 (their operands depend on nothing)
//   S4  res GEMM + end rows (4 chunks; end rows alone in a flow's last layer); waits for the phase's gated activations
//       (PRE_ACT) before the first; EPI after the last
void build_program(int n_total, int wn_layers, int ncc, std::vector<unsigned>& prog, std::vector<int>& layer_end) {
  prog.clear(); layer_end.clear();
  auto desc = [](int type, int idx, int pre, int post, int i, int lwn, int newl, int l) {
    return (unsigned)type | (unsigned)idx << 3 | (unsigned)pre << 8 | (unsigned)post << 10 | (unsigned)i << 12 | (unsigned)lwn << 16 | 1u << 17 |
           (unsigned)newl << 18 | (unsigned)l << 24;
  };
  for (int l = 0; l < n_total; ++l) {
    const int i = l % wn_layers;
    const bool first = i == 0, last = i == wn_layers - 1;
    const int nh_in = l == 0 ? 0 : NH, nh_out = l + 1 < n_total ? NH : 0;
    const size_t start = prog.size();
    for (int c = nh_in; c < ncc; ++c) prog.push_back(desc(CT_COND, c, PRE_NONE, POST_NONE, i, 0, 0, l));
    const int n2 = first ? 1 : NTAPC;
    for (int c = 0; c < n2; ++c)
      prog.push_back(desc(first ? CT_FIRST : CT_TAP, c, c == 0 ? PRE_H : PRE_NONE, c == n2 - 1 ? POST_GATE : POST_NONE, i, 0, 0, l) |
                     (first ? 0u : (unsigned)(c / CPT) << 19 | (unsigned)(c % CPT) << 21));
    for (int c = 0; c < nh_out; ++c) prog.push_back(desc(CT_COND, c, PRE_NONE, POST_NONE, i, 1, 0, l));
    for (int c = 0; c < NRESC; ++c)
      prog.push_back(desc(last ? CT_END : CT_RES, c, c == 0 ? PRE_ACT : PRE_NONE, c == NRESC - 1 ? POST_EPI : POST_NONE, i, 0, 0, l));
    prog[start] |= 1u << 18;
    layer_end.push_back((int)prog.size());
  }
  for (int k = 0; k < 3; ++k) prog.push_back(0u);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct WgpState {
  char* arena = nullptr;
  size_t arena_bytes = 0;
  std::vector<WgpLayer> layers;   // processing order (flow n_flows-1 first)
  std::vector<WgpFlow> flows;
  WgpLayer* layers_dev = nullptr;
  WgpFlow* flows_dev = nullptr;
  int n_cu = 0;
  unsigned long long poll_limit = 0;
  std::vector<unsigned> prog;     // build_program
  std::vector<int> layer_end;     // chunks up to and including layer l
  unsigned* prog_dev = nullptr;
  size_t o_w1[MAXF][8], o_wc[MAXF][8], o_w2[MAXF][8], o_b1[MAXF][8], o_b2[MAXF][8];
};

namespace {
struct WgpLayout {
  int NB, NBc, pitch, Tq, L, La;
  size_t hb, ab, xa, flags, zero_end, melq, aud, z, total;   // byte offsets; [0, zero_end) is zeroed per call
  size_t hb_stride, ab_stride, xa_stride;                    // floats
};
// column-block counts the kernel is instantiated for
int wgp_pick_nb(int T) {
  for (int nb : {7, 9, 10, 11, 12, 13, 16})
    if (T <= 16 * nb) return nb;
  return 0;
}
WgpLayout wgp_layout(const facppg_wg_config& c, int T) {
  WgpLayout w;
  const int P = c.hop_length / 8;
  w.NB = wgp_pick_nb(T); w.NBc = 16 * w.NB; w.pitch = round_up(w.NBc, 64); w.Tq = HQ + w.pitch + HQ;
  w.L = T * P; w.La = round_up(w.L, 64);
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  w.hb_stride = (size_t)64 * P * w.Tq * 4; w.ab_stride = (size_t)64 * P * w.pitch * 4; w.xa_stride = (size_t)4 * P * w.Tq * 4;
  w.hb = take((size_t)std::max(1, c.wn_layers - 1) * w.hb_stride * 4);
  w.xa = take(2 * w.xa_stride * 4);
  w.flags = take((size_t)(2 * P * NSL * NWV + 64) * 4);
  w.zero_end = off;
  w.ab = take((size_t)c.wn_layers * w.ab_stride * 4 + 4096);
  w.melq = take((size_t)round_up(((c.upsample_kernel + c.hop_length - 1) / c.hop_length) * NMEL, KCH) * w.pitch * 4 + 4096);
  w.aud = take((size_t)2 * 8 * w.La * 4);
  w.z = take((size_t)8 * w.L * 4 + 16);
  w.total = off;
  return w;
}
}  // namespace

size_t wgp_arena_bytes(const facppg_wg_config& c, int device) {
  const int P = c.hop_length / 8;
  int n_cu = 0;
  if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
  if (P * NSL > n_cu || c.n_flows * c.wn_layers < 1) return 0;   // one workgroup per CU, all co-resident
  const int kcp = round_up(((c.upsample_kernel + c.hop_length - 1) / c.hop_length) * NMEL, KCH);
  size_t off = 0;
  auto take = [&](size_t bytes) __attribute__((always_inline)) { off += (bytes + 255) / 256 * 256; };
  for (int k = 0; k < c.n_flows; ++k)
    for (int i = 0; i < c.wn_layers; ++i) {
      take((size_t)NSL * (i == 0 ? 3 : NGT) * 256 * sizeof(float4));
      take((size_t)P * NSL * (kcp / 16) * 256 * sizeof(float4));
      if (i < c.wn_layers - 1) take((size_t)NSL * 16 * 128 * sizeof(float4));
      take(NSL * 64 * 4);
      take(NSL * 32 * 4);
    }
  take((size_t)c.n_flows * c.wn_layers * sizeof(WgpLayer));
  take((size_t)c.n_flows * sizeof(WgpFlow));
  take((size_t)c.n_flows * c.wn_layers * 32 * 4 + 64);   // the program: < 32 chunks per layer
  return off;
}

int wgp_create(facppg_wg* h, char* arena, hipStream_t s) {
  (void)arena; (void)s;
  h->wgp = nullptr;
  const size_t bytes = wgp_arena_bytes(h->cfg, h->device);   // (FACPPG_WG_PERSIST=0 is honoured per call: wgp_eligible)
  if (!bytes) return FACPPG_OK;
  WgpState* st = new (std::nothrow) WgpState();
  FACPPG_REQUIRE(st, FACPPG_EINVAL, "out of host memory");
  if (hipMalloc((void**)&st->arena, bytes) != hipSuccess) {
    (void)hipGetLastError();   // not enough memory for the second set of images: the per-layer kernels serve every shape
    delete st;
    return FACPPG_OK;
  }
  st->arena_bytes = bytes;
  (void)hipDeviceGetAttribute(&st->n_cu, hipDeviceAttributeMultiprocessorCount, h->device);
  const facppg_wg_config& c = h->cfg;
  const int P = c.hop_length / 8, kcp = h->kcp;
  size_t off = 0;
  auto take = [&](size_t b) __attribute__((always_inline)) { size_t o = off; off += (b + 255) / 256 * 256; return o; };
  for (int k = 0; k < c.n_flows; ++k)
    for (int i = 0; i < c.wn_layers; ++i) {
      st->o_w1[k][i] = take((size_t)NSL * (i == 0 ? 3 : NGT) * 256 * sizeof(float4));
      st->o_wc[k][i] = take((size_t)P * NSL * (kcp / 16) * 256 * sizeof(float4));
      st->o_w2[k][i] = i < c.wn_layers - 1 ? take((size_t)NSL * 16 * 128 * sizeof(float4)) : 0;
      st->o_b1[k][i] = take(NSL * 64 * 4);
      st->o_b2[k][i] = take(NSL * 32 * 4);
    }
  st->layers_dev = (WgpLayer*)(st->arena + take((size_t)c.n_flows * c.wn_layers * sizeof(WgpLayer)));
  st->flows_dev = (WgpFlow*)(st->arena + take((size_t)c.n_flows * sizeof(WgpFlow)));
  st->prog_dev = (unsigned*)(st->arena + take((size_t)c.n_flows * c.wn_layers * 32 * 4 + 64));
  {
    const char* pl = getenv("FACPPG_POLL_LIMIT");
    const double seconds = pl ? strtod(pl, nullptr) : 20.0;
    int khz = 0;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device);
    st->poll_limit = seconds > 0 && khz > 0 ? (unsigned long long)(seconds * 1e3 * khz) : 0ull;
  }
  h->wgp = st;
  return FACPPG_OK;
}

int wgp_pack_layer(facppg_wg* h, int k, int i, const WgpLayerSrc& src, hipStream_t s) {
  WgpState* st = h->wgp;
  if (!st) return FACPPG_OK;
  const facppg_wg_config& c = h->cfg;
  const int P = c.hop_length / 8, kcp = h->kcp;
  const bool last = i == c.wn_layers - 1;
  float4* w1 = (float4*)(st->arena + st->o_w1[k][i]);
  if (i == 0) k_wgp_pack_w1f<<<(NSL * 3 * 256 + 255) / 256, 256, 0, s>>>(src.f0, w1);
  else k_wgp_pack_w1<<<(NSL * NGT * 256 + 255) / 256, 256, 0, s>>>(src.in_w, w1);
  {
    const size_t n = (size_t)P * NSL * (kcp / 16) * 256;
    k_wgp_pack_wc<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(src.folded, (float4*)(st->arena + st->o_wc[k][i]), P, kcp);
  }
  if (!last) k_wgp_pack_w2<<<(NSL * 16 * 128 + 255) / 256, 256, 0, s>>>(src.rs_w, (float4*)(st->arena + st->o_w2[k][i]));
  k_wgp_pack_bias<<<2, 256, 0, s>>>(src.b1pm, last ? nullptr : src.b2, (float*)(st->arena + st->o_b1[k][i]), (float*)(st->arena + st->o_b2[k][i]));
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

int wgp_finish_create(facppg_wg* h, hipStream_t s) {
  WgpState* st = h->wgp;
  if (!st) return FACPPG_OK;
  const facppg_wg_config& c = h->cfg;
  const int nf = c.n_flows;
  int early_seen = 0;
  for (int fi = 0; fi < nf; ++fi) {
    const int k = nf - 1 - fi;
    WgpFlow f;
    memset(&f, 0, sizeof(f));
    f.endb = h->endb[k]; f.winv = h->winv[k]; f.start_w = h->start_w[k]; f.start_b = h->start_b[k];
    f.n_half = h->n_half[k]; f.early = h->early[k];
    f.swap = c.alternate_halves && (k & 1);
    f.swap_next = c.alternate_halves && k > 0 && ((k - 1) & 1);
    f.final_flow = k == 0;
    f.early_index = early_seen;
    if (h->early[k]) ++early_seen;
    st->flows.push_back(f);
    for (int i = 0; i < c.wn_layers; ++i) {
      WgpLayer L;
      memset(&L, 0, sizeof(L));
      L.w1 = (const float4*)(st->arena + st->o_w1[k][i]);
      L.wc = (const float4*)(st->arena + st->o_wc[k][i]);
      L.w2 = i < c.wn_layers - 1 ? (const float4*)(st->arena + st->o_w2[k][i]) : nullptr;
      L.we = h->we[k][i];
      L.b1 = (const float*)(st->arena + st->o_b1[k][i]);
      L.b2 = (const float*)(st->arena + st->o_b2[k][i]);
      L.flow = fi; L.i = i; L.first = i == 0; L.last = i == c.wn_layers - 1; L.dil = 1 << i;
      st->layers.push_back(L);
    }
  }
  FACPPG_HIP_CHECK(hipMemcpyAsync(st->layers_dev, st->layers.data(), st->layers.size() * sizeof(WgpLayer), hipMemcpyHostToDevice, s));
  FACPPG_HIP_CHECK(hipMemcpyAsync(st->flows_dev, st->flows.data(), st->flows.size() * sizeof(WgpFlow), hipMemcpyHostToDevice, s));
  build_program(nf * c.wn_layers, c.wn_layers, h->kcp / WKCH, st->prog, st->layer_end);
  FACPPG_REQUIRE(st->prog.size() <= (size_t)nf * c.wn_layers * 32 + 16, FACPPG_EUNSUPPORTED, "program of %zu chunks does not fit its table", st->prog.size());
  FACPPG_HIP_CHECK(hipMemcpyAsync(st->prog_dev, st->prog.data(), st->prog.size() * 4, hipMemcpyHostToDevice, s));
  FACPPG_HIP_CHECK(hipStreamSynchronize(s));   // (the host vectors are pageable: the copies above must be done before they change)
  for (const void* fn : {(const void*)k_wg_persist<7>, (const void*)k_wg_persist<9>, (const void*)k_wg_persist<10>, (const void*)k_wg_persist<11>,
                         (const void*)k_wg_persist<12>, (const void*)k_wg_persist<13>, (const void*)k_wg_persist<16>})
    FACPPG_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  return FACPPG_OK;
}

void wgp_destroy(facppg_wg* h) {
  if (!h || !h->wgp) return;
  (void)hipFree(h->wgp->arena);
  delete h->wgp;
  h->wgp = nullptr;
}

// When the persistent launch runs.  FACPPG_WG_PERSIST=1: whenever it can (one utterance, T <= 256 frames, folded flow edges);
// =0: never; unset: where it MEASURES faster than the launch-per-layer sequence (profiles/r04_experiments.txt, one MI355X,
// hop 256): its time grows with the column blocks ceil(T/16) -- 5.2 ms at 7, 6.4 at 10, 7.9 at 13, 9.1 at 16 -- while the
// per-layer launches cost 4.8 ms up to 128 frames (16-frame tiles) and 7.7 ms from 129 to 256 (one 32-frame tile per CU,
// whatever T): it wins for 128 < T <= 192 frames, by 5-17 %.
bool wgp_eligible(const facppg_wg* h, int B, int T, const int32_t* T_valid_dev) {
  if (!h->wgp || B != 1 || T_valid_dev || wgp_pick_nb(T) == 0) return false;
  if (getenv("FACPPG_WN_TILE") || getenv("FACPPG_WG_UNFOLDED")) return false;   // a forced per-layer kernel shape
  const char* fold_env = getenv("FACPPG_WG_EDGE_FOLD");
  if (fold_env && atoi(fold_env) == 0) return false;
  const char* env = getenv("FACPPG_WG_PERSIST");     // (read per call: the tests flip it)
  if (env) return atoi(env) != 0;
  return T > 128 && T <= 192;
}

size_t wgp_workspace_bytes(const facppg_wg* h, int B, int T) {
  if (!h->wgp || B != 1 || wgp_pick_nb(T) == 0) return 0;
  return wgp_layout(h->cfg, T).total;
}

int wgp_infer(facppg_wg* h, const float* mel_dev, const float* z_dev, uint64_t seed, float sigma, int T, float* audio_dev,
              char* ws, hipStream_t s) {
  WgpState* st = h->wgp;
  const facppg_wg_config& c = h->cfg;
  const WgpLayout w = wgp_layout(c, T);
  const int P = c.hop_length / 8;
  FACPPG_HIP_CHECK(hipMemsetAsync(ws, 0, w.zero_end, s));   // zero margins / dead columns of h and xa; flags and the arrival counter
  float* melq = (float*)(ws + w.melq);
  k_wgp_melq<<<(h->kcp * w.pitch + 255) / 256, 256, 0, s>>>(mel_dev, melq, T, w.pitch, h->kc, h->kcp);
  const float* z = z_dev;
  if (!z) {
    wg_launch_noise((float*)(ws + w.z), (size_t)8 * w.L, seed, s);
    z = (const float*)(ws + w.z);
  }
  WgpArgs a;
  memset(&a, 0, sizeof(a));
  a.layers = st->layers_dev; a.flows = st->flows_dev;
  a.n_total = c.n_flows * c.wn_layers; a.wn_layers = c.wn_layers;
  if (const char* env = getenv("FACPPG_WGP_LAYERS")) {   // debugging: stop after this many layers (buffers keep their contents)
    const int v = atoi(env);
    if (v > 0 && v < a.n_total) a.n_total = v;
  }
  a.T = T; a.P = P; a.Tq = w.Tq; a.ncc = h->kcp / WKCH; a.ngc = h->kcp / 16;
  a.hb = (float*)(ws + w.hb); a.ab = (float*)(ws + w.ab); a.xa = (float*)(ws + w.xa);
  a.melq = melq; a.aud = (float*)(ws + w.aud); a.z = z; a.audio = audio_dev;
  a.flag_h = (unsigned*)(ws + w.flags); a.flag_a = a.flag_h + P * NSL * NWV; a.bar = a.flag_a + P * NSL * NWV;
  a.hb_stride = (long long)w.hb_stride; a.ab_stride = (long long)w.ab_stride; a.xa_stride = (long long)w.xa_stride;
  a.sigma = sigma; a.La = w.La; a.L = w.L; a.n_rem_first = h->n_rem[c.n_flows - 1];
  a.swap_begin = c.alternate_halves && ((c.n_flows - 1) & 1);
  a.xcd_map = P % 8 == 0;
  a.poll_limit = st->poll_limit;
  a.prog = st->prog_dev; a.n_chunks = st->layer_end[a.n_total - 1];
  const unsigned grid = (unsigned)(P * NSL);
  const size_t lds = std::max((size_t)82 * 1024, (size_t)2 * QC * w.pitch * 16 + 4096 + 16384 + ((size_t)a.n_chunks + 3 + 63) / 64 * 256);
  FACPPG_REQUIRE(lds <= 160 * 1024, FACPPG_EUNSUPPORTED, "persistent launch needs %zu bytes of LDS", lds);
  if (h->profiling) {
    if (h->profiling <= 1) h->ev_used = 0;
    while (h->ev.size() < (size_t)h->ev_used + 2) {
      hipEvent_t ev;
      FACPPG_HIP_CHECK(hipEventCreate(&ev));
      h->ev.push_back(ev);
    }
    FACPPG_HIP_CHECK(hipEventRecord(h->ev[h->ev_used++], s));
  }
  // All P * NSL workgroups hand over to one another through flags and must be co-resident (one per CU: each holds most of a
  // CU's LDS): a COOPERATIVE launch, which checks that the grid fits and never shares the device with another cooperative
  // launch -- two of these in flight at once (two threads or streams serving batch-1 requests) could otherwise each hold part
  // of the CUs and wait for workgroups that were never dispatched.  FACPPG_COOP_PLAIN=1: an ordinary launch (profilers).
  const void* fn = nullptr;
  switch (w.NB) {
    case 7: fn = (const void*)k_wg_persist<7>; break;
    case 9: fn = (const void*)k_wg_persist<9>; break;
    case 10: fn = (const void*)k_wg_persist<10>; break;
    case 11: fn = (const void*)k_wg_persist<11>; break;
    case 12: fn = (const void*)k_wg_persist<12>; break;
    case 13: fn = (const void*)k_wg_persist<13>; break;
    default: fn = (const void*)k_wg_persist<16>; break;
  }
  {
    void* args[] = {(void*)&a};
    const char* plain = getenv("FACPPG_COOP_PLAIN");
    if (plain && atoi(plain) != 0) FACPPG_HIP_CHECK(hipLaunchKernel(fn, dim3(grid), dim3(512), args, lds, s));
    else FACPPG_HIP_CHECK(hipLaunchCooperativeKernel(fn, dim3(grid), dim3(512), args, lds, s));
  }
  h->ev_layers = h->cfg.n_flows * h->cfg.wn_layers;   // (the event pair brackets the whole network: facppg_wg_last_layer_ms divides by this)
  if (h->profiling) FACPPG_HIP_CHECK(hipEventRecord(h->ev[h->ev_used++], s));
  h->last_tile = w.NBc; h->last_waves = 8; h->last_tiles = (int)grid;
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

}  // namespace facppg
