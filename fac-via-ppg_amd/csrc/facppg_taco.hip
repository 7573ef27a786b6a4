// Tacotron2-style PPG -> mel inference on gfx950: encoder, autoregressive decoder, postnet.
//
// Replaces src/common/model.py of the reference: Prenet (:124-135), Encoder.inference (:237-249),
// LocationLayer/Attention (:44-121), Decoder.inference/decode (:489-535, :387-442), Postnet
// (:138-184), Tacotron2.inference (:597-610) and the attention window mask
// get_mask_from_lengths_window_and_time_step (src/common/utils.py:46-78).
//
// Layout: activations are channel-major [B][C][T] (time contiguous), so every Linear/Conv1d of
// the encoder and postnet is one exact-fp32 MFMA tapped GEMM (facppg_gemm) with the bias, eval
// BatchNorm (as per-row scale/shift), ReLU/tanh, dropout mask and residual fused in its epilogue.
// The two sequential parts run as persistent kernels with a device-side loop (the reference syncs the
// host at least twice per frame: the stop test model.py:524 and the Python mask builder); the stop
// decision is taken on the device and the attention is evaluated only on the +-window positions the
// reference's mask keeps (<= 2W+1 of Tin).  Each has a throughput shape and latency shapes:
//   k_bilstm / k_bilstm_coop                       encoder BiLSTM recurrence (input projections were one GEMM)
//   k_decoder / k_decoder_coop / k_decoder_split   the autoregressive decoder loop
// The latency shapes spread one utterance over many co-resident workgroups (cooperative launch) that
// exchange hidden state through 8-byte {value, tag} words; _split and _bilstm_coop keep their weight
// slices in registers for the whole sequence.
// Recurrent weights are stored K-major ([k][rows], rows padded to 4) so a thread streams float4
// columns with fully coalesced 16-byte loads; a 1200x1200 step is 900 threads x 400 loads.

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "facppg_gemm.h"

using namespace facppg;

struct facppg_taco {
  facppg_taco_config c;
  int device;
  int coop_limit;   // workgroups the cooperative (co-resident) kernels may use: from the occupancy calculator, see facppg_taco_create
  int decoder_wg_limit;   // facppg_taco_set_decoder_workgroups: a tighter bound for the decoder alone (0 = none)
  int wall_khz;              // the device's constant-rate clock (facppg_taco_collect_frames bounds its wait in it)
  unsigned long long* frame_stream;   // facppg_taco_set_frame_stream: tagged mel frames as the split decoder emits them (B = 1), or null
  int frame_stream_frames;
  int last_streamed;         // the most recent decode published its frames there
  int last_mode, last_wgs;   // facppg_taco_last_decoder_launch: 0 one workgroup per utterance, 1 cooperative, 2 split; workgroups launched
  char* arena;
  // encoder
  float4 *pre0, *pre1, *conv[8], *wih;
  float *conv_b[8], *conv_scale[8], *conv_shift[8];
  float *whh_t[2], *lstm_b;        // [2][H][4H] k-major, bias [2][4H]
  float4* mem_w;                   // memory_layer packed
  // decoder (k-major, rows padded to a multiple of 4)
  float *dp0_t, *dp1_t, *att_t, *att_b, *dec_t, *dec_b, *q_t, *proj_t, *proj_b;
  float *loc_conv, *loc_dense, *v;
  // per-workgroup LSTM slices [NWG][K][4U] for k_decoder_coop, packed for three slice widths:
  // U = 8 (38 workgroups/utterance, B <= 6), 20 (15, B <= 16), 40 (8, B <= 30), 75 (4, B <= 60), 150 (2, B <= 120)
  float *att_coop[5], *dec_coop[5];
  int coop_U[5], coop_nwg[5];
  float *att_w4, *dec_w4;          // 4-unit slices for k_decoder_split's workers
  float *w1p_t, *b1p;              // prenet layer 1 composed with the projection: [D+E][Pp] k-major, [Pp]
  int split_nwk;
  // postnet
  float4* post[8];
  float *post_b[8], *post_scale[8], *post_shift[8];
};

namespace {

// The cooperating kernels (decoder, BiLSTM) need all their workgroups co-resident; hipLaunchCooperativeKernel checks that the grid
// fits and the launches below size their grids from the occupancy calculator.  FACPPG_COOP_PLAIN=1 launches the SAME kernels with
// the same grids as ordinary launches (read per call): rocprofv3 on this stack dies on cooperative launches, and an ordinary
// launch of a grid that fits an otherwise idle GPU is co-resident all the same (every poll is bounded in wall-clock time:
// a launch that did not fit traps instead of hanging) -- the profiling mode of tools/profile_coop.sh, not a production setting.
hipError_t launch_coop(const void* fn, dim3 grid, dim3 block, void** args, size_t smem, hipStream_t s) {
  const char* plain = getenv("FACPPG_COOP_PLAIN");
  if (plain && atoi(plain) != 0) return hipLaunchKernel(fn, grid, block, args, smem, s);
  return hipLaunchCooperativeKernel(fn, grid, block, args, smem, s);
}

constexpr int NT = 1024;  // threads of the one-workgroup-per-utterance persistent kernels (16 waves, <= 128 VGPRs)
constexpr int NTC = 512;  // threads of a cooperative decoder workgroup (8 waves, <= 256 VGPRs: no spills, deeper loads in flight)

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }
// tanh through one hardware exponential and one reciprocal (relative error ~1e-6; |x| > 9.02 is +-1 in
// fp32, so clamping to +-15 changes nothing): libm's tanhf is ~40 instructions per value
__device__ __forceinline__ float tanh_fast(float x) {
  const float e = __expf(-2.0f * fminf(fmaxf(x, -15.0f), 15.0f));
  return __fdividef(1.0f - e, 1.0f + e);
}

// dst[k_off + k][Rp] = src[r][k]  (k-major transpose with row padding)
__global__ void k_transpose_pad(const float* __restrict__ src, float* __restrict__ dst, int R, int K, int Rp, int k_off) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * Rp) return;
  const int k = i / Rp, r = i % Rp;
  dst[(size_t)(k_off + k) * Rp + r] = r < R ? src[(size_t)r * K + k] : 0.0f;
}
__global__ void k_add2(const float* a, const float* b, float* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = a[i] + b[i];
}
__global__ void k_bn_fold(const float* w, const float* b, const float* mean, const float* var, float eps, float* scale,
                          float* shift, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float s = w[i] / sqrtf(var[i] + eps);
    scale[i] = s;
    shift[i] = b[i] - mean[i] * s;
  }
}

// coop slices: dst[wg][k][g*U + j] = src_t[k][g*A + wg*U + j]  (0 beyond A)
__global__ void k_pack_coop(const float* __restrict__ src_t, float* __restrict__ dst, int K, int A, int U, int NWG) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int SC = 4 * U;
  if (i >= (size_t)NWG * K * SC) return;
  const int col = (int)(i % SC), k = (int)((i / SC) % K), wg = (int)(i / ((size_t)SC * K));
  const int u = wg * U + col % U;
  dst[i] = u < A ? src_t[(size_t)k * 4 * A + (col / U) * A + u] : 0.0f;
}

// Prenet layer 1 applied to the projection's output is linear in [dh | ctx] until its ReLU:
//   W1 (Wp v + bp) = (W1 Wp) v + W1 bp.   w1p_t[k][r] = sum_m proj_t[k][m] * dp0_t[m][r]  (row KP = the bias term)
// The split decoder's workers use it to start the prenet straight from the decoder state, without waiting
// for the mel frame to cross workgroups.
__global__ void k_fold_prenet(const float* __restrict__ proj_t, const float* __restrict__ proj_b, const float* __restrict__ dp0_t,
                              float* __restrict__ w1p_t, float* __restrict__ b1p, int KP, int NF, int NFp, int Pp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (KP + 1) * Pp) return;
  const int k = i / Pp, r = i % Pp;
  const float* a = k < KP ? proj_t + (size_t)k * NFp : proj_b;
  float v = 0.0f;
  for (int m = 0; m < NF; ++m) v = fmaf(a[m], dp0_t[(size_t)m * Pp + r], v);
  if (k < KP) w1p_t[i] = v;
  else b1p[r] = v;
}

// Philox-free cheap keep-mask: one 64-bit SplitMix hash per element (p = 0.5 Bernoulli).
__global__ void k_random_mask(uint8_t* __restrict__ out, size_t n, uint64_t seed) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  out[i] = (uint8_t)((z >> 40) & 1);
}

// Per-utterance keep-masks: bit (layer j, channel, frame) of utterance b is a function of seeds[b] alone, so a
// padded batch drawn with per-utterance seeds reproduces each utterance's own batch-1 draw.
__device__ __forceinline__ uint8_t mask_bit(uint64_t seed, uint64_t index) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (index + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint8_t)((z >> 40) & 1);
}
// enc [2][B][S][Tin]
__global__ void k_random_mask_enc_utt(uint8_t* __restrict__ out, const uint64_t* __restrict__ seeds, int B, int S, int Tin) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)2 * B * S * Tin) return;
  const int t = (int)(i % Tin), ch = (int)(i / Tin % S), b = (int)(i / Tin / S % B), j = (int)(i / Tin / S / B);
  out[i] = mask_bit(seeds[b], ((uint64_t)(j * S + ch) << 32) | (uint32_t)t);
}
// dec [steps][2][B][P]
__global__ void k_random_mask_dec_utt(uint8_t* __restrict__ out, const uint64_t* __restrict__ seeds, int B, int P, int steps) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)steps * 2 * B * P) return;
  const int u = (int)(i % P), b = (int)(i / P % B), j = (int)(i / P / B % 2), t = (int)(i / P / B / 2);
  out[i] = mask_bit(seeds[b] ^ 0xD1B54A32D192ED03ull, ((uint64_t)t << 32) | (uint32_t)(j * P + u));
}

// partial matvec: part[ks][Rp] = sum_{k in split ks} WT[k][Rp] * v[k]; thread = (slot of 4 rows, k split)
// UB = weight loads kept in flight per thread (the stream is L2-latency bound: bytes in flight decide)
template <int UB = 4>
__device__ __forceinline__ void matvec_part(const float* __restrict__ WT, int K, int Rp, int KS, const float* v, float* part,
                                            int tid) {
  const int ns = Rp >> 2;
  const int slot = tid % ns, ks = tid / ns;
  if (ks >= KS) return;
  const int k0 = (int)((long)ks * K / KS), k1 = (int)((long)(ks + 1) * K / KS);
  const float4* w = reinterpret_cast<const float4*>(WT) + slot;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int kb = k0; kb < k1; kb += UB) {
    float4 wv[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) wv[u] = w[(size_t)min(kb + u, k1 - 1) * ns];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const float vk = kb + u < k1 ? v[kb + u] : 0.0f;
      acc.x = fmaf(wv[u].x, vk, acc.x); acc.y = fmaf(wv[u].y, vk, acc.y);
      acc.z = fmaf(wv[u].z, vk, acc.z); acc.w = fmaf(wv[u].w, vk, acc.w);
    }
  }
  reinterpret_cast<float4*>(part)[ks * ns + slot] = acc;
}
template <int NT>
__device__ __forceinline__ int pick_ks(int Rp, int K) {
  int ks = NT / (Rp >> 2);
  if (ks > K) ks = K;
  return ks < 1 ? 1 : ks;
}
__device__ __forceinline__ float part_sum(const float* part, int Rp, int KS, int r) {
  float s = 0.0f;
#pragma unroll 8
  for (int i = 0; i < KS; ++i) s += part[i * Rp + r];
  return s;
}

// ------------------------------------------------------------------------------------------
// Encoder BiLSTM recurrence (nn.LSTM, gate order i,f,g,o; model.py:211-213, 246-247).
// xproj[b][t][dir*4H + row] = W_ih x_t + b_ih + b_hh (from the GEMM).  grid = (2, B).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_bilstm(const float* __restrict__ xproj, const float* __restrict__ whh_t0,
                                               const float* __restrict__ whh_t1, const int* __restrict__ lengths, int Tin, int H,
                                               float* __restrict__ mem_tm /*[B][Tin][2H]*/, float* __restrict__ mem_cm /*[B][2H][Tin]*/) {
  extern __shared__ float sm[];
  const int dir = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int R = 4 * H;
  float* hv = sm;             // [H]
  float* cv = hv + H;         // [H]
  float* part = cv + H;       // [KS][R]
  const float* WT = dir ? whh_t1 : whh_t0;
  const int len = lengths ? lengths[b] : Tin;
  const int KS = pick_ks<NT>(R, H);
  for (int i = tid; i < H; i += NT) { hv[i] = 0.0f; cv[i] = 0.0f; }
  __syncthreads();
  for (int s = 0; s < len; ++s) {
    const int t = dir ? len - 1 - s : s;
    matvec_part(WT, H, R, KS, hv, part, tid);
    __syncthreads();
    if (tid < H) {
      const float* xp = xproj + ((size_t)b * Tin + t) * (2 * R) + dir * R;
      const float gi = part_sum(part, R, KS, tid) + xp[tid];
      const float gf = part_sum(part, R, KS, H + tid) + xp[H + tid];
      const float gg = part_sum(part, R, KS, 2 * H + tid) + xp[2 * H + tid];
      const float go = part_sum(part, R, KS, 3 * H + tid) + xp[3 * H + tid];
      const float c = sigm(gf) * cv[tid] + sigm(gi) * tanhf(gg);
      const float h = sigm(go) * tanhf(c);
      cv[tid] = c; hv[tid] = h;
      mem_tm[((size_t)b * Tin + t) * (2 * H) + dir * H + tid] = h;
      mem_cm[((size_t)b * 2 * H + dir * H + tid) * Tin + t] = h;
    }
    __syncthreads();
  }
}

// Latency shape of the same recurrence (few utterances): NWG workgroups per (utterance, direction),
// each owning BU hidden units = 4*BU gate rows of W_hh, held IN REGISTERS for the whole sequence
// (4*BU rows x H/4 columns per thread = 75 floats at H = 300) -- a step streams nothing but the
// 4*BU input-projection values.  The new hidden values cross workgroups as 8-byte {value, step tag}
// words (sc1 stores, polled sc1 loads; see coop_gather), double-buffered by step parity: a word of
// step s+2 may overwrite step s only after its owner gathered all of step s+1, and every reader
// published its step-s+1 word after consuming step s.  Fixed-order reductions: bit-identical to the
// one-workgroup kernel's sums up to fp32 re-association of the K split.
// Poll an exchange word until its tag shows up.  The producers are co-resident by construction
// (hipLaunchCooperativeKernel refuses a grid that is not), so the wait is a few microseconds.
// Run-time bound: a lost producer (a logic error, or a grid that was not co-resident after all) turns into a trapped
// kernel and a HIP error instead of a hung GPU.  The bound is WALL-CLOCK time on the device's constant-rate counter
// (wall_clock64), checked every 1024 polls, so a slow but live run -- a debugger, a profiler, a time-sliced or SR-IOV
// GPU -- is not turned into a fatal one the way a poll count would.  PROCESS-WIDE (one device constant per loaded code
// object): FACPPG_POLL_LIMIT seconds, read when the first handle is created; default 20 s; 0 = never trap.
__constant__ unsigned long long g_poll_limit_ticks = 0;   // 0 until the first facppg_taco_create: unbounded
__device__ __forceinline__ unsigned long long poll_tag(const unsigned long long* w, unsigned tag) {
  unsigned long long v;
  const unsigned long long limit = g_poll_limit_ticks;
  unsigned long long t0 = 0;
  unsigned spins = 0;
#pragma unroll 1
  do {
    v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((++spins & 1023u) == 0 && limit) {
      const unsigned long long now = wall_clock64();
      if (!t0) t0 = now;
      else if (now - t0 > limit) __builtin_trap();
    }
  } while ((unsigned)(v >> 32) != tag);
  return v;
}

// BU = hidden units per workgroup, BKR = max columns of a gate row held per thread (NTC / (4*BU) K parts):
// <32, 96>: 10 workgroups per direction at H = 300 (H <= 384), B <= 12;  <64, 152>: 5 per direction (H <= 304), B <= 24.
template <int BU, int BKR>
__global__ __launch_bounds__(NTC) void k_bilstm_coop(const float* __restrict__ xproj, const float* __restrict__ whh_t0,
                                                     const float* __restrict__ whh_t1, const int* __restrict__ lengths, int Tin,
                                                     int H, unsigned long long* __restrict__ xchg /*[B][2][2][H]*/,
                                                     float* __restrict__ mem_tm, float* __restrict__ mem_cm) {
  constexpr int BKP = NTC / (4 * BU);    // K parts per gate row
  __shared__ float hv[BKP * BKR];        // gathered hidden vector [H]
  __shared__ float part[BKP][4 * BU];    // K-part partial sums per gate row
  __shared__ float gates[4 * BU];
  const int wg = blockIdx.x, dir = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int R = 4 * H, KR = (H + BKP - 1) / BKP;
  const int r = tid % (4 * BU), kp = tid / (4 * BU);
  const int u = wg * BU + r % BU, row = (r / BU) * H + u;    // this thread's gate row of W_hh
  const float* WT = dir ? whh_t1 : whh_t0;                   // [H][4H] k-major
  const int len = lengths ? lengths[b] : Tin;
  float w[BKR];
#pragma unroll
  for (int i = 0; i < BKR; ++i) {
    const int k = kp * KR + i;
    w[i] = (i < KR && k < H && u < H) ? WT[(size_t)k * R + row] : 0.0f;
  }
  for (int i = tid; i < BKP * BKR; i += NTC) hv[i] = 0.0f;
  float c = 0.0f;
  unsigned long long* xb = xchg + (size_t)(b * 2 + dir) * 2 * H;
  __syncthreads();
  for (int s = 0; s < len; ++s) {
    const int t = dir ? len - 1 - s : s;
    // input projection of this row (independent of h): issued first, consumed after the matvec
    const float xp = (kp == 0 && u < H) ? xproj[((size_t)b * Tin + t) * (2 * R) + dir * R + row] : 0.0f;
    float acc = 0.0f;
    const float* hk = hv + kp * KR;
#pragma unroll
    for (int i = 0; i < BKR; ++i) acc = fmaf(w[i], hk[i], acc);   // hk beyond KR multiplies zero weights
    part[kp][r] = acc;
    __syncthreads();
    if (kp == 0) {
      float g = xp;
#pragma unroll
      for (int j = 0; j < BKP; ++j) g += part[j][r];
      gates[r] = g;
    }
    __syncthreads();
    if (tid < BU && wg * BU + tid < H) {
      const float cn = sigm(gates[BU + tid]) * c + sigm(gates[tid]) * tanhf(gates[2 * BU + tid]);
      const float h = sigm(gates[3 * BU + tid]) * tanhf(cn);
      c = cn;
      const int uu = wg * BU + tid;
      __hip_atomic_store(xb + (size_t)((s + 1) & 1) * H + uu, ((unsigned long long)(unsigned)(s + 1) << 32) | __float_as_uint(h),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      mem_tm[((size_t)b * Tin + t) * (2 * H) + dir * H + uu] = h;
      mem_cm[((size_t)b * 2 * H + dir * H + uu) * Tin + t] = h;
    }
    if (s + 1 < len) {
      const unsigned long long* src = xb + (size_t)((s + 1) & 1) * H;
      for (int i = tid; i < H; i += NTC) {
        hv[i] = __uint_as_float((unsigned)poll_tag(src + i, (unsigned)(s + 1)));
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// Decoder.  Two launch shapes share the per-step building blocks below:
//   k_decoder       one workgroup per utterance (throughput mode: large batches; B workgroups)
//   k_decoder_coop  NWG workgroups per utterance (latency mode: small batches).  Every workgroup
//                   redundantly runs the small parts of a step (projection, prenet, attention) and
//                   owns U of the LSTM units of both LSTMCells: it streams only its 4U-row slice
//                   of the two 1200x1200 recurrent matrices (packed per workgroup, L2 resident)
//                   and the hidden vectors are exchanged through global memory with two grid
//                   barriers per frame.  Because the redundant parts are computed with identical
//                   code on identical data, every workgroup takes the same stop decision.
// ------------------------------------------------------------------------------------------
struct DecArgs {
  const float *dp0_t, *dp1_t, *att_t, *att_b, *dec_t, *dec_b, *q_t, *proj_t, *proj_b, *loc_conv, *loc_dense, *v;
  const float *att_coop, *dec_coop;  // [NWG][K][4U] slices (coop mode)
  unsigned long long* xchg;   // [B][2][A] {value, frame tag} hidden-state exchange words (coop mode)
  const float *att_w4, *dec_w4;   // [NWK][K][16] slices of 4 units (split mode workers)
  const float *w1p_t, *b1p;       // prenet-1 o projection (split mode workers)
  unsigned long long* xsplit;     // [B][NF + 1 + 2P + A + E + D + 8] exchange words of the split decoder
  long long* prof;       // optional [8] phase cycle counters (debug; FACPPG_DECODER_PROF=1)
  const float* memory;   // [B][Tin][E]
  const float* pm;       // [B][AD][Tin]
  const int* lengths;    // [B] or null
  const int* step_limits;   // [B] or null: utterance b stops after min(step_limits[b], max_steps) frames at the latest
  const uint8_t* masks;  // [steps][2][B][P]
  float* mel;            // [B][NF][max_steps]
  float* gate;           // [B][max_steps]
  float* align;          // [B][max_steps][Tin] or null
  int* out_len;          // [B]
  int B, Tin, E, P, A, D, AD, NF, NFIL, KSZ, window, max_steps, U;
  int b0;                // k_decoder_coop: first utterance of this launch (large batches run in chunks)
  float gate_thr;
  // k_decoder_split: worker workgroups per group
  int nwk, dbg_flags;
  // k_decoder_split, B = 1: every mel value of frame t is ALSO published as a {value, t + 1} word at melx[t * NF + row] the moment
  // its projection row is formed (agent-scope store: visible to kernels on other streams while this launch is still running --
  // the plain stores to `mel` are only guaranteed visible once it has ended); facppg_taco_collect_frames reads them
  unsigned long long* melx;
};

// The frame count at which utterance b stops if its gate never fires (model.py:524-528's max_decoder_steps,
// optionally tightened per utterance).
__device__ __forceinline__ int dec_step_limit(const DecArgs& p, int b) {
  return p.step_limits ? min(max(p.step_limits[b], 1), p.max_steps) : p.max_steps;
}

struct DecLds {
  float *in_att, *in_dec, *in_proj, *ac, *dc, *xin, *p1, *pq, *part, *feat, *lconv, *ldense, *vv, *wprev, *wcum, *en;
  int KA, KD, KP, ADp, NFp, Pp, G, AD32;
};

__host__ __device__ inline size_t dec_lds_floats(int P, int E, int A, int D, int NF, int AD, int NFIL, int KSZ, int Tin) {
  // feat [32][64], lconv^T [2*KSZ (even)][32], ldense [32][ADp32], v / processed query [ADp32]
  (void)NFIL;
  const size_t ADp32 = round_up(AD, 32);
  return (size_t)(P + E + A) + (A + E + D) + (D + E) + A + D + round_up(NF, 4) + round_up(P, 4) + ADp32 + 4096 +
         64 * 32 + (size_t)round_up(2 * KSZ, 2) * 32 + 32 * ADp32 + ADp32 + 3 * (size_t)Tin;
}

__device__ __forceinline__ void dec_carve(const DecArgs& p, float* sm, DecLds& L) {
  L.G = 4 * p.A;
  L.KA = p.P + p.E + p.A;   // attention LSTM input: [prenet | ctx | ah]
  L.KD = p.A + p.E + p.D;   // decoder LSTM input:   [ah | ctx | dh]
  L.KP = p.D + p.E;         // projection input:     [dh | ctx]
  L.ADp = round_up(p.AD, 4); L.NFp = round_up(p.NF + 1, 4); L.Pp = round_up(p.P, 4);
  L.AD32 = round_up(p.AD, 32);
  L.in_att = sm;
  L.in_dec = L.in_att + L.KA;
  L.in_proj = L.in_dec + L.KD;
  L.ac = L.in_proj + L.KP;
  L.dc = L.ac + p.A;
  L.xin = L.dc + p.D;
  L.p1 = L.xin + round_up(p.NF, 4);
  L.pq = L.p1 + L.Pp;
  L.part = L.pq + L.AD32;
  L.feat = L.part + 4096;
  L.lconv = L.feat + 64 * 32;
  L.ldense = L.lconv + round_up(2 * p.KSZ, 2) * 32;
  L.vv = L.ldense + 32 * L.AD32;
  L.wprev = L.vv + L.AD32;
  L.wcum = L.wprev + p.Tin;
  L.en = L.wcum + p.Tin;
}

template <int NT>
__device__ __forceinline__ void dec_init(const DecArgs& p, const DecLds& L, float* sm, int tid) {
  for (int i = tid; i < L.KA + L.KD + L.KP + p.A + p.D + round_up(p.NF, 4) + L.Pp + L.AD32; i += NT) sm[i] = 0.0f;
  // MFMA operand images, zero padded to 32 filters / a multiple of 32 attention dims:
  //   lconv^T[kk = c*KSZ + k][f]  (location conv, model.py:49-53), ldense[f][a] (location dense, :54-56)
  const int KK = 2 * p.KSZ;
  for (int i = tid; i < round_up(KK, 2) * 32; i += NT) {
    const int kk = i >> 5, f = i & 31;
    L.lconv[i] = (kk < KK && f < p.NFIL) ? p.loc_conv[f * KK + kk] : 0.0f;
  }
  for (int i = tid; i < 32 * L.AD32; i += NT) {
    const int f = i / L.AD32, a = i % L.AD32;
    L.ldense[i] = (f < p.NFIL && a < p.AD) ? p.loc_dense[a * p.NFIL + f] : 0.0f;
  }
  for (int i = tid; i < L.AD32; i += NT) L.vv[i] = i < p.AD ? p.v[i] : 0.0f;
  for (int i = tid; i < 3 * p.Tin; i += NT) L.wprev[i] = 0.0f;
}

// prenet: 2 x (Linear no bias, ReLU, dropout p=0.5 always on)  xin -> in_att[0:P]   (model.py:132-135)
template <int NT>
__device__ __forceinline__ void dec_prenet(const DecArgs& p, const DecLds& L, int t, int b, int tid) {
  const uint8_t* mk = p.masks + ((size_t)t * 2 * p.B + b) * p.P;
  {
    const int KS = pick_ks<NT>(L.Pp, p.NF);
    matvec_part<(NT <= 512 ? 16 : 4)>(p.dp0_t, p.NF, L.Pp, KS, L.xin, L.part, tid);
    __syncthreads();
    if (tid < p.P) L.p1[tid] = fmaxf(part_sum(L.part, L.Pp, KS, tid), 0.0f) * (float)mk[tid] * 2.0f;
    __syncthreads();
  }
  {
    const int KS = pick_ks<NT>(L.Pp, p.P);
    matvec_part<(NT <= 512 ? 16 : 4)>(p.dp1_t, p.P, L.Pp, KS, L.p1, L.part, tid);
    __syncthreads();
    if (tid < p.P) L.in_att[tid] = fmaxf(part_sum(L.part, L.Pp, KS, tid), 0.0f) * (float)mk[(size_t)p.B * p.P + tid] * 2.0f;
    __syncthreads();
  }
}

// The index range [lo, hi] of encoder frames the attention window mask of decoder step t keeps for an
// utterance of `len` frames (get_mask_from_lengths_window_and_time_step, src/common/utils.py:64-77):
// lo = min(max(0, t - W), len - 1), hi = min(t + W, len - 1) -- so once t - W has passed the end only the
// last frame stays unmasked (the reference's documented quirk, utils.py:65-69); window < 0 = no window
// (attention_window_size None).  The ONE definition every decoder shape and k_window_mask use.
__device__ __forceinline__ void attn_window_range(int window, int t, int len, int* lo, int* hi) {
  if (window >= 0) {
    *lo = min(max(0, t - window), len - 1);
    *hi = min(t + window, len - 1);
  } else {
    *lo = 0;
    *hi = len - 1;
  }
}

// LSTMCell pointwise for unit `u` from the four gate pre-activations (gate order i,f,g,o)
__device__ __forceinline__ float lstm_point(float gi, float gf, float gg, float go, float* c) {
  const float cn = sigm(gf) * (*c) + sigm(gi) * tanhf(gg);
  *c = cn;
  return sigm(go) * tanhf(cn);
}

// Location features of up to 64 window positions from c0 on: featT[f][pos] -> L.feat (uses L.part as
// scratch).  They depend only on the previous frame's attention weights, so the split decoder computes
// them while the attention LSTM's hidden state is still in flight.
template <int NT>
__device__ __forceinline__ void attn_features(const DecArgs& p, const DecLds& L, int c0, int nc, int tid) {
  const int lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
  const int half = (p.KSZ - 1) / 2, KK = 2 * p.KSZ;
  // features: the K = 2*KSZ reduction is dealt to NWF waves per 32-position block (partial products
  // meet in LDS in a fixed order) so the dependent MFMA chain is short
    {
      constexpr int NWF = 2;   // waves 0..3: (position block, K half)
      const int cbf = wave / NWF, part_i = wave % NWF;
      const int nsteps = (KK + 1) / 2, s_lo = part_i * nsteps / NWF, s_hi = (part_i + 1) * nsteps / NWF;
      float* fpart = L.part;   // [2][NWF][32 f][32 pos] = 4096 floats
      if (cbf < 2 && 32 * cbf < nc) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const int q0 = c0 + 32 * cbf + li - half;
        for (int s2 = s_lo; s2 < s_hi; ++s2) {
          const int kk = 2 * s2 + kh;
          const int c = kk >= p.KSZ, q = q0 + kk - c * p.KSZ;
          const float bv = (kk < KK && q >= 0 && q < p.Tin) ? (c ? L.wcum[q] : L.wprev[q]) : 0.0f;
          acc = mfma32x32x2(L.lconv[kk * 32 + li], bv, acc);
        }
        float* dst = fpart + (size_t)(cbf * NWF + part_i) * 1024;   // [f][32]
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(8 * (r >> 2) + (r & 3) + 4 * kh) * 32 + li] = acc[r];
      }
      __syncthreads();
      for (int i = tid; i < 2048; i += NT) {
        const int f = i >> 6, pos = i & 63, cb2 = pos >> 5;
        float v = 0.0f;
        if (32 * cb2 < nc) {
          const float* src = fpart + (size_t)cb2 * NWF * 1024 + f * 32 + (pos & 31);
#pragma unroll
          for (int j = 0; j < NWF; ++j) v += src[j * 1024];
        }
        L.feat[f * 64 + pos] = v;
      }
    }
}

// The part of the energies that does not need the query (model.py:101-104: e = v . tanh(W_q q + W_loc f + processed_memory)):
// this wave's block of pa = ldense . feat (MFMA) and its processed-memory values, for the FIRST 64-position pass of the window.
// The split decoder's main workgroup forms them while the attention LSTM's hidden state is still in flight; dec_attention<PRE>
// then only adds the processed query, takes the tanh and reduces -- the same values through the same expression, so the same bits.
// Waves < NRB: acc / pm of the 16 rows of row block `wave` (column block 0); with a short tail (see dec_attention) the other
// waves hold up to four 16x16 tail tiles (4 values each).  A wave whose share does not fit (second loop iteration, more than
// four tiles) computes the rest in dec_attention as before.
constexpr int PRE_TAIL_TILES = 4;
template <int NT>
__device__ __forceinline__ void attn_energy_pre(const DecArgs& p, const DecLds& L, const float* pm, int c0, int nc, int tid,
                                                float4* __restrict__ stash) {   // LDS, [8][NT] float4: slots 0-3 acc, 4-7 pm
  // (opaque thread index: otherwise hipcc hoists the ~50 per-lane row offsets below out of the frame loop, where they cost
  //  ~100 registers for the whole utterance and spill)
  asm volatile("" : "+v"(tid));
  float pacc[16], ppm[16];
  const int lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
  constexpr int NW = NT / 64;
  const int NRB = L.AD32 / 32;
  const int ntail = nc - 32;
  const bool tail_valu = ntail > 0 && ntail <= 16 && NW > NRB;
#pragma unroll
  for (int r = 0; r < 16; ++r) { pacc[r] = 0.0f; ppm[r] = 0.0f; }
  if (wave < (tail_valu ? NRB : 2 * NRB) && 32 * (tail_valu ? 0 : wave & 1) < nc) {
    const int rb = tail_valu ? wave : wave >> 1, cb = tail_valu ? 0 : wave & 1;
    const int pos = 32 * cb + li;
    const float* pmp = pm + (size_t)(32 * rb + 4 * kh) * p.Tin + c0 + pos;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int a = 32 * rb + 8 * (r >> 2) + (r & 3) + 4 * kh;
      ppm[r] = (pos < nc && a < p.AD) ? pmp[(size_t)(8 * (r >> 2) + (r & 3)) * p.Tin] : 0.0f;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      const int f = 2 * s2 + kh;
      acc = mfma32x32x2(L.ldense[f * L.AD32 + 32 * rb + li], L.feat[f * 64 + pos], acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) pacc[r] = acc[r];
  } else if (tail_valu) {
    const int pl = lane & 15, kq = lane >> 4, pos = 32 + pl;
#pragma unroll
    for (int j = 0; j < PRE_TAIL_TILES; ++j) {
      const int tile = wave - NRB + j * (NW - NRB);
      if (tile * 16 >= p.AD) break;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = 16 * tile + 4 * kq + r;
        ppm[4 * j + r] = (pl < ntail && a < p.AD) ? pm[(size_t)a * p.Tin + c0 + pos] : 0.0f;
      }
      f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2) {
        const int f = 4 * s2 + kq;
        acc = mfma16x16x4(L.ldense[f * L.AD32 + 16 * tile + pl], L.feat[f * 64 + pos], acc);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) pacc[4 * j + r] = acc[r];
    }
  }
  // (kept in LDS, not in registers, across the wait for the workers: the main workgroup already holds the query layer's 96
  //  weight registers there; each thread reads back only what it wrote, so no barrier is involved)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    stash[j * NT + tid] = make_float4(pacc[4 * j], pacc[4 * j + 1], pacc[4 * j + 2], pacc[4 * j + 3]);
    stash[(4 + j) * NT + tid] = make_float4(ppm[4 * j], ppm[4 * j + 1], ppm[4 * j + 2], ppm[4 * j + 3]);
  }
}

// The context's operand, the encoder memory rows of the attention window, RESIDENT IN REGISTERS across frames (split decoder's
// main workgroup).  The window [lo, hi] slides by at most one position per frame, so of its <= 41 rows (600 floats each: 98 KB
// re-read from L2 every frame, with the weights they are multiplied by known only after the softmax) at most one is new.  Wave g
// of the eight keeps the rows q = g (mod 8) -- the same sets, summed in the same ascending order, as the per-frame code's
// "offset (q - lo) mod 8" groups, which it then meets in the same fixed order: the context keeps its bits -- in CTX_SLOTS
// register slots, slot (q / 8) mod CTX_SLOTS: a row is overwritten by row q + 48, when the window (<= 41 wide) has long left
// it.  The entering row is requested at the top of the frame, before the main workgroup waits for the workers.
constexpr int CTX_SLOTS = 6, CTX_CR = 10;   // 6 x 8 = 48 >= 41 + 7 positions; 10 x 64 channels (encoder_embedding_dim <= 640)
struct CtxRows { float v[CTX_SLOTS][CTX_CR]; };
__device__ __forceinline__ bool ctx_rows_usable(const DecArgs& p) {
  return p.window >= 0 && 2 * p.window + 1 <= 8 * CTX_SLOTS - 7 && p.E <= 64 * CTX_CR;
}
template <int S>
__device__ __forceinline__ void ctx_row_load(CtxRows& R, const float* __restrict__ row, int lane, int E) {
#pragma unroll
  for (int r = 0; r < CTX_CR; ++r) {
    const int c = lane + 64 * r;
    R.v[S][r] = c < E ? row[c] : 0.0f;
  }
}
__device__ __forceinline__ void ctx_rows_update(const DecArgs& p, const float* __restrict__ mem, int hi_prev, int hi, int tid, CtxRows& R) {
  asm volatile("" : "+v"(tid));   // (no per-lane offsets hoisted out of the frame loop, see attn_energy_pre)
  const int lane = tid & 63, g = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
  for (int q = hi_prev + 1; q <= hi; ++q) {   // one row per frame once the window is full; hi + 1 rows in frame 0
    if ((q & 7) != g) continue;
    const float* row = mem + (size_t)q * p.E;
    switch ((q >> 3) % CTX_SLOTS) {
      case 0: ctx_row_load<0>(R, row, lane, p.E); break;
      case 1: ctx_row_load<1>(R, row, lane, p.E); break;
      case 2: ctx_row_load<2>(R, row, lane, p.E); break;
      case 3: ctx_row_load<3>(R, row, lane, p.E); break;
      case 4: ctx_row_load<4>(R, row, lane, p.E); break;
      default: ctx_row_load<5>(R, row, lane, p.E); break;
    }
  }
}
// accv[r] += sum over this wave's rows first, first + 8, ... <= hi (ascending), slot of `first` = S0
template <int S0>
__device__ __forceinline__ void ctx_rows_fma(const CtxRows& R, const float* __restrict__ en, int first, int hi, float (&accv)[CTX_CR]) {
#pragma unroll
  for (int k = 0; k < CTX_SLOTS; ++k) {
    const int q = first + 8 * k;
    const float wgt = q <= hi ? en[q] : 0.0f;
#pragma unroll
    for (int r = 0; r < CTX_CR; ++r) accv[r] = fmaf(wgt, R.v[(S0 + k) % CTX_SLOTS][r], accv[r]);
  }
}

// Location-sensitive attention (model.py:63-121) evaluated on the index range the reference's
// window mask keeps (utils.py:64-77).  Reads ah from in_att[P+E:], updates wprev/wcum and writes
// the context into in_att[P:], in_dec[A:], in_proj[D:].
// QR > 0: the query layer's weights arrive in registers (wq[i] = row k0 + i of this thread's k range of W_q^T, requested by the
// caller -- the split decoder's main workgroup -- before it started waiting for the hidden state).  Same products in the same
// order as matvec_part, so the same bits, without the stream's L2 latency between the hidden state and the energies.
struct NoQueryRegs { float4 w[1]; };
template <int NT, int QR = 0, bool PRE = false, bool ROWS = false>
__device__ __forceinline__ void dec_attention(const DecArgs& p, const DecLds& L, const float* mem, const float* pm, int len,
                                              int t, int b, int tid, bool write_out, bool feat_ready = false,
                                              const float4 (&wq)[QR > 0 ? QR : 1] = NoQueryRegs().w,
                                              const float4* __restrict__ stash = nullptr,   // PRE: attn_energy_pre's [8][NT] float4
                                              const CtxRows& rows = CtxRows(), bool rows_ok = false,      // ROWS: the window's memory rows
                                              unsigned long long* ctx_words = nullptr, unsigned ctx_tag = 0) {   // ROWS: publish the context
  if constexpr (PRE) asm volatile("" : "+v"(tid));   // (see attn_energy_pre: no per-lane offsets hoisted out of the frame loop)
  const int lane = tid & 63, wave = tid >> 6;
  long long atk = clock64();
#define APROF(slot)                                                     \
  if (p.prof && write_out && b == 0 && tid == 0) {                       \
    const long long now = clock64();                                     \
    p.prof[slot] += now - atk;                                           \
    atk = now;                                                           \
  }
  const float* ah = L.in_att + p.P + p.E;
  int lo, hi;
  attn_window_range(p.window, t, len, &lo, &hi);
  {
    const int KS = pick_ks<NT>(L.ADp, p.A);
    if constexpr (QR > 0) {
      const int ns = L.ADp >> 2, slot = tid % ns, ks = tid / ns;
      if (ks < KS) {
        const int k0 = (int)((long)ks * p.A / KS), k1 = (int)((long)(ks + 1) * p.A / KS);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < QR; ++i) {
          const float vk = k0 + i < k1 ? ah[k0 + i] : 0.0f;
          acc.x = fmaf(wq[i].x, vk, acc.x); acc.y = fmaf(wq[i].y, vk, acc.y);
          acc.z = fmaf(wq[i].z, vk, acc.z); acc.w = fmaf(wq[i].w, vk, acc.w);
        }
        for (int k = k0 + QR; k < k1; ++k) {      // (layer sizes whose k ranges exceed the prefetched rows: the rest in place)
          const float4 wv = reinterpret_cast<const float4*>(p.q_t)[(size_t)k * ns + slot];
          const float vk = ah[k];
          acc.x = fmaf(wv.x, vk, acc.x); acc.y = fmaf(wv.y, vk, acc.y); acc.z = fmaf(wv.z, vk, acc.z); acc.w = fmaf(wv.w, vk, acc.w);
        }
        reinterpret_cast<float4*>(L.part)[ks * ns + slot] = acc;
      }
    } else {
      matvec_part<(NT <= 512 ? 16 : 4)>(p.q_t, p.A, L.ADp, KS, ah, L.part, tid);
    }
    __syncthreads();
    if (tid < p.AD) L.pq[tid] = part_sum(L.part, L.ADp, KS, tid);
    __syncthreads();
  }
  APROF(8)
  // Location features and energies of up to 64 window positions per pass, as two small fp32 MFMA
  // products (positions are the N dimension; the <= 41-wide window of the reference is one pass):
  //   featT[f][pos] = sum_kk lconvT[kk][f] * wcat[kk / KSZ][pos + kk % KSZ - half]      (waves 0,1)
  //   pa[a][pos]    = sum_f  ldense[f][a] * featT[f][pos]                               (waves rb*2+cb)
  //   e[pos]        = sum_a  v[a] * tanh(pq[a] + pa[a][pos] + pm[pos][a])               (model.py:101-104)
  // The MFMA result layout leaves a lane with 16 rows (a) of one column (pos): the tanh and the
  // v-weighted sum are lane-local, then one shuffle and a fixed-order sum over the row blocks.
  const int NRB = L.AD32 / 32;
  const int li = lane & 31, kh = lane >> 5;
  float* epart = L.part;   // [NRB][64]
  bool fused_softmax = false;
  float e_reg = 0.0f;
  for (int c0 = lo; c0 <= hi; c0 += 64) {
    const int nc = min(64, hi - c0 + 1);
    if (!(feat_ready && c0 == lo)) attn_features<NT>(p, L, c0, nc, tid);
    __syncthreads();
    APROF(9)
    // energies: one (32 attention dims, 32 positions) block per wave and round.  The reference's window is 41
    // positions: a second position block would be 72 % padding and a second round for two of the waves, so
    // when the tail beyond 32 positions is at most 16 the waves without a row block take it as 16x16x4 MFMA
    // tiles and everything finishes in one round.
    constexpr int NW = NT / 64;
    const int ntail = nc - 32;
    const bool tail_valu = ntail > 0 && ntail <= 16 && NW > NRB;
    for (int pr = wave; pr < (tail_valu ? NRB : 2 * NRB); pr += NW) {
      const int rb = tail_valu ? pr : pr >> 1, cb = tail_valu ? 0 : pr & 1;
      if (32 * cb >= nc) continue;
      const int pos = 32 * cb + li;
      // processed-memory values of this lane's 16 rows, fetched up front (L2 latency under the MFMAs)
      float pmv[16];
      f32x16 acc;
      if (PRE && c0 == lo && pr == wave) {   // formed by attn_energy_pre before the hidden state arrived
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 av = stash[j * NT + tid], pv = stash[(4 + j) * NT + tid];
          acc[4 * j] = av.x; acc[4 * j + 1] = av.y; acc[4 * j + 2] = av.z; acc[4 * j + 3] = av.w;
          pmv[4 * j] = pv.x; pmv[4 * j + 1] = pv.y; pmv[4 * j + 2] = pv.z; pmv[4 * j + 3] = pv.w;
        }
      } else {
        const float* pmp = pm + (size_t)(32 * rb + 4 * kh) * p.Tin + c0 + pos;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int a = 32 * rb + 8 * (r >> 2) + (r & 3) + 4 * kh;
          pmv[r] = (pos < nc && a < p.AD) ? pmp[(size_t)(8 * (r >> 2) + (r & 3)) * p.Tin] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
          const int f = 2 * s2 + kh;
          acc = mfma32x32x2(L.ldense[f * L.AD32 + 32 * rb + li], L.feat[f * 64 + pos], acc);
        }
      }
      float e = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int a = 32 * rb + 8 * (r >> 2) + (r & 3) + 4 * kh;
        e = fmaf(L.vv[a], tanh_fast(L.pq[a] + acc[r] + pmv[r]), e);
      }
      e += __shfl_xor(e, 32);
      if (kh == 0) epart[rb * 64 + pos] = e;
    }
    if (tail_valu && wave >= NRB) {
      // tail positions 32 .. 32+ntail-1 as (16 attention dims x 16 positions) tiles of the 16x16x4 MFMA
      const int pl = lane & 15, kq = lane >> 4, pos = 32 + pl;
      float et = 0.0f;
      int tile = wave - NRB;
      if (PRE && c0 == lo) {   // the first PRE_TAIL_TILES tiles were formed by attn_energy_pre
#pragma unroll
        for (int j = 0; j < PRE_TAIL_TILES; ++j) {
          if (tile * 16 >= p.AD) break;
          const float4 av = stash[j * NT + tid], pv = stash[(4 + j) * NT + tid];
          const float a4[4] = {av.x, av.y, av.z, av.w}, p4[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int a = 16 * tile + 4 * kq + r;   // a < AD32: vv is zero past AD
            et = fmaf(L.vv[a], tanh_fast(L.pq[a] + a4[r] + p4[r]), et);
          }
          tile += NW - NRB;
        }
      }
      for (; tile * 16 < p.AD; tile += NW - NRB) {
        float pmv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int a = 16 * tile + 4 * kq + r;
          pmv[r] = (pl < ntail && a < p.AD) ? pm[(size_t)a * p.Tin + c0 + pos] : 0.0f;
        }
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
          const int f = 4 * s2 + kq;
          acc = mfma16x16x4(L.ldense[f * L.AD32 + 16 * tile + pl], L.feat[f * 64 + pos], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int a = 16 * tile + 4 * kq + r;   // a < AD32: vv is zero past AD
          et = fmaf(L.vv[a], tanh_fast(L.pq[a] + acc[r] + pmv[r]), et);
        }
      }
      et += __shfl_xor(et, 16);
      et += __shfl_xor(et, 32);
      if (lane < 16) epart[NRB * 64 + (wave - NRB) * 16 + lane] = et;   // one partial per free wave
    }
    __syncthreads();
    // ROWS (split decoder's main workgroup; the window is one pass): wave 0 forms the energies in registers and goes straight
    // on to the softmax below -- the same sums, maximum, exponentials and quotients, without the round trip through L.en and
    // its barrier
    fused_softmax = ROWS && c0 == lo && hi - lo + 1 <= 64;
    if (tid < nc) {
      float e = 0.0f;
      if (tail_valu && tid >= 32)
        for (int j = 0; j < NW - NRB; ++j) e += epart[NRB * 64 + j * 16 + tid - 32];
      else
        for (int j = 0; j < NRB; ++j) e += epart[j * 64 + tid];
      if (fused_softmax) e_reg = e;
      else L.en[c0 + tid] = e;
    }
    if (!fused_softmax) __syncthreads();
    APROF(10)
  }
  // The context's memory rows (2 rounds of 3 positions per wave cover the reference's 41-wide window) depend on the window only:
  // with PRE they are requested in front of the softmax, and their L2 latency passes under it and the weight update.  (Requested
  // earlier -- behind the query -- one or both rounds spill next to the energies' registers and the frame gets slower: measured.)
  constexpr int CW = 8;    // position groups (with 16 waves, waves w and w+8 share a group and split the channels)
  constexpr int QU = 3;    // positions per wave issued together (2 rounds cover the 41-wide window)
  constexpr int CR = 12 / (NT / 64 / CW);   // 64-channel rounds per wave (E <= 768)
  constexpr int PRE_ROUNDS = PRE && !ROWS ? 2 : 0;   // (ROWS: the rows are in registers already; its fallback loads them in place)
  const int grp = wave & (CW - 1), halfsel = wave / CW;
  float mvp[PRE_ROUNDS > 0 ? PRE_ROUNDS : 1][QU][CR];
  auto request_rows = [&](int rd) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < QU; ++j) {
      const int q = lo + grp + rd * CW * QU + j * CW;
#pragma unroll
      for (int r = 0; r < CR; ++r) {
        const int c = lane + 64 * (halfsel * CR + r);
        mvp[rd][j][r] = (q <= hi && c < p.E) ? mem[(size_t)q * p.E + c] : 0.0f;
      }
    }
  };
  if constexpr (PRE_ROUNDS > 0) { request_rows(0); request_rows(1); }
  if (fused_softmax) {
    if (wave == 0) {   // lane = position lo + lane (the loops below with one element per lane)
      const bool in = lo + lane <= hi;
      float mx = in ? e_reg : -INFINITY;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      const float ex = in ? expf(e_reg - mx) : 0.0f;
      float sum = ex;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      if (in) L.en[lo + lane] = ex / sum;
    }
  } else if (wave == 0) {   // softmax over [lo, hi]; everything else is masked to -inf => weight 0
    float mx = -INFINITY;
    for (int q = lo + lane; q <= hi; q += 64) mx = fmaxf(mx, L.en[q]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.0f;
    for (int q = lo + lane; q <= hi; q += 64) {
      const float ex = expf(L.en[q] - mx);
      L.en[q] = ex;
      sum += ex;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    for (int q = lo + lane; q <= hi; q += 64) L.en[q] = L.en[q] / sum;
  }
  __syncthreads();
  APROF(11)
  for (int q = tid; q < p.Tin; q += NT) {
    const float wv = (q >= lo && q <= hi) ? L.en[q] : 0.0f;
    L.wprev[q] = wv;
    L.wcum[q] += wv;
    if (write_out && p.align) p.align[((size_t)b * p.max_steps + t) * p.Tin + q] = wv;
  }
  APROF(12)
  // context = sum_q w[q] * memory[q][:].  Positions are dealt to 8 waves so the window's global
  // loads are independent and in flight together; the 8 partial vectors meet in LDS (part+feat are
  // free here) and are summed in a FIXED order: the result must be bitwise reproducible, because
  // in coop mode every workgroup recomputes it and must reach the same stop decision.
  {
    float* cpart = L.part;   // [CW][E]  (part + feat = 6144 floats >= 8 * E for E <= 768)
    static_assert(NT / 64 == 2 * CW || NT / 64 == CW, "context code deals positions to 8 waves or wave pairs");
    bool rows_done = false;
    if constexpr (ROWS) {
      if (rows_ok) {   // (workgroup-uniform)
        static_assert(!ROWS || NT / 64 == CW, "resident rows: one wave per residue");
        float ra[CTX_CR];
#pragma unroll
        for (int r = 0; r < CTX_CR; ++r) ra[r] = 0.0f;
        const int g = __builtin_amdgcn_readfirstlane(wave) & 7;
        const int first = lo + ((g - lo) & 7);   // this wave's first row of the window: the group "offset (first - lo)" of the code below
        switch ((first >> 3) % CTX_SLOTS) {
          case 0: ctx_rows_fma<0>(rows, L.en, first, hi, ra); break;
          case 1: ctx_rows_fma<1>(rows, L.en, first, hi, ra); break;
          case 2: ctx_rows_fma<2>(rows, L.en, first, hi, ra); break;
          case 3: ctx_rows_fma<3>(rows, L.en, first, hi, ra); break;
          case 4: ctx_rows_fma<4>(rows, L.en, first, hi, ra); break;
          default: ctx_rows_fma<5>(rows, L.en, first, hi, ra); break;
        }
        const int off = (first - lo) & 7;
#pragma unroll
        for (int r = 0; r < CTX_CR; ++r) {
          const int c = lane + 64 * r;
          if (c < p.E) cpart[off * p.E + c] = ra[r];
        }
        rows_done = true;
      }
    }
    float accv[CR];
#pragma unroll
    for (int r = 0; r < CR; ++r) accv[r] = 0.0f;
    if constexpr (PRE_ROUNDS > 0) {
#pragma unroll
      for (int rd = 0; rd < PRE_ROUNDS; ++rd)
#pragma unroll
        for (int j = 0; j < QU; ++j) {
          const int q = lo + grp + rd * CW * QU + j * CW;
          const float wgt = q <= hi ? L.en[q] : 0.0f;
#pragma unroll
          for (int r = 0; r < CR; ++r) accv[r] = fmaf(wgt, mvp[rd][j][r], accv[r]);
        }
    }
    for (int qb = lo + grp + PRE_ROUNDS * CW * QU; qb <= hi && !rows_done; qb += CW * QU) {
      float mv[QU][CR];
#pragma unroll
      for (int j = 0; j < QU; ++j) {
        const int q = qb + j * CW;
#pragma unroll
        for (int r = 0; r < CR; ++r) {
          const int c = lane + 64 * (halfsel * CR + r);
          mv[j][r] = (q <= hi && c < p.E) ? mem[(size_t)q * p.E + c] : 0.0f;
        }
      }
#pragma unroll
      for (int j = 0; j < QU; ++j) {
        const int q = qb + j * CW;
        const float wgt = q <= hi ? L.en[q] : 0.0f;
#pragma unroll
        for (int r = 0; r < CR; ++r) accv[r] = fmaf(wgt, mv[j][r], accv[r]);
      }
    }
    if (!rows_done) {
#pragma unroll
      for (int r = 0; r < CR; ++r) {
        const int c = lane + 64 * (halfsel * CR + r);
        if (c < p.E) cpart[grp * p.E + c] = accv[r];
      }
    }
    __syncthreads();
    for (int i = tid; i < p.E; i += NT) {
      float s = 0.0f;
#pragma unroll
      for (int j = 0; j < CW; ++j) s += cpart[j * p.E + i];
      L.in_att[p.P + i] = s; L.in_dec[p.A + i] = s; L.in_proj[p.D + i] = s;
      if constexpr (ROWS) {   // the split decoder's main workgroup: the workers wait for exactly this value -- publish it from here
        if (ctx_words) __hip_atomic_store(ctx_words + i, ((unsigned long long)ctx_tag << 32) | __float_as_uint(s), __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
  }
  APROF(13)
#undef APROF
}

// linear projection + gate on [dh | ctx] (model.py:436-441) for frame index t; returns the stop
// decision (model.py:524-528: the stopping frame is kept) through *s_stop.
template <int NT>
__device__ __forceinline__ void dec_project(const DecArgs& p, const DecLds& L, int t, int b, int tid, bool write_out,
                                            int* s_stop) {
  const int KS = pick_ks<NT>(L.NFp, L.KP);
  matvec_part<(NT <= 512 ? 16 : 4)>(p.proj_t, L.KP, L.NFp, KS, L.in_proj, L.part, tid);
  __syncthreads();
  if (tid <= p.NF) {
    const float v = part_sum(L.part, L.NFp, KS, tid) + p.proj_b[tid];
    if (tid < p.NF) {
      L.xin[tid] = v;
      if (write_out) p.mel[((size_t)b * p.NF + tid) * p.max_steps + t] = v;
    } else {
      if (write_out) p.gate[(size_t)b * p.max_steps + t] = v;
      if (sigm(v) > p.gate_thr || t + 1 == dec_step_limit(p, b)) *s_stop = 1;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(NT) void k_decoder(DecArgs p) {
  extern __shared__ float sm[];
  __shared__ int s_stop;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int len = p.lengths ? p.lengths[b] : p.Tin;
  DecLds L;
  dec_carve(p, sm, L);
  dec_init<NT>(p, L, sm, tid);
  if (tid == 0) s_stop = 0;
  __syncthreads();
  const float* mem = p.memory + (size_t)b * p.Tin * p.E;
  const float* pm = p.pm + (size_t)b * p.Tin * p.AD;
  float* ah = L.in_att + p.P + p.E;   // attention hidden lives inside in_att; copied into in_dec[0:A]
  float* dh = L.in_dec + p.A + p.E;   // decoder hidden lives inside in_dec; copied into in_proj[0:D]
  int t = 0;
  for (;; ++t) {
    dec_prenet<NT>(p, L, t, b, tid);
    {  // attention LSTMCell on [prenet | ctx | ah]   (model.py:400-403)
      const int KS = pick_ks<NT>(L.G, L.KA);
      matvec_part<(NT <= 512 ? 16 : 4)>(p.att_t, L.KA, L.G, KS, L.in_att, L.part, tid);
      __syncthreads();
      float hnew = 0.0f;
      if (tid < p.A)
        hnew = lstm_point(part_sum(L.part, L.G, KS, tid) + p.att_b[tid], part_sum(L.part, L.G, KS, p.A + tid) + p.att_b[p.A + tid],
                          part_sum(L.part, L.G, KS, 2 * p.A + tid) + p.att_b[2 * p.A + tid],
                          part_sum(L.part, L.G, KS, 3 * p.A + tid) + p.att_b[3 * p.A + tid], &L.ac[tid]);
      __syncthreads();   // all reads of the old ah (inside in_att) are done
      if (tid < p.A) { ah[tid] = hnew; L.in_dec[tid] = hnew; }
      __syncthreads();
    }
    dec_attention<NT>(p, L, mem, pm, len, t, b, tid, true);
    {  // decoder LSTMCell on [ah | ctx | dh]   (model.py:425-428)
      const int KS = pick_ks<NT>(L.G, L.KD);
      matvec_part<(NT <= 512 ? 16 : 4)>(p.dec_t, L.KD, L.G, KS, L.in_dec, L.part, tid);
      __syncthreads();
      float hnew = 0.0f;
      if (tid < p.D)
        hnew = lstm_point(part_sum(L.part, L.G, KS, tid) + p.dec_b[tid], part_sum(L.part, L.G, KS, p.D + tid) + p.dec_b[p.D + tid],
                          part_sum(L.part, L.G, KS, 2 * p.D + tid) + p.dec_b[2 * p.D + tid],
                          part_sum(L.part, L.G, KS, 3 * p.D + tid) + p.dec_b[3 * p.D + tid], &L.dc[tid]);
      __syncthreads();
      if (tid < p.D) { dh[tid] = hnew; L.in_proj[tid] = hnew; }
      __syncthreads();
    }
    dec_project<NT>(p, L, t, b, tid, true, &s_stop);
    if (s_stop) break;
  }
  if (tid == 0) p.out_len[b] = t + 1;
}

// One LSTMCell slice in coop mode: this workgroup's 4U gate rows (columns g*U + j of the packed
// slice) over the full input vector; returns the new hidden value of unit j in thread j < U.
template <int NT>
__device__ __forceinline__ void coop_lstm_slice(const float* __restrict__ Wslice, const float* __restrict__ bias, int K, int U,
                                                int A, int unit0, const float* in, float* part, float* cstate,
                                                unsigned long long* xchg_out, unsigned tag, int tid) {
  // the slice is a K-major [K][4U] matrix: the same float4-column stream as the other matvecs, with
  // ~20 loads per thread in flight (one L2 round trip covers U = 8)
  const int SC = 4 * U, KS = pick_ks<NT>(SC, K);
  matvec_part<(NT <= 512 ? 16 : 4)>(Wslice, K, SC, KS, in, part, tid);
  __syncthreads();
  for (int c = tid; c < SC; c += NT) {
    const int u = unit0 + c % U;
    part[4 * NT + c] = part_sum(part, SC, KS, c) + (u < A ? bias[(c / U) * A + u] : 0.0f);   // partials fill <= 4*NT floats
  }
  __syncthreads();
  if (tid < U && unit0 + tid < A) {
    const float* gs = part + 4 * NT;
    const float h = lstm_point(gs[tid], gs[U + tid], gs[2 * U + tid], gs[3 * U + tid], &cstate[tid]);
    // publish {value, frame tag} as ONE 8-byte sc1 store: readers poll the word itself (coop_gather)
    __hip_atomic_store(xchg_out + unit0 + tid, ((unsigned long long)tag << 32) | __float_as_uint(h), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Hidden-state exchange between the workgroups of one utterance without a barrier: every element
// is an 8-byte {value, tag} word written once per frame by its owner (sc1 store) and polled by its
// readers (sc1 loads) until the tag of this frame shows up -- one store-to-load latency instead of
// store drain + ticket atomic + counter poll + load.  Reuse is safe because the two exchanges of a
// frame alternate: nobody can publish frame t+1's attention state before having read every
// decoder-state word of frame t, and each of those was written after its owner read frame t's
// attention state (and vice versa).
template <int NT>
__device__ __forceinline__ void coop_gather(const unsigned long long* xchg, unsigned tag, int n, float* dst0, float* dst1, int tid) {
  for (int i = tid; i < n; i += NT) {
    const float h = __uint_as_float((unsigned)poll_tag(xchg + i, tag));
    dst0[i] = h; dst1[i] = h;
  }
  __syncthreads();
}

__global__ __launch_bounds__(NTC) void k_decoder_coop(DecArgs p) {
  extern __shared__ float sm[];
  __shared__ int s_stop;
  __shared__ float c_att[160], c_dec[160];   // U <= 160 units per workgroup
  const int wg = blockIdx.x, b = p.b0 + blockIdx.y, tid = threadIdx.x;
  const bool leader = wg == 0;
  const int len = p.lengths ? p.lengths[b] : p.Tin;
  DecLds L;
  dec_carve(p, sm, L);
  dec_init<NTC>(p, L, sm, tid);
  if (tid == 0) s_stop = 0;
  if (tid < 160) { c_att[tid] = 0.0f; c_dec[tid] = 0.0f; }
  __syncthreads();
  const float* mem = p.memory + (size_t)b * p.Tin * p.E;
  const float* pm = p.pm + (size_t)b * p.Tin * p.AD;
  float* ah = L.in_att + p.P + p.E;
  float* dh = L.in_dec + p.A + p.E;
  unsigned long long* xa = p.xchg + (size_t)b * 2 * p.A;
  unsigned long long* xd = xa + p.A;
  const int SC = 4 * p.U, unit0 = wg * p.U;
  const float* att_slice = p.att_coop + (size_t)wg * L.KA * SC;
  const float* dec_slice = p.dec_coop + (size_t)wg * L.KD * SC;
  long long tk = clock64();
#define PROF(slot)                                                        \
  if (p.prof && leader && b == 0 && tid == 0) {                           \
    const long long now = clock64();                                      \
    p.prof[slot] += now - tk;                                             \
    tk = now;                                                             \
  }
  // Every workgroup of the utterance computes the same stop decision from bit-identical state, so
  // they all leave the loop at the same frame and the barrier population stays NWG.
  for (int t = 0;; ++t) {
    if (t > 0) {   // finish frame t-1: projection, gate, stop decision
      dec_project<NTC>(p, L, t - 1, b, tid, leader, &s_stop);
      if (s_stop) {
        if (leader && tid == 0) p.out_len[b] = t;
        break;
      }
    }
    PROF(0)
    dec_prenet<NTC>(p, L, t, b, tid);
    PROF(1)
    coop_lstm_slice<NTC>(att_slice, p.att_b, L.KA, p.U, p.A, unit0, L.in_att, L.part, c_att, xa, t + 1, tid);
    PROF(2)
    coop_gather<NTC>(xa, t + 1, p.A, ah, L.in_dec, tid);
    PROF(3)
    PROF(4)
    dec_attention<NTC>(p, L, mem, pm, len, t, b, tid, leader);
    PROF(5)
    coop_lstm_slice<NTC>(dec_slice, p.dec_b, L.KD, p.U, p.D, unit0, L.in_dec, L.part, c_dec, xd, t + 1, tid);
    PROF(6)
    coop_gather<NTC>(xd, t + 1, p.D, dh, L.in_proj, tid);
    PROF(7)
  }
#undef PROF
}

// ------------------------------------------------------------------------------------------
// k_decoder_split: the latency shape for up to 3 utterances.  Per utterance, ONE main workgroup
// runs the attention (the only part that needs the encoder memory) and NWK worker workgroups run
// every dense layer of the step, row-sliced, with their weights in REGISTERS for the whole
// utterance: worker w owns LSTM units 4w..4w+3 of both LSTMCells (16 gate rows x K/32 columns per
// thread) and, for w < ceil(rows/16), rows 16w..16w+15 of the projection+gate and of the two prenet
// layers.  A step therefore streams no weights at all; what remains is a chain of tagged-word
// exchanges (see coop_gather; ~1.5 us each):
//   workers: [project rows -> MEL(t) | gather -> stop?]  prenet-1 rows -> X1 | gather | prenet-2 rows
//            -> X2 | gather | attention-LSTM slice -> AH | gather AH, CTX | decoder-LSTM slice -> DH | gather
//   main:    [gate word of MEL(t) -> stop?]  location features (they depend only on the previous
//            weights: overlapped with the whole worker chain) | gather AH | query, energies, softmax,
//            context -> CTX
// Every word is written once per frame and a writer reaches frame t+1 only through gathers that
// required all its readers to have consumed frame t, so single buffers suffice.  Sums are in a fixed
// order and every value has exactly one producer, so all roles see identical bits and stop together.
// ------------------------------------------------------------------------------------------
constexpr int SU = 4, SSC = 4 * SU, SKP = NTC / SSC;   // 16 rows x 32 K parts per workgroup
constexpr int SKR_LSTM = 40, SKR_PROJ = 32, SKR_P1 = 4, SKR_P2 = 12;   // columns per thread: K <= 1280 / 1024 / 128 / 384

__device__ __forceinline__ void xpub(unsigned long long* w, float v, unsigned tag) {
  __hip_atomic_store(w, ((unsigned long long)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float xwait(const unsigned long long* w, unsigned tag) {
  return __uint_as_float((unsigned)poll_tag(w, tag));
}

// thread (row r = tid % 16, K part kp = tid / 16) keeps columns kp*KR .. kp*KR+KR-1 of row `row0 + r` of a
// K-major matrix W[k][ld] (zero beyond nrows / K)
template <int KRM>
__device__ __forceinline__ void stat_load(float (&w)[KRM], const float* __restrict__ W, int ld, int row0, int nrows, int K, int tid) {
  const int r = tid % SSC, kp = tid / SSC, KR = (K + SKP - 1) / SKP;
#pragma unroll
  for (int i = 0; i < KRM; ++i) {
    const int k = kp * KR + i;
    w[i] = (i < KR && k < K && row0 + r < nrows) ? W[(size_t)k * ld + row0 + r] : 0.0f;
  }
}
// 16-row matvec from the register-resident stretch; the row sums land in part[256 .. 271] (after a barrier)
template <int KRM>
__device__ __forceinline__ void stat_mv16(const float (&w)[KRM], int K, const float* in, float* part, int tid) {
  const int r = tid % SSC, kp = tid / SSC, KR = (K + SKP - 1) / SKP;
  const int kb = kp * KR;
  float acc = 0.0f;
#pragma unroll
  for (int i = 0; i < KRM; ++i) acc = fmaf(w[i], in[kb + i], acc);   // reads past K (into the next, finite LDS array) meet zero weights
  // the 4 K parts inside a wave by shuffles, the 8 waves through LDS, both in a fixed order
  acc += __shfl_xor(acc, 16);
  acc += __shfl_xor(acc, 32);
  if ((tid & 63) < SSC) part[(tid >> 6) * SSC + r] = acc;
  __syncthreads();
  if (tid < SSC) {
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < NTC / 64; ++j) s += part[j * SSC + tid];
    part[256 + tid] = s;
  }
  __syncthreads();
}

constexpr int SNU = 4;   // max utterances sharing one set of workers

// The same for up to NU input vectors at once (in0 + u * ustride, utterances with their `alive` bit set): one
// pass over the register-resident weights, one pair of barriers; row sums land in part[512 + 16 u .. ].
template <int KRM, int NU>
__device__ __forceinline__ void stat_mv16n(const float (&w)[KRM], int K, const float* in0, int ustride, unsigned alive, float* part,
                                           int tid) {
  if constexpr (NU > 2) {   // three or more accumulator sets next to 128 weight registers spill: one vector at a time
#pragma unroll
    for (int u = 0; u < NU; ++u)
      if (alive >> u & 1) {
        stat_mv16(w, K, in0 + (size_t)u * ustride, part, tid);
        if (tid < SSC) part[512 + u * SSC + tid] = part[256 + tid];
      }
    return;
  }
  const int r = tid % SSC, kp = tid / SSC, KR = (K + SKP - 1) / SKP;
  const int kb = kp * KR;
  float acc[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) acc[u] = 0.0f;
#pragma unroll
  for (int i = 0; i < KRM; ++i)
#pragma unroll
    for (int u = 0; u < NU; ++u) acc[u] = fmaf(w[i], in0[(size_t)u * ustride + kb + i], acc[u]);   // dead slots read zeros
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    acc[u] += __shfl_xor(acc[u], 16);
    acc[u] += __shfl_xor(acc[u], 32);
    if ((tid & 63) < SSC) part[((tid >> 6) * NU + u) * SSC + r] = acc[u];
  }
  __syncthreads();
  if (tid < SSC * NU) {
    const int u = tid / SSC, rr = tid % SSC;
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < NTC / 64; ++j) s += part[(j * NU + u) * SSC + rr];
    part[512 + tid] = s;
  }
  __syncthreads();
  (void)alive;
}

template <int NU>
__global__ __launch_bounds__(NTC) void k_decoder_split(DecArgs p) {
  extern __shared__ float sm[];
  __shared__ int s_stop[SNU];
  __shared__ float c_att[SNU][SU], c_dec[SNU][SU];
  // A group = NU utterances that share NWK workers (the register-resident weights serve all of them) and
  // have one main workgroup each: blocks 0..NU-1 are the mains, the rest the workers.
  const int blk = blockIdx.x, grp = blockIdx.y, tid = threadIdx.x;
  const size_t xstride = (size_t)(p.NF + 1 + 2 * p.P + p.A + p.E + p.D + 8);
  const int KA = p.P + p.E + p.A, KD = p.A + p.E + p.D, KP = p.D + p.E;
  if (tid < SNU) s_stop[tid] = 0;
  if (blk >= NU) {   // ------------------------------------------------ worker
    const int wg = blk - NU, unit0 = wg * SU, row0 = wg * SSC;
    // per utterance: in_att [prenet | ctx | ah], in_dec [ah | ctx | dh], in_proj [dh | ctx], xin (previous mel
    // frame), p1 (prenet layer-1 output); then 512 floats of reduction scratch
    const int ustride = KA + KD + KP + round_up(p.NF, 4) + round_up(p.P, 4);
    float* part = sm + (size_t)NU * ustride;   // 1024 floats
    for (int i = tid; i < NU * ustride + 1024; i += NTC) sm[i] = 0.0f;
    if (tid < SNU * SU) { (&c_att[0][0])[tid] = 0.0f; (&c_dec[0][0])[tid] = 0.0f; }
    // w_p1 holds rows of W1.Wp (prenet layer 1 composed with the projection, k_fold_prenet): the prenet
    // starts from [dh | ctx] in the same stage as the projection, one exchange earlier
    float w_att[SKR_LSTM], w_dec[SKR_LSTM], w_proj[SKR_PROJ], w_p1[SKR_PROJ], w_p2[SKR_P2];
    stat_load(w_att, p.att_w4 + (size_t)wg * KA * SSC, SSC, 0, SSC, KA, tid);
    stat_load(w_dec, p.dec_w4 + (size_t)wg * KD * SSC, SSC, 0, SSC, KD, tid);
    // The projection + gate rows go to the workers BEHIND those that hold prenet rows (workers 19..24 with the reference's widths)
    // when there are enough of them: both matvecs start from the same [dh | ctx] and then run side by side instead of one after
    // the other in workers 0..5, the slowest link of every frame (worker 0: 2.3 us of its 11 us outside the attention).  Same rows,
    // same sums: same bits.
    const int pre_wk = (p.P + SSC - 1) / SSC, proj_wk = (p.NF + 1 + SSC - 1) / SSC;
    const int proj_w0 = pre_wk + proj_wk <= p.nwk ? pre_wk : 0;
    const int prow0 = (wg - proj_w0) * SSC;   // (negative: none)
    stat_load(w_proj, p.proj_t, round_up(p.NF + 1, 4), prow0 < 0 ? p.NF + 1 : prow0, p.NF + 1, KP, tid);
    stat_load(w_p1, p.w1p_t, round_up(p.P, 4), row0, p.P, KP, tid);
    stat_load(w_p2, p.dp1_t, round_up(p.P, 4), row0, p.P, p.P, tid);
    const bool has_proj = prow0 >= 0 && prow0 < p.NF + 1, has_pre = row0 < p.P;
    unsigned alive = 0;   // bit u: utterance u of the group is still decoding (identical in every workgroup)
    for (int u = 0; u < NU; ++u)
      if (grp * NU + u < p.B) alive |= 1u << u;
    __syncthreads();
#define FOR_ALIVE(u) _Pragma("unroll") for (int u = 0; u < NU; ++u) if (alive >> u & 1)
#define UTT(u)                                                                                          \
  const int b = grp * NU + u;                                                                           \
  unsigned long long* MEL = p.xsplit + (size_t)b * xstride;                                             \
  unsigned long long *X1 = MEL + p.NF + 1, *X2 = X1 + p.P, *AH = X2 + p.P, *CTX = AH + p.A, *DH = CTX + p.E; \
  float* in_att = sm + (size_t)u * ustride; float* in_dec = in_att + KA; float* in_proj = in_dec + KD;  \
  float* xin = in_proj + KP; float* p1 = xin + round_up(p.NF, 4);                                       \
  (void)MEL; (void)X1; (void)X2; (void)AH; (void)CTX; (void)DH; (void)in_att; (void)in_dec; (void)in_proj; (void)xin; (void)p1
    long long wtk = clock64();
#define WPROF(slot)                                                       \
  if (p.prof && wg == 0 && grp == 0 && tid == 0) {                        \
    const long long now = clock64();                                      \
    p.prof[16 + slot] += now - wtk;                                       \
    wtk = now;                                                            \
  }
    for (int t = 0;; ++t) {
      const unsigned tag = t + 1;
      if (t > 0) {
        // projection + gate rows of frame t-1 (model.py:436-441); the stopping frame is kept (:524-528)
        if (has_proj) stat_mv16n<SKR_PROJ, NU>(w_proj, KP, sm + KA + KD, ustride, alive, part, tid);
        if (has_proj) FOR_ALIVE(u) {
          UTT(u);
          if (tid < SSC && prow0 + tid <= p.NF) {
            const int row = prow0 + tid;
            const float v = part[512 + u * SSC + tid] + p.proj_b[row];
            if (row < p.NF) {
              p.mel[((size_t)b * p.NF + row) * p.max_steps + t - 1] = v;
              if (p.melx) xpub(p.melx + (size_t)(t - 1) * p.NF + row, v, tag - 1);
            } else { p.gate[(size_t)b * p.max_steps + t - 1] = v; xpub(MEL + row, v, tag); }   // only the gate crosses workgroups
          }
        }
        // prenet layer 1 of frame t from the same [dh | ctx]: relu((W1 Wp) v + W1 bp), dropout  (model.py:132-135)
        if (has_pre) stat_mv16n<SKR_PROJ, NU>(w_p1, KP, sm + KA + KD, ustride, alive, part, tid);
        if (has_pre) FOR_ALIVE(u) {
          UTT(u);
          const uint8_t* mk = p.masks + ((size_t)t * 2 * p.B + b) * p.P;
          if (tid < SSC && row0 + tid < p.P)
            xpub(X1 + row0 + tid, fmaxf(part[512 + u * SSC + tid] + p.b1p[row0 + tid], 0.0f) * (float)mk[row0 + tid] * 2.0f, tag);
        }
        WPROF(0)   // projection + prenet-1 rows
        FOR_ALIVE(u) {
          UTT(u);
          if (tid == 0) s_stop[u] = sigm(xwait(MEL + p.NF, tag)) > p.gate_thr || t == dec_step_limit(p, b);
        }
        __syncthreads();
        for (int u = 0; u < NU; ++u)
          if (s_stop[u]) alive &= ~(1u << u);
        if (!alive) break;
        FOR_ALIVE(u) {
          UTT(u);
          for (int i = tid; i < p.P; i += NTC) p1[i] = xwait(X1 + i, tag);
        }
        __syncthreads();
        WPROF(1)   // gather gate + X1
      }
      // (frame 0: the go frame is zero and the prenet has no bias, so p1 = 0 as initialised)
      if (has_pre) stat_mv16n<SKR_P2, NU>(w_p2, p.P, sm + KA + KD + KP + round_up(p.NF, 4), ustride, alive, part, tid);
      if (has_pre) FOR_ALIVE(u) {
        UTT(u);
        const uint8_t* mk = p.masks + ((size_t)t * 2 * p.B + b) * p.P;
        if (tid < SSC && row0 + tid < p.P)
          xpub(X2 + row0 + tid, fmaxf(part[512 + u * SSC + tid], 0.0f) * (float)mk[(size_t)p.B * p.P + row0 + tid] * 2.0f, tag);
      }
      WPROF(2)   // prenet-2 rows
      FOR_ALIVE(u) {
        UTT(u);
        for (int i = tid; i < p.P; i += NTC) in_att[i] = xwait(X2 + i, tag);
      }
      __syncthreads();
      WPROF(3)   // gather X2
      // attention LSTMCell slice on [prenet | ctx | ah]  (model.py:400-403)
      stat_mv16n<SKR_LSTM, NU>(w_att, KA, sm, ustride, alive, part, tid);
      FOR_ALIVE(u) {
        UTT(u);
        if (tid < SU && unit0 + tid < p.A) {
          const float* gs = part + 512 + u * SSC;
          const int un = unit0 + tid;
          xpub(AH + un, lstm_point(gs[tid] + p.att_b[un], gs[SU + tid] + p.att_b[p.A + un], gs[2 * SU + tid] + p.att_b[2 * p.A + un],
                                   gs[3 * SU + tid] + p.att_b[3 * p.A + un], &c_att[u][tid]), tag);
        }
      }
      WPROF(4)   // attention LSTM slice
      FOR_ALIVE(u) {
        UTT(u);
        for (int i = tid; i < p.A; i += NTC) { const float h = xwait(AH + i, tag); in_att[p.P + p.E + i] = h; in_dec[i] = h; }
      }
      WPROF(5)   // gather AH
      FOR_ALIVE(u) {
        UTT(u);
        for (int i = tid; i < p.E; i += NTC) {
          const float c = xwait(CTX + i, tag);
          in_att[p.P + i] = c; in_dec[p.A + i] = c; in_proj[p.D + i] = c;
        }
      }
      __syncthreads();
      WPROF(6)   // gather CTX (= the main workgroup's attention)
      // decoder LSTMCell slice on [ah | ctx | dh]  (model.py:425-428)
      stat_mv16n<SKR_LSTM, NU>(w_dec, KD, sm + KA, ustride, alive, part, tid);
      FOR_ALIVE(u) {
        UTT(u);
        if (tid < SU && unit0 + tid < p.D) {
          const float* gs = part + 512 + u * SSC;
          const int un = unit0 + tid;
          xpub(DH + un, lstm_point(gs[tid] + p.dec_b[un], gs[SU + tid] + p.dec_b[p.D + un], gs[2 * SU + tid] + p.dec_b[2 * p.D + un],
                                   gs[3 * SU + tid] + p.dec_b[3 * p.D + un], &c_dec[u][tid]), tag);
        }
      }
      FOR_ALIVE(u) {
        UTT(u);
        for (int i = tid; i < p.D; i += NTC) { const float h = xwait(DH + i, tag); in_dec[p.A + p.E + i] = h; in_proj[i] = h; }
      }
      WPROF(7)   // decoder LSTM slice + gather DH
      __syncthreads();
      WPROF(8)
    }
#undef WPROF
#undef FOR_ALIVE
#undef UTT
    return;
  }
  // ------------------------------------------------------------------ main of utterance b
  const int b = grp * NU + blk;
  if (b >= p.B) return;
  unsigned long long* MEL = p.xsplit + (size_t)b * xstride;
  unsigned long long *AH = MEL + p.NF + 1 + 2 * p.P, *CTX = AH + p.A;
  const int len = p.lengths ? p.lengths[b] : p.Tin;
  DecLds L;
  dec_carve(p, sm, L);
  dec_init<NTC>(p, L, sm, tid);
  __syncthreads();
  const float* mem = p.memory + (size_t)b * p.Tin * p.E;
  const float* pm = p.pm + (size_t)b * p.Tin * p.AD;
  float* ah = L.in_att + p.P + p.E;
  // attn_energy_pre's results, behind everything dec_carve laid out
  float4* stash = reinterpret_cast<float4*>(sm + round_up((int)dec_lds_floats(p.P, p.E, p.A, p.D, p.NF, p.AD, p.NFIL, p.KSZ, p.Tin), 4));
  // the query layer's weights of this thread's (4-row slot, k range) are REQUESTED every frame before the main workgroup starts
  // waiting for the workers (gate, then the attention LSTM's hidden state: ~10 us): the 180 KB stream's L2 latency, which was
  // 2.9 us of the 12.7 us attention, hides in that wait.  (Keeping them in registers for the whole utterance spills: 96 registers
  // next to the context's 48 and the kernel's 224 for the workers' LSTM slices.)
  constexpr int QR = 24;
  const int q_ks = pick_ks<NTC>(L.ADp, p.A);
  const int q_ns = L.ADp >> 2, q_slot = tid % q_ns, q_part = tid / q_ns;
  const int q_k0 = (int)((long)q_part * p.A / q_ks), q_k1 = (int)((long)(q_part + 1) * p.A / q_ks);
  CtxRows crows;
#pragma unroll
  for (int i = 0; i < CTX_SLOTS; ++i)
#pragma unroll
    for (int r = 0; r < CTX_CR; ++r) crows.v[i][r] = 0.0f;
  const bool rows_on = ctx_rows_usable(p) && !(p.dbg_flags & 1);   // (dbg_flags bit 0: FACPPG_DECODER_NO_ROWS, A/B runs)
  int hi_prev = -1;
  long long tk = clock64();
#define PROF(slot)                                                        \
  if (p.prof && b == 0 && tid == 0) {                                     \
    const long long now = clock64();                                      \
    p.prof[slot] += now - tk;                                             \
    tk = now;                                                             \
  }
  for (int t = 0;; ++t) {
    const unsigned tag = t + 1;
    int lo, hi;
    attn_window_range(p.window, t, len, &lo, &hi);
    attn_features<NTC>(p, L, lo, min(64, hi - lo + 1), tid);   // needs only frame t-1's weights
    __syncthreads();
    PROF(6)
    attn_energy_pre<NTC>(p, L, pm, lo, min(64, hi - lo + 1), tid, stash);   // ... and so does this part of the energies
    PROF(7)
    if (rows_on) { ctx_rows_update(p, mem, hi_prev, hi, tid, crows); hi_prev = hi; }   // ... and the window's entering memory row
    PROF(14)
    asm volatile("" ::: "memory");   // (the 96 registers of query weights requested next must not be hoisted over it: spills)
    float4 wq[QR];
#pragma unroll
    for (int i = 0; i < QR; ++i)
      wq[i] = (q_part < q_ks && q_k0 + i < q_k1) ? reinterpret_cast<const float4*>(p.q_t)[(size_t)(q_k0 + i) * q_ns + q_slot]
                                                           : make_float4(0.f, 0.f, 0.f, 0.f);
    PROF(2)
    if (t > 0) {
      if (tid == 0) s_stop[0] = sigm(xwait(MEL + p.NF, tag)) > p.gate_thr || t == dec_step_limit(p, b);
      __syncthreads();
      if (s_stop[0]) {
        if (tid == 0) {
          __hip_atomic_store(p.out_len + b, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (a frame collector on another stream watches it)
        }
        break;
      }
    }
    PROF(0)
    for (int i = tid; i < p.A; i += NTC) ah[i] = xwait(AH + i, tag);
    __syncthreads();
    PROF(3)
    dec_attention<NTC, QR, true, true>(p, L, mem, pm, len, t, b, tid, true, true, wq, stash, crows, rows_on, CTX, tag);   // (publishes CTX)
    PROF(5)
  }
#undef PROF
}

// ------------------------------------------------------------------------------------------
struct TWs {
  size_t a0, a1, xproj, mem_cm, mask, xchg, splitk, splitk_bytes, total;
};
TWs tws_layout(const facppg_taco_config& c, int B, int Tin) {
  TWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t E = c.encoder_embedding_dim > c.symbols_embedding_dim ? c.encoder_embedding_dim : c.symbols_embedding_dim;
  w.a0 = take((size_t)B * E * Tin * 4);
  w.a1 = take((size_t)B * E * Tin * 4);
  w.xproj = take((size_t)B * Tin * 4 * c.encoder_embedding_dim * 4);   // [B][Tin][2*4H], 4H = 2E
  w.mem_cm = take((size_t)B * c.encoder_embedding_dim * Tin * 4);
  w.mask = take((size_t)2 * B * c.symbols_embedding_dim * Tin);
  w.xchg = take((size_t)B * 2 * 2 * (c.encoder_embedding_dim / 2) * 8);   // k_bilstm_coop {value, tag} words
  // split-K partial sums for the small-N encoder GEMMs (only short batches split; see gemm_launch)
  const size_t mx = c.encoder_embedding_dim > c.symbols_embedding_dim ? c.encoder_embedding_dim : c.symbols_embedding_dim;
  w.splitk_bytes = (size_t)16 * B * mx * Tin * 4;   // <= 16 splits of the widest split product (rows x Tin per utterance)
  w.splitk = take(w.splitk_bytes);
  w.total = off;
  return w;
}

}  // namespace

static int taco_check(const facppg_taco_config* c) {
  FACPPG_REQUIRE(c, FACPPG_EINVAL, "config is NULL");
  FACPPG_REQUIRE(c->n_symbols > 0 && c->symbols_embedding_dim > 0 && c->encoder_embedding_dim > 0 && c->encoder_embedding_dim % 2 == 0,
                 FACPPG_EINVAL, "bad encoder dims");
  FACPPG_REQUIRE(c->symbols_embedding_dim == c->encoder_embedding_dim, FACPPG_EUNSUPPORTED,
                 "symbols_embedding_dim must equal encoder_embedding_dim (prenet feeds the conv bank)");
  FACPPG_REQUIRE(c->encoder_n_convolutions >= 0 && c->encoder_n_convolutions <= 8 && c->postnet_n_convolutions >= 2 &&
                     c->postnet_n_convolutions <= 8 && c->encoder_kernel_size % 2 == 1 && c->postnet_kernel_size % 2 == 1,
                 FACPPG_EUNSUPPORTED, "conv stack sizes out of range");
  FACPPG_REQUIRE(c->attention_rnn_dim == c->decoder_rnn_dim && c->attention_rnn_dim % 4 == 0 && 4 * c->attention_rnn_dim <= 4096,
                 FACPPG_EUNSUPPORTED, "attention_rnn_dim must equal decoder_rnn_dim, be a multiple of 4 and <= 1024");
  FACPPG_REQUIRE(c->attention_location_n_filters >= 1 && c->attention_location_n_filters <= 32 && c->attention_dim >= 1 &&
                     c->attention_dim <= 256,
                 FACPPG_EUNSUPPORTED, "attention_location_n_filters must be <= 32 and attention_dim <= 256 (one MFMA pass each)");
  FACPPG_REQUIRE(c->encoder_embedding_dim <= 768 && c->prenet_dim <= NTC && c->attention_dim <= NTC && c->n_acoustic_feat_dims < NTC &&
                     (2 * c->encoder_embedding_dim) % 4 == 0,
                 FACPPG_EUNSUPPORTED, "decoder dims exceed the 512-thread cooperative workgroup");
  FACPPG_REQUIRE(c->attention_location_kernel_size % 2 == 1 && c->attention_location_n_filters > 0, FACPPG_EUNSUPPORTED,
                 "attention_location_kernel_size must be odd");
  return FACPPG_OK;
}

static size_t taco_count(const facppg_taco_config* c) {
  const size_t S = c->symbols_embedding_dim, E = c->encoder_embedding_dim, H = E / 2, K = c->encoder_kernel_size;
  const size_t P = c->prenet_dim, A = c->attention_rnn_dim, D = c->decoder_rnn_dim, AD = c->attention_dim, NF = c->n_acoustic_feat_dims;
  size_t n = S * c->n_symbols + S * S;
  n += (size_t)c->encoder_n_convolutions * (E * E * K + 5 * E);
  n += 2 * (4 * H * E + 4 * H * H + 8 * H);
  n += P * NF + P * P;
  n += 4 * A * (P + E) + 4 * A * A + 8 * A;
  n += AD * A + AD * E + AD + (size_t)c->attention_location_n_filters * 2 * c->attention_location_kernel_size +
       AD * c->attention_location_n_filters;
  n += 4 * D * (A + E) + 4 * D * D + 8 * D;
  n += NF * (D + E) + NF + (D + E) + 1;
  const size_t PE = c->postnet_embedding_dim, PK = c->postnet_kernel_size;
  for (int j = 0; j < c->postnet_n_convolutions; ++j) {
    const size_t ci = j == 0 ? NF : PE, co = j == c->postnet_n_convolutions - 1 ? NF : PE;
    n += co * ci * PK + 5 * co;
  }
  return n;
}

extern "C" size_t facppg_taco_weight_count(const facppg_taco_config* c) {
  if (taco_check(c) != FACPPG_OK) return 0;
  return taco_count(c);
}

extern "C" int facppg_taco_create(const facppg_taco_config* cfg, const float* wsrc, size_t n_floats, int device, void* stream_,
                                  facppg_taco** out) {
  if (int rc = taco_check(cfg)) return rc;
  FACPPG_REQUIRE(wsrc && out, FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(n_floats == taco_count(cfg), FACPPG_EINVAL, "weight blob has %zu floats, expected %zu", n_floats, taco_count(cfg));
  hipStream_t s = (hipStream_t)stream_;
  FACPPG_HIP_CHECK(hipSetDevice(device));
  int n_cu = 0;
  FACPPG_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device));
  facppg_taco* h = new (std::nothrow) facppg_taco();
  FACPPG_REQUIRE(h, FACPPG_EINVAL, "out of host memory");
  struct Guard {   // every early return below frees the handle (and its arena, once allocated)
    facppg_taco* h;
    ~Guard() { if (h) { if (h->arena) (void)hipFree(h->arena); delete h; } }
  } guard{h};
  h->c = *cfg; h->device = device;
  {
    // How many workgroups the cooperative kernels may keep co-resident: what the occupancy calculator says fits
    // per CU for each of them at its largest dynamic LDS, but never more than ONE per CU (each workgroup is sized
    // for a whole CU's LDS / L1 bandwidth), less 1/16 of the chip kept free so that a second handle's launch or
    // the neighbouring stages' kernels on other streams find a CU.  0 => the one-workgroup kernels run instead.
    int per_cu = 1;
    const struct { const void* fn; size_t lds; } coop_kernels[] = {
        {(const void*)k_decoder_coop, 150 * 1024}, {(const void*)k_decoder_split<1>, 150 * 1024},
        {(const void*)k_decoder_split<2>, 150 * 1024}, {(const void*)k_decoder_split<3>, 150 * 1024},
        {(const void*)k_bilstm_coop<32, 96>, 0}, {(const void*)k_bilstm_coop<64, 152>, 0}};
    for (const auto& k : coop_kernels) {
      int nb = 0;
      if (k.lds) FACPPG_HIP_CHECK(hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k.lds));
      FACPPG_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k.fn, NTC, k.lds));
      per_cu = std::min(per_cu, nb);
    }
    const int resident = n_cu * per_cu;
    h->coop_limit = resident - resident / 16;
    // hipMemcpyToSymbol writes the CURRENT device's copy of the constant only, so the bound is uploaded once PER DEVICE (a
    // process-wide "done" flag left every later GPU of a multi-GPU process at 0 = unbounded); the set is mutex-guarded
    // because handles may be created from several host threads
    static std::mutex poll_mu;
    static unsigned long long poll_devices = 0;   // bit d: device d's copy is set (devices >= 64: uploaded at every create)
    {
      std::lock_guard<std::mutex> lock(poll_mu);
      const bool known = device < 64 && ((poll_devices >> device) & 1ull);
      if (!known) {
        const char* pl = getenv("FACPPG_POLL_LIMIT");
        const double seconds = pl ? strtod(pl, nullptr) : 20.0;
        int khz = 0;
        FACPPG_HIP_CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device));
        const unsigned long long ticks = seconds > 0 && khz > 0 ? (unsigned long long)(seconds * 1e3 * khz) : 0ull;
        FACPPG_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_poll_limit_ticks), &ticks, sizeof(ticks)));
        if (device < 64) poll_devices |= 1ull << device;
      }
    }
  }
  (void)hipDeviceGetAttribute(&h->wall_khz, hipDeviceAttributeWallClockRate, device);
  const facppg_taco_config& c = *cfg;
  const int S = c.symbols_embedding_dim, E = c.encoder_embedding_dim, H = E / 2, K = c.encoder_kernel_size;
  const int P = c.prenet_dim, A = c.attention_rnn_dim, D = c.decoder_rnn_dim, AD = c.attention_dim, NF = c.n_acoustic_feat_dims;
  const int NFIL = c.attention_location_n_filters, KSZ = c.attention_location_kernel_size;
  const int PE = c.postnet_embedding_dim, PK = c.postnet_kernel_size;
  const int Pp = round_up(P, 4), ADp = round_up(AD, 4), NFp = round_up(NF + 1, 4), G = 4 * A;

  // pass 1: sizes
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  struct { size_t pre0, pre1, conv[8], conv_b[8], conv_sc[8], conv_sh[8], wih, whh[2], lstm_b, mem_w, dp0, dp1, att, att_b, dec, dec_b, q,
           proj, proj_b, lc, ld, v, attc[5], decc[5], att4, dec4, w1p, b1p, post[8], post_b[8], post_sc[8], post_sh[8]; } o;
  o.pre0 = take(packed_a_float4s(S, c.n_symbols) * 16);
  o.pre1 = take(packed_a_float4s(S, S) * 16);
  for (int j = 0; j < c.encoder_n_convolutions; ++j) {
    o.conv[j] = take(packed_a_float4s(E, E * K) * 16);
    o.conv_b[j] = take(E * 4); o.conv_sc[j] = take(E * 4); o.conv_sh[j] = take(E * 4);
  }
  o.wih = take(packed_a_float4s(8 * H, E) * 16);
  o.whh[0] = take((size_t)H * 4 * H * 4); o.whh[1] = take((size_t)H * 4 * H * 4);
  o.lstm_b = take((size_t)8 * H * 4);
  o.mem_w = take(packed_a_float4s(AD, E) * 16);
  o.dp0 = take((size_t)NF * Pp * 4); o.dp1 = take((size_t)P * Pp * 4);
  o.att = take((size_t)(P + E + A) * G * 4); o.att_b = take((size_t)G * 4);
  o.dec = take((size_t)(A + E + D) * G * 4); o.dec_b = take((size_t)G * 4);
  o.q = take((size_t)A * ADp * 4);
  o.proj = take((size_t)(D + E) * NFp * 4); o.proj_b = take((size_t)NFp * 4);
  const int CUs[5] = {8, 20, 40, 75, 150};
  for (int v = 0; v < 5; ++v) {
    const int nwg = (A + CUs[v] - 1) / CUs[v];
    o.attc[v] = take((size_t)nwg * (P + E + A) * 4 * CUs[v] * 4); o.decc[v] = take((size_t)nwg * (A + E + D) * 4 * CUs[v] * 4);
  }
  const int nwk = (A + 3) / 4;   // k_decoder_split workers: 4 units each
  o.att4 = take((size_t)nwk * (P + E + A) * 16 * 4); o.dec4 = take((size_t)nwk * (A + E + D) * 16 * 4);
  o.w1p = take((size_t)(D + E) * Pp * 4); o.b1p = take((size_t)Pp * 4);
  o.lc = take((size_t)NFIL * 2 * KSZ * 4); o.ld = take((size_t)AD * NFIL * 4); o.v = take((size_t)AD * 4);
  for (int j = 0; j < c.postnet_n_convolutions; ++j) {
    const int ci = j == 0 ? NF : PE, co = j == c.postnet_n_convolutions - 1 ? NF : PE;
    o.post[j] = take(packed_a_float4s(co, ci * PK) * 16);
    o.post_b[j] = take(co * 4); o.post_sc[j] = take(co * 4); o.post_sh[j] = take(co * 4);
  }
  if (hipMalloc((void**)&h->arena, off) != hipSuccess) {
    h->arena = nullptr;
    set_error("hipMalloc(%zu) failed", off);
    return FACPPG_EHIP;
  }
  int rc = FACPPG_OK;
  auto F = [&](size_t x) { return (float*)(h->arena + x); };
  auto F4 = [&](size_t x) { return (float4*)(h->arena + x); };
  auto hipok = [&](hipError_t e) { if (e != hipSuccess && !rc) { set_error("HIP error in facppg_taco_create: %s", hipGetErrorString(e)); rc = FACPPG_EHIP; } };
  auto cpy = [&](float* dst, const float* src, size_t n) { hipok(hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToDevice, s)); };
  auto tr = [&](const float* src, float* dst, int R, int Kk, int Rp, int k_off) {
    const int n = Kk * Rp;
    k_transpose_pad<<<(n + 255) / 256, 256, 0, s>>>(src, dst, R, Kk, Rp, k_off);
  };
  auto bn = [&](const float*& src, int n, float* sc, float* sh) {
    k_bn_fold<<<(n + 255) / 256, 256, 0, s>>>(src, src + n, src + 2 * n, src + 3 * n, c.bn_eps, sc, sh, n);
    src += 4 * (size_t)n;
  };
  hipok(hipMemsetAsync(h->arena, 0, off, s));
  const float* src = wsrc;
  h->pre0 = F4(o.pre0); h->pre1 = F4(o.pre1);
  if (!rc) rc = pack_a(src, S, c.n_symbols, 1, h->pre0, s); src += (size_t)S * c.n_symbols;
  if (!rc) rc = pack_a(src, S, S, 1, h->pre1, s); src += (size_t)S * S;
  for (int j = 0; j < c.encoder_n_convolutions; ++j) {
    h->conv[j] = F4(o.conv[j]); h->conv_b[j] = F(o.conv_b[j]); h->conv_scale[j] = F(o.conv_sc[j]); h->conv_shift[j] = F(o.conv_sh[j]);
    if (!rc) rc = pack_a(src, E, E, K, h->conv[j], s); src += (size_t)E * E * K;
    cpy(h->conv_b[j], src, E); src += E;
    bn(src, E, h->conv_scale[j], h->conv_shift[j]);
  }
  // LSTM: both directions' W_ih stacked into one [8H][E] GEMM operand
  h->wih = F4(o.wih); h->whh_t[0] = F(o.whh[0]); h->whh_t[1] = F(o.whh[1]); h->lstm_b = F(o.lstm_b);
  {
    float* tmp = nullptr;
    hipok(hipMalloc((void**)&tmp, (size_t)8 * H * E * 4));
    for (int d = 0; d < 2 && !rc; ++d) {
      const float* wih = src; src += (size_t)4 * H * E;
      const float* whh = src; src += (size_t)4 * H * H;
      const float* bih = src; src += 4 * H;
      const float* bhh = src; src += 4 * H;
      cpy(tmp + (size_t)d * 4 * H * E, wih, (size_t)4 * H * E);
      tr(whh, h->whh_t[d], 4 * H, H, 4 * H, 0);
      k_add2<<<(4 * H + 255) / 256, 256, 0, s>>>(bih, bhh, h->lstm_b + d * 4 * H, 4 * H);
    }
    if (!rc) rc = pack_a(tmp, 8 * H, E, 1, h->wih, s);
    hipok(hipStreamSynchronize(s));
    (void)hipFree(tmp);
  }
  // decoder
  h->dp0_t = F(o.dp0); h->dp1_t = F(o.dp1); h->att_t = F(o.att); h->att_b = F(o.att_b); h->dec_t = F(o.dec); h->dec_b = F(o.dec_b);
  h->q_t = F(o.q); h->proj_t = F(o.proj); h->proj_b = F(o.proj_b); h->loc_conv = F(o.lc); h->loc_dense = F(o.ld); h->v = F(o.v);
  h->mem_w = F4(o.mem_w);
  tr(src, h->dp0_t, P, NF, Pp, 0); src += (size_t)P * NF;
  tr(src, h->dp1_t, P, P, Pp, 0); src += (size_t)P * P;
  tr(src, h->att_t, G, P + E, G, 0); src += (size_t)G * (P + E);
  tr(src, h->att_t, G, A, G, P + E); src += (size_t)G * A;
  k_add2<<<(G + 255) / 256, 256, 0, s>>>(src, src + G, h->att_b, G); src += 2 * (size_t)G;
  tr(src, h->q_t, AD, A, ADp, 0); src += (size_t)AD * A;
  if (!rc) rc = pack_a(src, AD, E, 1, h->mem_w, s); src += (size_t)AD * E;
  cpy(h->v, src, AD); src += AD;
  cpy(h->loc_conv, src, (size_t)NFIL * 2 * KSZ); src += (size_t)NFIL * 2 * KSZ;
  cpy(h->loc_dense, src, (size_t)AD * NFIL); src += (size_t)AD * NFIL;
  tr(src, h->dec_t, G, A + E, G, 0); src += (size_t)G * (A + E);
  tr(src, h->dec_t, G, D, G, A + E); src += (size_t)G * D;
  k_add2<<<(G + 255) / 256, 256, 0, s>>>(src, src + G, h->dec_b, G); src += 2 * (size_t)G;
  for (int v = 0; v < 5; ++v) {
    const int U = CUs[v], nwg = (A + U - 1) / U;
    h->att_coop[v] = F(o.attc[v]); h->dec_coop[v] = F(o.decc[v]); h->coop_U[v] = U; h->coop_nwg[v] = nwg;
    const size_t na = (size_t)nwg * (P + E + A) * 4 * U, nd = (size_t)nwg * (A + E + D) * 4 * U;
    k_pack_coop<<<(unsigned)((na + 255) / 256), 256, 0, s>>>(h->att_t, h->att_coop[v], P + E + A, A, U, nwg);
    k_pack_coop<<<(unsigned)((nd + 255) / 256), 256, 0, s>>>(h->dec_t, h->dec_coop[v], A + E + D, D, U, nwg);
  }
  {
    h->att_w4 = F(o.att4); h->dec_w4 = F(o.dec4); h->split_nwk = nwk;
    const size_t na = (size_t)nwk * (P + E + A) * 16, nd = (size_t)nwk * (A + E + D) * 16;
    k_pack_coop<<<(unsigned)((na + 255) / 256), 256, 0, s>>>(h->att_t, h->att_w4, P + E + A, A, 4, nwk);
    k_pack_coop<<<(unsigned)((nd + 255) / 256), 256, 0, s>>>(h->dec_t, h->dec_w4, A + E + D, D, 4, nwk);
  }
  // projection rows 0..NF-1, gate row NF, K-major [D+E][NFp]
  {
    const float* pw = src; src += (size_t)NF * (D + E);
    const float* pb = src; src += NF;
    const float* gw = src; src += (D + E);
    const float* gb = src; src += 1;
    tr(pw, h->proj_t, NF, D + E, NFp, 0);
    // gate weights into column NF: a [1][D+E] matrix transposed with row offset -> strided copy
    hipok(hipMemcpy2DAsync(h->proj_t + NF, (size_t)NFp * 4, gw, 4, 4, D + E, hipMemcpyDeviceToDevice, s));
    cpy(h->proj_b, pb, NF);
    cpy(h->proj_b + NF, gb, 1);
    h->w1p_t = F(o.w1p); h->b1p = F(o.b1p);
    const int n = (D + E + 1) * Pp;
    k_fold_prenet<<<(n + 255) / 256, 256, 0, s>>>(h->proj_t, h->proj_b, h->dp0_t, h->w1p_t, h->b1p, D + E, NF, NFp, Pp);
  }
  for (int j = 0; j < c.postnet_n_convolutions; ++j) {
    const int ci = j == 0 ? NF : PE, co = j == c.postnet_n_convolutions - 1 ? NF : PE;
    h->post[j] = F4(o.post[j]); h->post_b[j] = F(o.post_b[j]); h->post_scale[j] = F(o.post_sc[j]); h->post_shift[j] = F(o.post_sh[j]);
    if (!rc) rc = pack_a(src, co, ci, PK, h->post[j], s); src += (size_t)co * ci * PK;
    cpy(h->post_b[j], src, co); src += co;
    bn(src, co, h->post_scale[j], h->post_shift[j]);
  }
  hipok(hipGetLastError());
  hipok(hipStreamSynchronize(s));
  if (!rc && (size_t)(src - wsrc) != n_floats) { set_error("internal: consumed %zu of %zu weights", (size_t)(src - wsrc), n_floats); rc = FACPPG_EINVAL; }
  if (rc) return rc;
  guard.h = nullptr;
  *out = h;
  return FACPPG_OK;
}

extern "C" void facppg_taco_destroy(facppg_taco* h) {
  if (!h) return;
  (void)hipFree(h->arena);
  delete h;
}

extern "C" size_t facppg_taco_workspace_bytes(const facppg_taco* h, int B, int Tin) {
  if (!h || B <= 0 || Tin <= 0) return 0;
  return tws_layout(h->c, B, Tin).total;
}

// Encoder.inference (model.py:237-249) + Attention.memory_layer (model.py:334).
extern "C" int facppg_taco_encode(facppg_taco* h, const float* ppg_dev, const int32_t* lengths_dev, const uint8_t* masks_dev,
                                  uint64_t seed, int B, int Tin, float* memory_dev, float* pm_dev, void* ws_, size_t ws_bytes,
                                  void* stream_) {
  FACPPG_REQUIRE(h && ppg_dev && memory_dev && pm_dev && ws_, FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(B > 0 && Tin > 0 && B <= 65535, FACPPG_EINVAL, "bad B/Tin");
  const facppg_taco_config& c = h->c;
  const TWs w = tws_layout(c, B, Tin);
  FACPPG_REQUIRE(ws_bytes >= w.total, FACPPG_EWORKSPACE, "workspace has %zu bytes, need %zu", ws_bytes, w.total);
  hipStream_t s = (hipStream_t)stream_;
  char* ws = (char*)ws_;
  float* a0 = (float*)(ws + w.a0);
  float* a1 = (float*)(ws + w.a1);
  float* xproj = (float*)(ws + w.xproj);
  float* mem_cm = (float*)(ws + w.mem_cm);
  const int S = c.symbols_embedding_dim, E = c.encoder_embedding_dim, H = E / 2;
  const uint8_t* masks = masks_dev;
  if (!masks) {
    uint8_t* m = (uint8_t*)(ws + w.mask);
    const size_t n = (size_t)2 * B * S * Tin;
    k_random_mask<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(m, n, seed);
    masks = m;
  }
  GemmArgs g;
  g.B = B; g.N = Tin; g.n_valid = lengths_dev;
  g.splitk_ws = (float*)(ws + w.splitk); g.splitk_ws_bytes = w.splitk_bytes;
  // prenet: 2 x (Linear no bias, ReLU, dropout p=.5 always on)  model.py:124-135
  g.A = h->pre0; g.M = S; g.Cin = c.n_symbols; g.X = ppg_dev; g.x_bs = (long)c.n_symbols * Tin; g.ldx = Tin; g.act = ACT_RELU;
  g.mask = masks; g.mask_bs = (long)S * Tin; g.ldmask = Tin; g.C = a0; g.c_bs = (long)S * Tin; g.ldc = Tin;
  if (int rc = gemm_launch(g, s)) return rc;
  g.A = h->pre1; g.Cin = S; g.X = a0; g.x_bs = (long)S * Tin; g.mask = masks + (size_t)B * S * Tin; g.C = a1;
  if (int rc = gemm_launch(g, s)) return rc;
  // conv bank: conv k + BN(eval) + ReLU  model.py:241-242
  float* cur = a1;
  float* nxt = a0;
  g.mask = nullptr;
  for (int j = 0; j < c.encoder_n_convolutions; ++j) {
    g.A = h->conv[j]; g.M = E; g.Cin = E; g.taps = c.encoder_kernel_size; g.pad = (c.encoder_kernel_size - 1) / 2; g.X = cur;
    g.x_bs = (long)E * Tin; g.bias = h->conv_b[j]; g.scale = h->conv_scale[j]; g.shift = h->conv_shift[j]; g.C = nxt;
    g.c_bs = (long)E * Tin;
    if (int rc = gemm_launch(g, s)) return rc;
    float* t = cur; cur = nxt; nxt = t;
  }
  // LSTM input projections for both directions, time-major out [B][Tin][8H]
  GemmArgs p;
  p.splitk_ws = g.splitk_ws; p.splitk_ws_bytes = g.splitk_ws_bytes;
  p.B = B; p.N = Tin; p.n_valid = lengths_dev; p.A = h->wih; p.M = 8 * H; p.Cin = E; p.X = cur; p.x_bs = (long)E * Tin; p.ldx = Tin;
  p.bias = h->lstm_b; p.C = xproj; p.c_bs = (long)Tin * 8 * H; p.ldc = 8 * H; p.c_transposed = 1;
  if (int rc = gemm_launch(p, s)) return rc;
  // latency shapes: W_hh register-resident, sliced over co-resident workgroups per (utterance, direction):
  // 32 units per workgroup (4 K parts of <= 96 columns) while they fit, else 64 units (2 K parts of <= 152)
  const char* bilstm_mode = getenv("FACPPG_BILSTM_MODE");   // "single" forces the one-workgroup kernel, "wide" the 64-unit slices
  const bool no_coop = bilstm_mode && !strcmp(bilstm_mode, "single");
  const bool fit32 = (long)B * 2 * ((H + 31) / 32) <= h->coop_limit && (H + 3) / 4 <= 96 && !(bilstm_mode && !strcmp(bilstm_mode, "wide"));
  const bool fit64 = (long)B * 2 * ((H + 63) / 64) <= h->coop_limit && (H + 1) / 2 <= 152;
  if (!no_coop && (fit32 || fit64)) {
    unsigned long long* xchg = (unsigned long long*)(ws + w.xchg);
    FACPPG_HIP_CHECK(hipMemsetAsync(xchg, 0, (size_t)B * 2 * 2 * H * 8, s));
    const float *w0 = h->whh_t[0], *w1 = h->whh_t[1];
    const float* xp = xproj;
    int Tin_ = Tin, H_ = H;
    void* args[] = {(void*)&xp, (void*)&w0, (void*)&w1, (void*)&lengths_dev, (void*)&Tin_, (void*)&H_, (void*)&xchg,
                    (void*)&memory_dev, (void*)&mem_cm};
    if (fit32) FACPPG_HIP_CHECK(launch_coop((const void*)k_bilstm_coop<32, 96>, dim3((H + 31) / 32, 2, B), dim3(NTC), args, 0, s));
    else FACPPG_HIP_CHECK(launch_coop((const void*)k_bilstm_coop<64, 152>, dim3((H + 63) / 64, 2, B), dim3(NTC), args, 0, s));
  } else {
    const int KS = NT / H < H ? NT / H : H;
    const size_t smem = (size_t)(2 * H + (KS > 0 ? KS : 1) * 4 * H) * 4;
    k_bilstm<<<dim3(2, B), NT, smem, s>>>(xproj, h->whh_t[0], h->whh_t[1], lengths_dev, Tin, H, memory_dev, mem_cm);
  }
  // processed_memory = memory_layer(memory), time-major [B][Tin][AD]
  GemmArgs m;
  m.splitk_ws = g.splitk_ws; m.splitk_ws_bytes = g.splitk_ws_bytes;
  m.B = B; m.N = Tin; m.n_valid = lengths_dev; m.A = h->mem_w; m.M = c.attention_dim; m.Cin = E; m.X = mem_cm; m.x_bs = (long)E * Tin;
  m.ldx = Tin; m.C = pm_dev; m.c_bs = (long)Tin * c.attention_dim; m.ldc = Tin;   // [B][AD][Tin]: positions contiguous for the energy pass
  if (int rc = gemm_launch(m, s)) return rc;
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

namespace {
struct DecWs { size_t mask, xchg, prof, xsplit, total; };
DecWs dec_ws(const facppg_taco_config& c, int B, int max_steps) {
  DecWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  w.mask = take((size_t)max_steps * 2 * B * c.prenet_dim);
  w.xchg = take((size_t)B * 2 * c.attention_rnn_dim * 8);
  w.prof = take(32 * 8);
  w.xsplit = take((size_t)B * (c.n_acoustic_feat_dims + 1 + 2 * c.prenet_dim + c.attention_rnn_dim + c.encoder_embedding_dim + c.decoder_rnn_dim + 8) * 8);
  w.total = off;
  return w;
}
}  // namespace

extern "C" int facppg_taco_draw_dropout(const facppg_taco* h, const uint64_t* seeds_dev, int B, int Tin, int max_steps,
                                        uint8_t* enc_masks_dev, uint8_t* dec_masks_dev, void* stream_) {
  FACPPG_REQUIRE(h && seeds_dev, FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(B > 0 && Tin > 0 && max_steps > 0, FACPPG_EINVAL, "bad B/Tin/max_steps");
  hipStream_t s = (hipStream_t)stream_;
  const int S = h->c.symbols_embedding_dim, P = h->c.prenet_dim;
  if (enc_masks_dev) {
    const size_t n = (size_t)2 * B * S * Tin;
    k_random_mask_enc_utt<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(enc_masks_dev, seeds_dev, B, S, Tin);
  }
  if (dec_masks_dev) {
    const size_t n = (size_t)max_steps * 2 * B * P;
    k_random_mask_dec_utt<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(dec_masks_dev, seeds_dev, B, P, max_steps);
  }
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

namespace {
// get_mask_from_lengths_window_and_time_step (src/common/utils.py:46-78) as a byte mask, 1 = masked:
// the decoder kernels never materialise it (they evaluate the attention on [lo, hi] only); this kernel
// writes the same range out through the same attn_window_range() so it can be compared bit for bit.
__global__ void k_window_mask(const int32_t* __restrict__ lengths, int B, int Tmax, int window, int t, uint8_t* __restrict__ mask) {
  const int b = blockIdx.y;
  int lo, hi;
  attn_window_range(window, t, lengths[b], &lo, &hi);
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < Tmax; q += gridDim.x * blockDim.x)
    mask[(size_t)b * Tmax + q] = (q >= lo && q <= hi) ? 0 : 1;
}
}  // namespace

extern "C" int facppg_attention_window_mask(const int32_t* lengths_dev, int B, int Tmax, int window, int time_step,
                                            uint8_t* mask_dev, void* stream) {
  FACPPG_REQUIRE(lengths_dev && mask_dev, FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(B > 0 && Tmax > 0 && B <= 65535 && time_step >= 0, FACPPG_EINVAL, "bad sizes B=%d Tmax=%d t=%d", B, Tmax, time_step);
  k_window_mask<<<dim3((Tmax + 255) / 256, B), 256, 0, (hipStream_t)stream>>>(lengths_dev, B, Tmax, window, time_step, mask_dev);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

// Decoder.inference (model.py:489-535).
extern "C" int facppg_taco_decode(facppg_taco* h, const float* memory_dev, const float* pm_dev, const int32_t* lengths_dev,
                                  const int32_t* step_limits_dev, const uint8_t* masks_dev, uint64_t seed, int B, int Tin, int max_steps, float* mel_dev,
                                  float* gate_dev, float* align_dev, int32_t* out_lengths_dev, void* ws_, size_t ws_bytes,
                                  void* stream_) {
  FACPPG_REQUIRE(h && memory_dev && pm_dev && mel_dev && gate_dev && out_lengths_dev && ws_, FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(B > 0 && Tin > 0 && max_steps > 0, FACPPG_EINVAL, "bad B/Tin/max_steps");
  FACPPG_REQUIRE(Tin <= 8192, FACPPG_EUNSUPPORTED, "Tin > 8192 does not fit the decoder's LDS state");
  const facppg_taco_config& c = h->c;
  hipStream_t s = (hipStream_t)stream_;
  const DecWs w = dec_ws(c, B, max_steps);
  FACPPG_REQUIRE(ws_bytes >= w.total, FACPPG_EWORKSPACE, "decode workspace has %zu bytes, need %zu", ws_bytes, w.total);
  char* ws = (char*)ws_;
  const uint8_t* masks = masks_dev;
  const size_t nmask = (size_t)max_steps * 2 * B * c.prenet_dim;
  if (!masks) {
    k_random_mask<<<(unsigned)((nmask + 255) / 256), 256, 0, s>>>((uint8_t*)(ws + w.mask), nmask, seed ^ 0xD1B54A32D192ED03ull);
    masks = (const uint8_t*)(ws + w.mask);
  }
  DecArgs a;
  a.melx = nullptr; h->last_streamed = 0;
  a.dp0_t = h->dp0_t; a.dp1_t = h->dp1_t; a.att_t = h->att_t; a.att_b = h->att_b; a.dec_t = h->dec_t; a.dec_b = h->dec_b;
  a.q_t = h->q_t; a.proj_t = h->proj_t; a.proj_b = h->proj_b; a.loc_conv = h->loc_conv; a.loc_dense = h->loc_dense; a.v = h->v;
  a.xchg = (unsigned long long*)(ws + w.xchg);
  a.prof = getenv("FACPPG_DECODER_PROF") ? (long long*)(ws + w.prof) : nullptr;
  a.memory = memory_dev; a.pm = pm_dev; a.lengths = lengths_dev; a.step_limits = step_limits_dev; a.masks = masks; a.mel = mel_dev; a.gate = gate_dev;
  a.align = align_dev; a.out_len = out_lengths_dev;
  a.B = B; a.Tin = Tin; a.E = c.encoder_embedding_dim; a.P = c.prenet_dim; a.A = c.attention_rnn_dim; a.D = c.decoder_rnn_dim;
  a.AD = c.attention_dim; a.NF = c.n_acoustic_feat_dims; a.NFIL = c.attention_location_n_filters;
  a.KSZ = c.attention_location_kernel_size; a.window = c.attention_window_size; a.max_steps = max_steps;
  a.gate_thr = c.gate_threshold;
  const size_t smem = dec_lds_floats(a.P, a.E, a.A, a.D, a.NF, a.AD, a.NFIL, a.KSZ, Tin) * 4;
  FACPPG_REQUIRE(smem <= 160 * 1024 - 1024, FACPPG_EUNSUPPORTED, "decoder state (%zu bytes) exceeds LDS", smem);
  // latency mode (few utterances): NWG cooperating workgroups per utterance; throughput mode: one each
  // slice width: the narrowest (most workgroups per utterance) that keeps B * NWG co-resident (coop_limit)
  // the decoder holds a CU's whole LDS per workgroup: a caller that runs it UNDER another stream's kernels (the vocoder of
  // the previous batch, facppg.pipeline.synthesize_stream) bounds the CUs it takes away from them
  const int wg_limit = h->decoder_wg_limit > 0 && h->decoder_wg_limit < h->coop_limit ? h->decoder_wg_limit : h->coop_limit;
  const char* mode = getenv("FACPPG_DECODER_MODE");
  int variant = -1;
  const char* force_u = getenv("FACPPG_DECODER_COOP_U");   // tests / tuning: a specific slice width
  for (int v = 0; v < 5 && variant < 0; ++v)
    if ((long)B * h->coop_nwg[v] <= wg_limit && (!force_u || atoi(force_u) == h->coop_U[v])) variant = v;
  // beyond that the widest slices run the batch in chunks of co-resident utterances, one cooperative launch
  // after the other (0.31 ms per utterance at 200 frames; the one-workgroup kernel needs 0.45)
  int chunk = B;
  if (variant < 0 && !force_u && wg_limit / h->coop_nwg[4] >= 1) { variant = 4; chunk = wg_limit / h->coop_nwg[4]; }
  bool coop = variant >= 0;
  if (mode && !strcmp(mode, "single")) coop = false;
  if (mode && !strcmp(mode, "coop")) FACPPG_REQUIRE(coop, FACPPG_EUNSUPPORTED, "coop decoder needs B <= 120");
  if (coop) {
    a.att_coop = h->att_coop[variant]; a.dec_coop = h->dec_coop[variant]; a.U = h->coop_U[variant];
  }
  // split shape: split_nwk register-resident dense-layer workers serve NU utterances, each with its own main
  // (attention) workgroup.  NU = the fewest utterances per worker set that keeps every workgroup co-resident;
  // a worker's per-frame work grows with NU, so beyond SPLIT_MAX_NU the cooperative kernel wins.
  static const char* no_split = getenv("FACPPG_DECODER_NO_SPLIT");
  const char* max_nu_env = getenv("FACPPG_DECODER_SPLIT_MAX_NU");
  const int max_nu = max_nu_env ? atoi(max_nu_env) : 3;   // 4 utterances per worker set only ties with the cooperative kernel (12.5 vs 12.4 ms at B = 12)
  const size_t ustride = (size_t)(a.P + a.E + a.A) + (a.A + a.E + a.D) + (a.D + a.E) + round_up(a.NF, 4) + round_up(a.P, 4);
  int NU = 0;
  for (int nu = 1; nu <= SNU && nu <= max_nu && !NU; ++nu)
    if ((long)((B + nu - 1) / nu) * (h->split_nwk + nu) <= wg_limit && (nu * ustride + 1024) * 4 <= 150 * 1024) NU = nu;
  const bool split = coop && !no_split && NU > 0 && a.P + a.E + a.A <= SKP * SKR_LSTM &&
                     a.A + a.E + a.D <= SKP * SKR_LSTM && a.D + a.E <= SKP * SKR_PROJ && a.NF <= SKP * SKR_P1 &&
                     a.P <= SKP * SKR_P2 && a.NF + 1 <= h->split_nwk * SSC && a.P <= h->split_nwk * SSC &&
                     !(mode && !strcmp(mode, "coop"));
  if (mode && !strcmp(mode, "split")) FACPPG_REQUIRE(split, FACPPG_EUNSUPPORTED, "split decoder needs a small batch and the reference's layer widths");
  if (split) {
    a.att_w4 = h->att_w4; a.dec_w4 = h->dec_w4; a.xsplit = (unsigned long long*)(ws + w.xsplit);
    a.w1p_t = h->w1p_t; a.b1p = h->b1p;
    FACPPG_HIP_CHECK(hipMemsetAsync(ws + w.xchg, 0, w.total - w.xchg, s));
    size_t ssm = (NU * ustride + 1024) * 4;
    const size_t main_smem = (round_up((int)(smem / 4), 4) + (size_t)8 * NTC * 4) * 4;   // + attn_energy_pre's stash
    if (ssm < main_smem) ssm = main_smem;
    const void* fn = NU == 1 ? (const void*)k_decoder_split<1> : NU == 2 ? (const void*)k_decoder_split<2>
                   : NU == 3 ? (const void*)k_decoder_split<3> : (const void*)k_decoder_split<4>;
    FACPPG_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ssm));
    void* args[] = {(void*)&a};
    const int groups = (B + NU - 1) / NU;
    a.nwk = h->split_nwk;
    a.melx = (B == 1 && h->frame_stream && h->frame_stream_frames >= max_steps) ? h->frame_stream : nullptr;
    h->last_streamed = a.melx != nullptr;
    a.dbg_flags = getenv("FACPPG_DECODER_NO_ROWS") ? 1 : 0;
    FACPPG_HIP_CHECK(launch_coop(fn, dim3(h->split_nwk + NU, groups), dim3(NTC), args, ssm, s));
    h->last_mode = 2; h->last_wgs = (h->split_nwk + NU) * groups;
    if (a.prof) {
      long long pr[32];
      FACPPG_HIP_CHECK(hipMemcpyAsync(pr, a.prof, sizeof(pr), hipMemcpyDeviceToHost, s));
      FACPPG_HIP_CHECK(hipStreamSynchronize(s));
      fprintf(stderr, "[facppg split decoder prof (main), shader cycles] location features %lld energy_pre %lld entering row %lld query weights requested %lld wait_gate %lld wait_ah %lld attention %lld | att: query %lld energy %lld softmax %lld update %lld context %lld\n",
              pr[6], pr[7], pr[14], pr[2], pr[0], pr[3], pr[5], pr[8], pr[10], pr[11], pr[12], pr[13]);
      fprintf(stderr, "[facppg split decoder prof (worker 0), shader cycles] proj+prenet1 %lld | gather gate,X1 %lld | prenet2 %lld | gather X2 %lld | "
              "att LSTM %lld | gather AH %lld | gather CTX %lld | dec LSTM + gather DH %lld | barrier %lld\n",
              pr[16], pr[17], pr[18], pr[19], pr[20], pr[21], pr[22], pr[23], pr[24]);
    }
  } else
  if (coop) {
    FACPPG_HIP_CHECK(hipMemsetAsync(ws + w.xchg, 0, w.total - w.xchg, s));
    const void* fn = (const void*)k_decoder_coop;
    FACPPG_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    for (int b0 = 0; b0 < B; b0 += chunk) {
      const int nb = B - b0 < chunk ? B - b0 : chunk;
      int cv = variant;   // the last chunk may be small enough for narrower slices
      if (!force_u)
        for (int v = 0; v < variant; ++v)
          if ((long)nb * h->coop_nwg[v] <= wg_limit) { cv = v; break; }
      a.b0 = b0; a.att_coop = h->att_coop[cv]; a.dec_coop = h->dec_coop[cv]; a.U = h->coop_U[cv];
      void* args[] = {(void*)&a};
      FACPPG_HIP_CHECK(launch_coop(fn, dim3(h->coop_nwg[cv], nb), dim3(NTC), args, smem, s));
      h->last_mode = 1; h->last_wgs = h->coop_nwg[cv] * nb;
    }
    if (a.prof) {
      long long pr[16];
      FACPPG_HIP_CHECK(hipMemcpyAsync(pr, a.prof, sizeof(pr), hipMemcpyDeviceToHost, s));
      FACPPG_HIP_CHECK(hipStreamSynchronize(s));
      fprintf(stderr, "[facppg decoder prof, shader cycles] project %lld prenet %lld att_slice %lld sync1 %lld gather %lld attention %lld dec_slice %lld sync2 %lld | att: query %lld feat %lld energy %lld softmax %lld update %lld context %lld\n",
              pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6], pr[7], pr[8], pr[9], pr[10], pr[11], pr[12], pr[13]);
    }
  } else {
    FACPPG_HIP_CHECK(hipFuncSetAttribute((const void*)k_decoder, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    k_decoder<<<B, NT, smem, s>>>(a);
    h->last_mode = 0; h->last_wgs = B;
  }
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

// Postnet + residual (model.py:178-184, 604-605): mel_post = mel + postnet(mel).
extern "C" int facppg_taco_postnet(facppg_taco* h, const float* mel_dev, const int32_t* out_lengths_dev, int B, int T, int ld,
                                   float* mel_post_dev, void* ws_, size_t ws_bytes, void* stream_) {
  FACPPG_REQUIRE(h && mel_dev && mel_post_dev && ws_, FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(B > 0 && T > 0 && ld >= T, FACPPG_EINVAL, "bad B/T/ld");
  const facppg_taco_config& c = h->c;
  const int PE = c.postnet_embedding_dim, NF = c.n_acoustic_feat_dims, n = c.postnet_n_convolutions;
  const size_t need = facppg_taco_postnet_workspace_bytes(h, B, T);
  FACPPG_REQUIRE(ws_bytes >= need, FACPPG_EWORKSPACE, "workspace has %zu bytes, need %zu", ws_bytes, need);
  hipStream_t s = (hipStream_t)stream_;
  float* buf[2] = {(float*)ws_, (float*)ws_ + (size_t)B * PE * T};
  float* skws = (float*)ws_ + (size_t)2 * B * PE * T;
  const size_t skws_bytes = need - (size_t)2 * B * PE * T * 4;
  const float* cur = mel_dev;
  long cur_bs = (long)NF * ld;
  int cur_ld = ld;
  for (int j = 0; j < n; ++j) {
    const int ci = j == 0 ? NF : PE, co = j == n - 1 ? NF : PE;
    GemmArgs g;
    g.splitk_ws = skws; g.splitk_ws_bytes = skws_bytes;
    g.B = B; g.N = T; g.n_valid = out_lengths_dev; g.A = h->post[j]; g.M = co; g.Cin = ci; g.taps = c.postnet_kernel_size;
    g.pad = (c.postnet_kernel_size - 1) / 2; g.X = cur; g.x_bs = cur_bs; g.ldx = cur_ld; g.bias = h->post_b[j];
    g.scale = h->post_scale[j]; g.shift = h->post_shift[j];
    if (j < n - 1) {
      g.act = ACT_TANH; g.C = buf[j & 1]; g.c_bs = (long)PE * T; g.ldc = T;
    } else {
      g.res = mel_dev; g.res_bs = (long)NF * ld; g.ldres = ld; g.C = mel_post_dev; g.c_bs = (long)NF * ld; g.ldc = ld;
    }
    if (int rc = gemm_launch(g, s)) return rc;
    cur = g.C; cur_bs = g.c_bs; cur_ld = g.ldc;
  }
  return FACPPG_OK;
}

extern "C" size_t facppg_taco_postnet_workspace_bytes(const facppg_taco* h, int B, int T) {
  if (!h || B <= 0 || T <= 0) return 0;
  const size_t sk = (size_t)16 * B * h->c.postnet_embedding_dim * T * 4;   // split-K partial sums
  return (size_t)2 * B * h->c.postnet_embedding_dim * T * 4 + sk;
}

extern "C" int facppg_taco_set_decoder_workgroups(facppg_taco* h, int max_workgroups) {
  FACPPG_REQUIRE(h && max_workgroups >= 0, FACPPG_EINVAL, "NULL handle or negative limit");
  h->decoder_wg_limit = max_workgroups;
  return FACPPG_OK;
}

// ------------------------------------------------------------------------------------------
// Streaming the decoder's frames into the postnet while the decoder is still running (B = 1, split decoder).
// ------------------------------------------------------------------------------------------
namespace {
// frames [fa, fb) of the tagged stream -> dst[row][f] (channel-major, ld).  A word is waited for at most `limit` wall-clock ticks
// (FACPPG_STREAM_WAIT_MS, default 100 ms: the decoder emits a frame every 20 us); a frame that does not show up in that time does
// NOT trap -- the block is declared void like one the decoder stopped short of, and the caller's tail covers its frames: a
// profiler that serialises kernels (rocprofv3 counter passes) may run this launch BEFORE the decoder it waits for, and the
// result must then be late, not wrong or fatal.  If the decoder has stopped short of a frame (out_len set, frame >= out_len) the whole block is void:
// *void_flag = 1 -- every launch of THIS block that is handed the flag does nothing, and the caller handles those frames once it
// knows the length.  A block behind a void block is void as well (prev_flag).  One flag per block: blocks overlap in time (the
// seed pass of block k runs while block k + 1 is being collected), a shared flag would cut a valid block's launches short.
__global__ __launch_bounds__(256) void k_collect_frames(const unsigned long long* __restrict__ melx, const int* __restrict__ out_len,
                                                        int NF, int fa, int fb, float* __restrict__ dst, int ld, int* void_flag,
                                                        const int* prev_flag, unsigned long long limit) {
  if (prev_flag && __hip_atomic_load(prev_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
    if (void_flag) __hip_atomic_store(void_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const int n = (fb - fa) * NF;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int f = fa + i / NF, row = i - (f - fa) * NF;
    const unsigned long long* w = melx + (size_t)f * NF + row;
    unsigned long long v, t0 = 0;
    unsigned spins = 0;
    for (;;) {
      v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)(v >> 32) == (unsigned)(f + 1)) break;
      const int ol = __hip_atomic_load(out_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (ol > 0 && f >= ol) {   // the decoder stopped before this frame
        if (void_flag) __hip_atomic_store(void_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
      __builtin_amdgcn_s_sleep(32);
      if ((++spins & 255u) == 0 && limit) {
        const unsigned long long now = wall_clock64();
        if (!t0) t0 = now;
        else if (now - t0 > limit) {   // the frames are not coming (in time): leave them to the caller's tail
          if (void_flag) __hip_atomic_store(void_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          return;
        }
      }
    }
    dst[(size_t)row * ld + f] = __uint_as_float((unsigned)v);
  }
}
}  // namespace

extern "C" int facppg_taco_set_frame_stream(facppg_taco* h, void* words_dev, int frames) {
  FACPPG_REQUIRE(h && frames >= 0 && (words_dev || frames == 0), FACPPG_EINVAL, "NULL handle, or a NULL buffer with frames > 0");
  h->frame_stream = (unsigned long long*)words_dev; h->frame_stream_frames = words_dev ? frames : 0;
  return FACPPG_OK;
}

extern "C" int facppg_taco_last_decode_streamed(const facppg_taco* h, int* streamed) {
  FACPPG_REQUIRE(h && streamed, FACPPG_EINVAL, "NULL argument");
  *streamed = h->last_streamed;
  return FACPPG_OK;
}

extern "C" int facppg_taco_collect_frames(const facppg_taco* h, const void* words_dev, const int32_t* out_length_dev, int frame_a,
                                          int frame_b, float* mel_dev, int ld, int32_t* void_flag_dev, const int32_t* prev_flag_dev,
                                          void* stream_) {
  FACPPG_REQUIRE(h && words_dev && out_length_dev && mel_dev, FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(frame_a >= 0 && frame_b > frame_a && frame_b <= ld, FACPPG_EINVAL, "frames [%d, %d) with ld %d", frame_a, frame_b, ld);
  const int n = (frame_b - frame_a) * h->c.n_acoustic_feat_dims;
  const char* wait_env = getenv("FACPPG_STREAM_WAIT_MS");
  const double wait_ms = wait_env ? strtod(wait_env, nullptr) : 100.0;
  // (without a void flag -- the caller's tail, behind the decoder on its own stream -- the wait is the process-wide poll limit's)
  const unsigned long long limit = void_flag_dev && wait_ms > 0 && h->wall_khz > 0 ? (unsigned long long)(wait_ms * h->wall_khz) : 0ull;
  k_collect_frames<<<(n + 255) / 256 < 8 ? (n + 255) / 256 : 8, 256, 0, (hipStream_t)stream_>>>(
      (const unsigned long long*)words_dev, out_length_dev, h->c.n_acoustic_feat_dims, frame_a, frame_b, mel_dev, ld, void_flag_dev,
      prev_flag_dev, limit);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

// Postnet + residual (model.py:178-184, 604-605) as a STREAMING convolution stack: every layer is extended by the columns
// whose inputs have become final.  With f frames of mel known, layer j (1-based, kernel 2*pad + 1) is final up to f - j*pad;
// a call advances the frontier from f_prev to f_new frames (final_T > 0: the utterance has ended at final_T frames, every
// layer runs up to it and the convolutions' zero padding starts there).  Every output column is the same sum, in the same
// order, as in facppg_taco_postnet's one-shot launch (the split-K factor depends on K alone): same bits.
extern "C" size_t facppg_taco_postnet_stream_workspace_bytes(const facppg_taco* h, int max_frames) {
  if (!h || max_frames <= 0) return 0;
  const size_t PE = h->c.postnet_embedding_dim, n = h->c.postnet_n_convolutions;
  return (n - 1) * PE * max_frames * 4 + (size_t)16 * PE * max_frames * 4;
}

extern "C" int facppg_taco_postnet_range(facppg_taco* h, const float* mel_dev, int ld, int f_prev, int f_new, int final_T,
                                         float* mel_post_dev, int ld_post, void* ws_, size_t ws_bytes, int max_frames,
                                         const int32_t* skip_dev, void* stream_) {
  FACPPG_REQUIRE(h && mel_dev && mel_post_dev && ws_, FACPPG_EINVAL, "NULL argument");
  const facppg_taco_config& c = h->c;
  const int PE = c.postnet_embedding_dim, NF = c.n_acoustic_feat_dims, n = c.postnet_n_convolutions, pad = (c.postnet_kernel_size - 1) / 2;
  FACPPG_REQUIRE(f_prev >= 0 && f_new >= f_prev && f_new <= max_frames && max_frames <= ld && final_T <= max_frames, FACPPG_EINVAL,
                 "bad frame range [%d, %d) / final %d / max %d", f_prev, f_new, final_T, max_frames);
  FACPPG_REQUIRE(final_T == 0 || final_T == f_new, FACPPG_EINVAL, "a final call covers exactly the utterance's frames");
  const size_t need = facppg_taco_postnet_stream_workspace_bytes(h, max_frames);
  FACPPG_REQUIRE(ws_bytes >= need, FACPPG_EWORKSPACE, "workspace has %zu bytes, need %zu", ws_bytes, need);
  hipStream_t s = (hipStream_t)stream_;
  float* bufs = (float*)ws_;
  float* skws = bufs + (size_t)(n - 1) * PE * max_frames;
  const size_t skws_bytes = (size_t)16 * PE * max_frames * 4;
  int src_hi = f_new;
  for (int j = 0; j < n; ++j) {
    const int lag = pad * (j + 1);
    const int lo = f_prev - lag > 0 ? f_prev - lag : 0;
    const int hi = final_T > 0 ? final_T : f_new - lag;
    if (hi > lo) {
      const int ci = j == 0 ? NF : PE, co = j == n - 1 ? NF : PE;
      GemmArgs g;
      g.splitk_ws = skws; g.splitk_ws_bytes = skws_bytes;
      g.B = 1; g.N = hi; g.col0 = lo; g.src_hi = src_hi; g.skip = skip_dev;
      g.A = h->post[j]; g.M = co; g.Cin = ci; g.taps = c.postnet_kernel_size; g.pad = pad;
      g.X = j == 0 ? mel_dev : bufs + (size_t)(j - 1) * PE * max_frames; g.ldx = j == 0 ? ld : max_frames;
      g.bias = h->post_b[j]; g.scale = h->post_scale[j]; g.shift = h->post_shift[j];
      if (j < n - 1) { g.act = ACT_TANH; g.C = bufs + (size_t)j * PE * max_frames; g.ldc = max_frames; }
      else { g.res = mel_dev; g.ldres = ld; g.C = mel_post_dev; g.ldc = ld_post; }
      if (int rc = gemm_launch(g, s)) return rc;
    }
    src_hi = hi > 0 ? hi : 0;   // the next layer reads this one's final columns only
    if (src_hi == 0) break;
  }
  return FACPPG_OK;
}

extern "C" int facppg_taco_last_decoder_launch(const facppg_taco* h, int* mode, int* workgroups) {
  FACPPG_REQUIRE(h && mode && workgroups, FACPPG_EINVAL, "NULL argument");
  *mode = h->last_mode; *workgroups = h->last_wgs;
  return FACPPG_OK;
}

extern "C" size_t facppg_taco_decode_workspace_bytes(const facppg_taco* h, int B, int max_steps) {
  if (!h || B <= 0 || max_steps <= 0) return 0;
  return dec_ws(h->c, B, max_steps).total;
}
