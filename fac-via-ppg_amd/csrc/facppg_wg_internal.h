// Internal to libfacppg_hip.so (not part of the C ABI): the WaveGlow inference handle and what the two translation
// units that run it share -- facppg_wg.hip (one launch per WaveNet layer, any batch) and facppg_wgp.hip (ONE persistent
// launch for a single short utterance, the metric's "batch = 1" case).
#pragma once
#include <cstdint>
#include <vector>

#include "facppg_common.h"

namespace facppg {

constexpr int C = 256;        // WN channels (WN_config.n_channels)
constexpr int MAXF = 32;      // max flows
constexpr int NMEL = 80;      // mel channels (NCOND / n_group)
constexpr int HQ = 16;        // zero margin in frames on both sides of a phase row (>= 128 / (hop/8) + 1)
constexpr int KCH = 64;       // K rows per LDS chunk of the per-layer kernels

// K order inside a 16-wide group of the v_mfma_f32_16x16x4_f32 kernels: MFMA s (0..3), lane quarter kq (0..3) -> k16(s, kq).
// It is chosen so that the running sum meets the K entries in the same sequence as the 32x32x2 kernels do
// (0,4,1,5,2,6,3,7 within each group of 8): a tile then gets the same bits from either kind of kernel.
__device__ __forceinline__ int k16(int s, int kq) { return 8 * (s >> 1) + 2 * (s & 1) + (kq >> 1) + 4 * (kq & 1); }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// tanh(a) * sigmoid(b) with two hardware exponentials and one reciprocal:
//   (1 - e^-2a) / ((1 + e^-2a) (1 + e^-b)).  tanh saturates to +-1 in fp32 beyond |a| > 9.02, so
// clamping a to +-15 keeps e^-2a finite without changing the result; e^-b -> inf gives 0 as it must.
// Relative error ~1e-6 (v_exp_f32 / v_rcp_f32 are 1-ulp), i.e. fp32-roundoff class for this path;
// the libm tanhf/expf pair it replaces cost ~150 VALU instructions per element (16 % of the kernel).
__device__ __forceinline__ float gate_tanh_sigmoid(float a, float b) {
  const float ea = __expf(-2.0f * fminf(fmaxf(a, -15.0f), 15.0f));
  const float eb = __expf(-b);
  // v_rcp_f32 directly: hipcc expands __fdividef to the full IEEE division sequence (v_div_scale x2, v_div_fmas, v_div_fixup)
  return (1.0f - ea) * __builtin_amdgcn_rcpf((1.0f + ea) * (1.0f + eb));
}

struct WgpState;   // facppg_wgp.hip: images, tables and launch state of the persistent small-launch path (null: not built)

}  // namespace facppg

struct facppg_wg {
  facppg_wg_config cfg;
  int device;
  int n_rem[facppg::MAXF], n_half[facppg::MAXF], early[facppg::MAXF];
  char* arena;  // one device allocation holding everything below
  size_t arena_bytes;
  float *up_w, *up_b;
  float *start_w[facppg::MAXF], *start_b[facppg::MAXF], *end_w[facppg::MAXF], *end_b[facppg::MAXF], *winv[facppg::MAXF], *wfwd[facppg::MAXF];
  float4* w1[facppg::MAXF][8];
  float4* w2[facppg::MAXF][8];
  float *b1[facppg::MAXF][8], *b2[facppg::MAXF][8];
  // phase-major inference images (k_wn_layer<PM>): convolution part, folded conditioning per phase, folded bias
  int P, nj, kc, kcp;
  float4* w1pm[facppg::MAXF][8];
  float4* wcpm[facppg::MAXF][8];
  float* b1pm[facppg::MAXF][8];
  float4 *w1_16[facppg::MAXF][8], *wc_16[facppg::MAXF][8], *w2_16[facppg::MAXF][8];   // the same weights as k_wn_layer16's 16x16x4 images
  // folded flow edges (k_fold_end_rows / k_fold_first): end-row image per layer, folded end bias per flow, the first layer's
  // folded tap image (both lane orders), res-rows-only images of the non-last res_skip convs (both lane orders)
  float* we[facppg::MAXF][8];
  float* endb[facppg::MAXF];
  float4 *w1f[facppg::MAXF], *w1f_16[facppg::MAXF];
  float4 *w2r[facppg::MAXF][8], *w2r_16[facppg::MAXF][8];
  int profiling;
  std::vector<hipEvent_t> ev;  // pairs around each k_wn_layer launch
  int ev_used;
  int ev_layers;               // layers one event pair brackets (a flow's back-to-back layer launches)
  void* ltab[facppg::MAXF];    // device tables of a flow's per-layer operands (WnLayerPtrs[8] per flow; k_cond_seed)
  unsigned long long poll_limit;   // wall-clock ticks an in-launch wait may last before it traps
  int n_cu;
  facppg::WgpState* wgp;   // persistent small-launch path (facppg_wgp.hip), or null
  int last_tile, last_waves, last_tiles;   // shape of the WN layer launches of the most recent infer (facppg_wg_last_launch_shape)
};

namespace facppg {

// ---- facppg_wgp.hip: WaveGlow.infer of ONE short utterance as ONE persistent launch
// What wgp_create packs per (flow, layer), from buffers that exist only inside facppg_wg_create:
struct WgpLayerSrc {
  const float* in_w;     // [512][256][3]  dilated conv
  const float* folded;   // [512][P * kcp] folded conditioning matrices of every phase (Wc . U)
  const float* rs_w;     // [512 or 256][256]  res_skip conv (res rows first)
  const float* b1pm;     // [512] folded gate bias (in + cond + upsample bias through cond)
  const float* b2;       // [512 or 256] res_skip bias
  const float* f0;       // [512][64] the first layer's taps through the start conv (k_fold_first), flows' first layers only
};
size_t wgp_arena_bytes(const facppg_wg_config& c, int device);                  // 0: the configuration / device is not eligible
int wgp_create(facppg_wg* h, char* arena, hipStream_t s);                       // tables; images are packed layer by layer:
int wgp_pack_layer(facppg_wg* h, int k, int i, const WgpLayerSrc& src, hipStream_t s);
int wgp_finish_create(facppg_wg* h, hipStream_t s);                             // uploads the tables (after every layer was packed)
void wgp_destroy(facppg_wg* h);
bool wgp_eligible(const facppg_wg* h, int B, int T, const int32_t* T_valid_dev);
size_t wgp_workspace_bytes(const facppg_wg* h, int B, int T);
int wgp_infer(facppg_wg* h, const float* mel_dev, const float* z_dev, uint64_t seed, float sigma, int T, float* audio_dev,
              char* ws, hipStream_t s);
// facppg_wg.hip: z[0..n) ~ N(0, 1), Philox4x32-10 keyed by `seed` (the noise of facppg_wg_infer when none is injected)
void wg_launch_noise(float* z, size_t n, uint64_t seed, hipStream_t s);

}  // namespace facppg
