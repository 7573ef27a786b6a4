// nnet3 TDNN acoustic-model inference for MI355X (gfx950): feature frames -> senone posteriors (PPGs).
//
// Replaces the reference's compute_full_ppg (src/ppg/compute_ppg.py:42-70), which drives Kaldi's
// nnet3::DecodableNnetSimple through PyKaldi: set_batchnorm_test_mode / collapse_model, then the network output for
// every frame, with the input replicated at the utterance edges for the model's left / right context.
//
// The host side (common/nnet3.py plan_layers) hands over a chain of fused layers, each
//     y[:, t] = act(W . [x[:, t + first]; x[:, t + first + dil]; ...; x[:, t + first + (taps-1) dil]] + b)   (+ renorm)
// with test-mode BatchNorm already folded into the consuming layer.  Activations are channel-major [C][Tp],
// Tp = L + T + R frames (positions contiguous: coalesced loads, the layout k_gemm wants); every layer is the exact-fp32
// MFMA tapped GEMM of facppg_gemm.hip.  k_gemm reads columns n + tap*dil (pad = 0), so layer l stores true frame t at
// column t - s_l with s_l = s_{l-1} - first_l, s_0 = -L: the shifts telescope to s_last = 0, i.e. the last layer's
// columns 0..T-1 are frames 0..T-1, and the columns polluted by the zero fill past Tp lie outside every frame's
// dependency cone.  Bound: MFMA for the two big products (input splice x hidden, hidden x senones), latency otherwise.
#include <cmath>
#include <new>
#include <vector>

#include "facppg_gemm.h"

using namespace facppg;

struct facppg_tdnn {
  int device = 0;
  int n_layers = 0, final_op = 0, in_dim = 0, out_dim = 0, left = 0, right = 0, max_dim = 0;
  std::vector<facppg_tdnn_layer> layers;
  std::vector<float4*> A;
  std::vector<float*> bias;
  float* arena = nullptr;
};

namespace {

// feats [T][D] row-major -> x0 [D][Tp], x0[c][j] = feats[clamp(j - L, 0, T-1)][c]  (DecodableNnetSimple replicates the
// first / last frame for the context beyond the utterance)
__global__ void k_tdnn_input(const float* __restrict__ feats, float* __restrict__ x0, int T, int D, int L, int Tp) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (j >= Tp) return;
  const int t = min(max(j - L, 0), T - 1);
  x0[(size_t)c * Tp + j] = feats[(size_t)t * D + c];
}

// NormalizeComponent (nnet-normalize-component.cc): y = x * (max(sum x^2 / (C * rms^2), 2^-66))^(-1/2), per frame
__global__ void k_tdnn_renorm(float* __restrict__ x, int C, int Tp, float target_rms) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Tp) return;
  float ss = 0.0f;
  for (int c = 0; c < C; ++c) { const float v = x[(size_t)c * Tp + j]; ss = fmaf(v, v, ss); }
  const float floor_ = 1.3552527156068805e-20f;   // 2^-66 (kSquaredNormFloor)
  const float scale = 1.0f / sqrtf(fmaxf(ss / ((float)C * target_rms * target_rms), floor_));
  for (int c = 0; c < C; ++c) x[(size_t)c * Tp + j] *= scale;
}

// (Log)Softmax over the M channels of frames 0..T-1; writes out [T][M] row-major.  final_op: 0 copy, 1 softmax, 2 log-softmax
__global__ void k_tdnn_output(const float* __restrict__ y, int M, int Tp, int T, int final_op, float* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  float mx = -INFINITY, sum = 0.0f;
  if (final_op != 0) {
    for (int m = 0; m < M; ++m) mx = fmaxf(mx, y[(size_t)m * Tp + t]);
    for (int m = 0; m < M; ++m) sum += expf(y[(size_t)m * Tp + t] - mx);
  }
  const float lse = logf(sum);
  for (int m = 0; m < M; ++m) {
    const float v = y[(size_t)m * Tp + t];
    out[(size_t)t * M + m] = final_op == 0 ? v : final_op == 1 ? expf(v - mx) / sum : v - mx - lse;
  }
}

struct TdnnWs { size_t x[2], splitk, splitk_bytes, total; int Tp; };
TdnnWs tdnn_ws(const facppg_tdnn* h, int T) {
  TdnnWs w;
  w.Tp = h->left + T + h->right;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  w.x[0] = take((size_t)h->max_dim * w.Tp * 4);
  w.x[1] = take((size_t)h->max_dim * w.Tp * 4);
  w.splitk_bytes = (size_t)16 * h->max_dim * w.Tp * 4;
  w.splitk = take(w.splitk_bytes);
  w.total = off;
  return w;
}

}  // namespace

extern "C" size_t facppg_tdnn_weight_count(const facppg_tdnn_layer* layers, int n_layers) {
  if (!layers || n_layers <= 0) return 0;
  size_t n = 0;
  for (int i = 0; i < n_layers; ++i) {
    const facppg_tdnn_layer& l = layers[i];
    if (l.out_dim <= 0 || l.in_dim <= 0 || l.taps <= 0 || l.dil <= 0) return 0;
    n += (size_t)l.out_dim * l.taps * l.in_dim + l.out_dim;
  }
  return n;
}

extern "C" int facppg_tdnn_create(const facppg_tdnn_layer* layers, int n_layers, int final_op, const float* weights_dev,
                                  size_t n_weights, int device, void* stream_, facppg_tdnn** out) {
  FACPPG_REQUIRE(layers && weights_dev && out && n_layers > 0, FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(final_op >= 0 && final_op <= 2, FACPPG_EINVAL, "final_op must be 0 (none), 1 (softmax) or 2 (log-softmax)");
  FACPPG_REQUIRE(n_weights == facppg_tdnn_weight_count(layers, n_layers) && n_weights > 0, FACPPG_EINVAL,
                 "weight blob has %zu values, the layer table needs %zu", n_weights, facppg_tdnn_weight_count(layers, n_layers));
  hipStream_t s = (hipStream_t)stream_;
  facppg_tdnn* h = new (std::nothrow) facppg_tdnn;
  FACPPG_REQUIRE(h, FACPPG_EHIP, "out of host memory");
  h->device = device; h->n_layers = n_layers; h->final_op = final_op;
  h->layers.assign(layers, layers + n_layers);
  h->in_dim = layers[0].in_dim; h->out_dim = layers[n_layers - 1].out_dim; h->max_dim = h->in_dim;
  size_t a_f4 = 0, b_f = 0;
  for (int i = 0; i < n_layers; ++i) {
    const facppg_tdnn_layer& l = layers[i];
    if (i > 0 && l.in_dim != layers[i - 1].out_dim) {
      set_error("layer %d reads %d channels, layer %d writes %d", i, l.in_dim, i - 1, layers[i - 1].out_dim);
      delete h;
      return FACPPG_EINVAL;
    }
    // context of the chain: frame t of layer i needs frames t + first .. t + first + (taps-1) dil of layer i-1
    h->left += l.first < 0 ? -l.first : 0;
    const int hi = l.first + (l.taps - 1) * l.dil;
    h->right += hi > 0 ? hi : 0;
    if (l.first > 0 || hi < 0) {
      set_error("layer %d: a splice that excludes the current frame's side (first %d, last %d) is not supported", i, l.first, hi);
      delete h;
      return FACPPG_EUNSUPPORTED;
    }
    h->max_dim = l.out_dim > h->max_dim ? l.out_dim : h->max_dim;
    a_f4 += packed_a_float4s(l.out_dim, l.in_dim * l.taps);
    b_f += (size_t)round_up(l.out_dim, 64);
  }
  if (hipMalloc(&h->arena, a_f4 * 16 + b_f * 4) != hipSuccess) {
    set_error("hipMalloc of %zu bytes of packed TDNN weights failed", a_f4 * 16 + b_f * 4);
    delete h;
    return FACPPG_EHIP;
  }
  float4* ap = (float4*)h->arena;
  float* bp = (float*)(ap + a_f4);
  const float* src = weights_dev;
  for (int i = 0; i < n_layers; ++i) {
    const facppg_tdnn_layer& l = layers[i];
    const int K = l.taps * l.in_dim;
    // blob layout per layer: W [out][taps * in] (tap-major: Kaldi's Append order), then b [out]
    const int rc = pack_a_strided(src, l.out_dim, l.in_dim, l.taps, K, 1, l.in_dim, 0, ap, s);
    if (rc != FACPPG_OK || hipMemcpyAsync(bp, src + (size_t)l.out_dim * K, (size_t)l.out_dim * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) {
      if (rc == FACPPG_OK) set_error("copy of the layer-%d bias failed", i);
      (void)hipFree(h->arena);
      delete h;
      return FACPPG_EHIP;
    }
    h->A.push_back(ap);
    h->bias.push_back(bp);
    ap += packed_a_float4s(l.out_dim, K);
    bp += round_up(l.out_dim, 64);
    src += (size_t)l.out_dim * K + l.out_dim;
  }
  if (hipStreamSynchronize(s) != hipSuccess) {   // the caller may free weights_dev when this returns
    set_error("packing the TDNN weights failed");
    (void)hipFree(h->arena);
    delete h;
    return FACPPG_EHIP;
  }
  *out = h;
  return FACPPG_OK;
}

extern "C" void facppg_tdnn_destroy(facppg_tdnn* h) {
  if (!h) return;
  if (h->arena) (void)hipFree(h->arena);
  delete h;
}

extern "C" int facppg_tdnn_context(const facppg_tdnn* h, int* left, int* right) {
  FACPPG_REQUIRE(h && left && right, FACPPG_EINVAL, "NULL argument");
  *left = h->left; *right = h->right;
  return FACPPG_OK;
}

extern "C" size_t facppg_tdnn_workspace_bytes(const facppg_tdnn* h, int T) {
  if (!h || T <= 0) return 0;
  return tdnn_ws(h, T).total;
}

extern "C" int facppg_tdnn_forward(facppg_tdnn* h, const float* feats_dev, int T, float* out_dev, void* ws_, size_t ws_bytes,
                                   void* stream_) {
  FACPPG_REQUIRE(h && feats_dev && out_dev && ws_, FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(T > 0, FACPPG_EINVAL, "T must be positive (got %d)", T);
  const TdnnWs w = tdnn_ws(h, T);
  FACPPG_REQUIRE(ws_bytes >= w.total, FACPPG_EWORKSPACE, "workspace has %zu bytes, need %zu", ws_bytes, w.total);
  hipStream_t s = (hipStream_t)stream_;
  char* ws = (char*)ws_;
  float* x[2] = {(float*)(ws + w.x[0]), (float*)(ws + w.x[1])};
  const int Tp = w.Tp;
  k_tdnn_input<<<dim3((Tp + 255) / 256, h->in_dim), 256, 0, s>>>(feats_dev, x[0], T, h->in_dim, h->left, Tp);
  int cur = 0;
  for (int i = 0; i < h->n_layers; ++i) {
    const facppg_tdnn_layer& l = h->layers[i];
    GemmArgs g;
    g.A = h->A[i]; g.M = l.out_dim; g.Cin = l.in_dim; g.taps = l.taps; g.dil = l.dil; g.pad = 0;
    g.X = x[cur]; g.ldx = Tp; g.N = Tp; g.bias = h->bias[i]; g.act = l.relu ? ACT_RELU : ACT_NONE;
    g.C = x[cur ^ 1]; g.ldc = Tp; g.B = 1;
    g.splitk_ws = (float*)(ws + w.splitk); g.splitk_ws_bytes = w.splitk_bytes;
    const int rc = gemm_launch(g, s);
    if (rc != FACPPG_OK) return rc;
    cur ^= 1;
    if (l.renorm_target_rms > 0.0f) k_tdnn_renorm<<<(Tp + 255) / 256, 256, 0, s>>>(x[cur], l.out_dim, Tp, l.renorm_target_rms);
  }
  k_tdnn_output<<<(T + 63) / 64, 64, 0, s>>>(x[cur], h->out_dim, Tp, T, h->final_op, out_dev);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}
