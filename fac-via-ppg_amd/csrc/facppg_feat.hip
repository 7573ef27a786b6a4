// The blob-free part of the PPG front-end: the acoustic model's input features
//   wav -> MFCC(13) -> cepstral mean normalisation -> splice +-3 frames -> LDA (40 x 91)     (src/ppg/compute_ppg.py:97-134,
//   src/common/feat.py:74-156; Kaldi feature-window.cc / feature-mfcc.cc / mel-computations.cc behind pykaldi)
// and the senone -> monophone reduction of a PPG (compute_ppg.py:73-94).  The nnet3 acoustic model between the two is a
// binary blob the reference does not ship (data/am/final.raw), so the chain stops at its input.
//
// MFCC as three products.  Per frame Kaldi removes the DC offset, pre-emphasises, applies the povey window, zero-pads to 512
// and takes the FFT: every step is linear in the raw frame, so the host folds them into ONE [2*257 x 400] matrix
// (facppg_mfcc_create) and the frames go through the exact-fp32 MFMA GEMM (k_gemm); power spectrum, 23 mel bins, log, DCT and
// lifter are a fused per-frame kernel (k_mfcc_tail); frame extraction with Kaldi's snip_edges=false reflection is k_mfcc_frames.
#include <new>

#include "facppg_gemm.h"

struct facppg_mfcc {
  int frame_length, frame_shift, nbins, n_mel, n_ceps, device;
  float4* basis;     // packed A operand [2*nbins][frame_length]
  float* mel;        // [n_mel][nbins]
  float* dct;        // [n_ceps][n_mel] (lifter folded in)
  char* arena;
};

namespace facppg {
namespace {

// frames[k][t] = wav[reflect(first(t) + k)], first(t) = t*shift + shift/2 - len/2  (feature-window.cc: snip_edges = false);
// energy[t] = log(max(sum_k (x_k - mean)^2, floor))  (raw_energy, after DC removal) for use_energy
__global__ void k_mfcc_frames(const float* __restrict__ wav, int N, int T, int len, int shift, float* __restrict__ frames,
                              float* __restrict__ energy, int Tld) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int first = t * shift + shift / 2 - len / 2;
  float s = 0.f, s2 = 0.f;
  for (int k = 0; k < len; ++k) {
    int i = first + k;
    while (i < 0 || i >= N) i = i < 0 ? -i - 1 : 2 * N - 1 - i;
    const float v = wav[i];
    frames[(size_t)k * Tld + t] = v;
    s += v; s2 = fmaf(v, v, s2);
  }
  if (energy) {
    const float e = s2 - s * s / (float)len;
    energy[t] = logf(fmaxf(e, 1.1920929e-7f));   // max(energy, FLT_EPSILON), feature-window.cc ProcessWindow
  }
}

// spec [2*nbins][Tld] (real rows, then imaginary rows) -> mfcc [T][n_ceps]; one wavefront per frame
__global__ __launch_bounds__(256) void k_mfcc_tail(const float* __restrict__ spec, int Tld, int T, int nbins, const float* __restrict__ mel,
                                                   int n_mel, const float* __restrict__ dct, int n_ceps, const float* __restrict__ energy,
                                                   float* __restrict__ out) {
  extern __shared__ float sm[];   // per wave: power [nbins] + logmel [n_mel]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, t = blockIdx.x * 4 + wave;
  float* pw = sm + wave * (nbins + n_mel);
  float* lm = pw + nbins;
  if (t < T)
    for (int k = lane; k < nbins; k += 64) {
      const float re = spec[(size_t)k * Tld + t], im = spec[(size_t)(nbins + k) * Tld + t];
      pw[k] = re * re + im * im;
    }
  __syncthreads();
  if (t < T)
    for (int m = lane; m < n_mel; m += 64) {
      float v = 0.f;
      for (int k = 0; k < nbins; ++k) v = fmaf(mel[m * nbins + k], pw[k], v);
      lm[m] = logf(fmaxf(v, 1.1920929e-7f));      // ApplyFloor(FLT_EPSILON); ApplyLog()
    }
  __syncthreads();
  if (t < T)
    for (int c = lane; c < n_ceps; c += 64) {
      float v = 0.f;
      for (int m = 0; m < n_mel; ++m) v = fmaf(dct[c * n_mel + m], lm[m], v);
      if (energy && c == 0) v = energy[t];          // use_energy: C0 replaced by the log energy
      out[(size_t)t * n_ceps + c] = v;
    }
}

// column means of feats [T][D] -> mean[D] (fixed order: one workgroup, strided partial sums then a tree)
__global__ __launch_bounds__(256) void k_col_mean(const float* __restrict__ x, int T, int D, float* __restrict__ mean) {
  __shared__ float red[256];
  for (int d = 0; d < D; ++d) {
    float v = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) v += x[(size_t)t * D + d];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) mean[d] = red[0] / (float)T;
    __syncthreads();
  }
}

// out[t][m] = off[m] + sum_{j=-left..right} sum_d A[m][(j+left)*D + d] * (x[clamp(t+j)][d] - mean[d])   (transform given)
// out[t][(j+left)*D + d] = x[clamp(t+j)][d] - mean[d]                                                    (transform null: splice only)
__global__ void k_cmn_splice_transform(const float* __restrict__ x, int T, int D, const float* __restrict__ mean, int left, int right,
                                       const float* __restrict__ A, int M, int cols, float* __restrict__ out) {
  const int t = blockIdx.x, W = (left + right + 1) * D;
  if (!A) {
    for (int i = threadIdx.x; i < W; i += blockDim.x) {
      const int j = i / D - left, d = i % D, tt = min(max(t + j, 0), T - 1);
      out[(size_t)t * W + i] = x[(size_t)tt * D + d] - (mean ? mean[d] : 0.f);
    }
    return;
  }
  for (int m = threadIdx.x; m < M; m += blockDim.x) {
    float v = cols == W + 1 ? A[(size_t)m * cols + W] : 0.f;
    for (int i = 0; i < W; ++i) {
      const int j = i / D - left, d = i % D, tt = min(max(t + j, 0), T - 1);
      v = fmaf(A[(size_t)m * cols + i], x[(size_t)tt * D + d] - (mean ? mean[d] : 0.f), v);
    }
    out[(size_t)t * M + m] = v;
  }
}

// Kaldi LinearResample (feat/resample.cc) as used by DownsampleWaveForm (allow_downsample): windowed-sinc low-pass at
// cutoff = 0.99 * min(fs_in, fs_out) / 2 with num_zeros = 6, Hanning-windowed:
//   y[n] = sum_i x[i] * f(n / fs_out - i / fs_in) / fs_in,  f(t) = 2 c sinc(2 c t) * 0.5 (1 + cos(2 pi c t / 6)) for |t| < 6 / (2 c)
__global__ void k_resample(const float* __restrict__ x, int n_in, double fs_in, double fs_out, float* __restrict__ y, int n_out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_out) return;
  const double cutoff = 0.99 * 0.5 * fmin(fs_in, fs_out), width = 6.0 / (2.0 * cutoff), t_out = n / fs_out;
  const int i0 = max(0, (int)ceil((t_out - width) * fs_in)), i1 = min(n_in - 1, (int)floor((t_out + width) * fs_in));
  double acc = 0.0;
  for (int i = i0; i <= i1; ++i) {
    const double t = t_out - i / fs_in;
    if (fabs(t) >= width) continue;
    const double win = 0.5 * (1.0 + cos(2.0 * M_PI * cutoff / 6.0 * t));
    const double filt = t != 0.0 ? sin(2.0 * M_PI * cutoff * t) / (M_PI * t) : 2.0 * cutoff;
    acc += (double)x[i] * filt * win;
  }
  y[n] = (float)(acc / fs_in);
}

// out[t][m] = sum_k ppg[t][k] * Rt[k][m]; one workgroup per frame, M <= 64 columns x 4 k-lanes, fixed-order sum
__global__ __launch_bounds__(256) void k_reduce_ppg(const float* __restrict__ ppg, const float* __restrict__ Rt, int K, int M,
                                                    float* __restrict__ out) {
  __shared__ float red[4][64];
  const int t = blockIdx.x, m = threadIdx.x & 63, part = threadIdx.x >> 6;
  float v = 0.f;
  if (m < M)
    for (int k = part; k < K; k += 4) v = fmaf(ppg[(size_t)t * K + k], Rt[(size_t)k * M + m], v);
  red[part][m] = v;
  __syncthreads();
  if (part == 0 && m < M) out[(size_t)t * M + m] = (red[0][m] + red[1][m]) + (red[2][m] + red[3][m]);
}

}  // namespace
}  // namespace facppg

using namespace facppg;

extern "C" int facppg_mfcc_create(int frame_length, int frame_shift, int nbins, const float* basis_dev, const float* mel_dev, int n_mel,
                                  const float* dct_dev, int n_ceps, int device, void* stream_, facppg_mfcc** out) {
  FACPPG_REQUIRE(basis_dev && mel_dev && dct_dev && out, FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(frame_length > 0 && frame_shift > 0 && nbins > 0 && n_mel > 0 && n_mel <= 256 && n_ceps > 0 && n_ceps <= n_mel, FACPPG_EINVAL,
                 "bad MFCC dimensions");
  hipStream_t s = (hipStream_t)stream_;
  FACPPG_HIP_CHECK(hipSetDevice(device));
  facppg_mfcc* h = new (std::nothrow) facppg_mfcc();
  FACPPG_REQUIRE(h, FACPPG_EINVAL, "out of host memory");
  h->frame_length = frame_length; h->frame_shift = frame_shift; h->nbins = nbins; h->n_mel = n_mel; h->n_ceps = n_ceps; h->device = device;
  const size_t b_bytes = packed_a_float4s(2 * nbins, frame_length) * 16, m_bytes = (size_t)n_mel * nbins * 4, d_bytes = (size_t)n_ceps * n_mel * 4;
  const size_t total = ((b_bytes + 255) / 256 + (m_bytes + 255) / 256 + (d_bytes + 255) / 256) * 256;
  if (hipMalloc((void**)&h->arena, total) != hipSuccess) {
    set_error("hipMalloc(%zu) failed", total);
    delete h;
    return FACPPG_EHIP;
  }
  h->basis = (float4*)h->arena;
  h->mel = (float*)(h->arena + (b_bytes + 255) / 256 * 256);
  h->dct = (float*)((char*)h->mel + (m_bytes + 255) / 256 * 256);
  int rc = pack_a(basis_dev, 2 * nbins, frame_length, 1, h->basis, s);
  if (!rc && (hipMemcpyAsync(h->mel, mel_dev, m_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess ||
              hipMemcpyAsync(h->dct, dct_dev, d_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)) {
    set_error("HIP error in facppg_mfcc_create");
    rc = FACPPG_EHIP;
  }
  if (rc) {
    (void)hipFree(h->arena);
    delete h;
    return rc;
  }
  *out = h;
  return FACPPG_OK;
}

extern "C" void facppg_mfcc_destroy(facppg_mfcc* h) {
  if (!h) return;
  (void)hipFree(h->arena);
  delete h;
}

extern "C" int facppg_mfcc_num_frames(const facppg_mfcc* h, int n_samples) {
  if (!h || n_samples <= 0) return 0;
  return (n_samples + h->frame_shift / 2) / h->frame_shift;     // feature-window.cc NumFrames, snip_edges = false
}

extern "C" size_t facppg_mfcc_workspace_bytes(const facppg_mfcc* h, int n_samples) {
  const int T = facppg_mfcc_num_frames(h, n_samples);
  if (T <= 0) return 0;
  const size_t Tld = round_up(T, 64);
  return ((size_t)h->frame_length + 2 * h->nbins + 1) * Tld * 4;
}

extern "C" int facppg_mfcc_compute(facppg_mfcc* h, const float* wav_dev, int n_samples, int use_energy, float* mfcc_dev, void* ws_dev,
                                   size_t ws_bytes, void* stream_) {
  FACPPG_REQUIRE(h && wav_dev && mfcc_dev && ws_dev, FACPPG_EINVAL, "NULL argument");
  const int T = facppg_mfcc_num_frames(h, n_samples);
  FACPPG_REQUIRE(T > 0, FACPPG_EINVAL, "no frames in %d samples", n_samples);
  FACPPG_REQUIRE(ws_bytes >= facppg_mfcc_workspace_bytes(h, n_samples), FACPPG_EWORKSPACE, "workspace has %zu bytes, need %zu", ws_bytes,
                 facppg_mfcc_workspace_bytes(h, n_samples));
  hipStream_t s = (hipStream_t)stream_;
  const int Tld = round_up(T, 64);
  float* frames = (float*)ws_dev;
  float* spec = frames + (size_t)h->frame_length * Tld;
  float* energy = spec + (size_t)2 * h->nbins * Tld;
  k_mfcc_frames<<<(T + 255) / 256, 256, 0, s>>>(wav_dev, n_samples, T, h->frame_length, h->frame_shift, frames, use_energy ? energy : nullptr, Tld);
  GemmArgs g;
  g.A = h->basis; g.M = 2 * h->nbins; g.Cin = h->frame_length; g.X = frames; g.ldx = Tld; g.N = T; g.C = spec; g.ldc = Tld; g.B = 1;
  if (int rc = gemm_launch(g, s)) return rc;
  k_mfcc_tail<<<(T + 3) / 4, 256, (size_t)4 * (h->nbins + h->n_mel) * 4, s>>>(spec, Tld, T, h->nbins, h->mel, h->n_mel, h->dct, h->n_ceps,
                                                                                use_energy ? energy : nullptr, mfcc_dev);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

extern "C" int facppg_cmn_splice_transform(const float* feats_dev, int T, int D, int do_cmn, int left, int right, const float* transform_dev,
                                           int M, int cols, float* out_dev, float* mean_ws_dev, void* stream_) {
  FACPPG_REQUIRE(feats_dev && out_dev && (mean_ws_dev || !do_cmn), FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(T > 0 && D > 0 && left >= 0 && right >= 0, FACPPG_EINVAL, "bad T/D/context");
  const int W = (left + right + 1) * D;
  if (transform_dev)
    FACPPG_REQUIRE(M > 0 && (cols == W || cols == W + 1), FACPPG_EINVAL, "Transform matrix has bad dimension %dx%d versus feat dim %d", M, cols, W);
  hipStream_t s = (hipStream_t)stream_;
  if (do_cmn) k_col_mean<<<1, 256, 0, s>>>(feats_dev, T, D, mean_ws_dev);
  k_cmn_splice_transform<<<T, 64, 0, s>>>(feats_dev, T, D, do_cmn ? mean_ws_dev : nullptr, left, right, transform_dev, M, cols, out_dev);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

extern "C" int facppg_reduce_ppg(const float* ppg_dev, const float* transform_t_dev, int T, int K, int M, float* out_dev, void* stream_) {
  FACPPG_REQUIRE(ppg_dev && transform_t_dev && out_dev, FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(T > 0 && K > 0 && M > 0 && M <= 64, FACPPG_EINVAL, "bad T/K/M (M <= 64)");
  k_reduce_ppg<<<T, 256, 0, (hipStream_t)stream_>>>(ppg_dev, transform_t_dev, K, M, out_dev);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

extern "C" int facppg_resample_num_samples(int n_in, int fs_in, int fs_out) {
  if (n_in <= 0 || fs_in <= 0 || fs_out <= 0) return 0;
  // LinearResample::GetNumOutputSamples with flush = true: outputs strictly inside the input's time span
  const long long num = (long long)n_in * fs_out;
  long long last = num / fs_in;
  if (last * fs_in == num) --last;
  return (int)(last + 1);
}

extern "C" int facppg_resample(const float* wav_dev, int n_in, int fs_in, int fs_out, float* out_dev, void* stream_) {
  FACPPG_REQUIRE(wav_dev && out_dev && n_in > 0 && fs_in > 0 && fs_out > 0, FACPPG_EINVAL, "bad argument");
  const int n_out = facppg_resample_num_samples(n_in, fs_in, fs_out);
  k_resample<<<(n_out + 255) / 256, 256, 0, (hipStream_t)stream_>>>(wav_dev, n_in, (double)fs_in, (double)fs_out, out_dev, n_out);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}
