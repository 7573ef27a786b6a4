// STFT / inverse STFT / mel analysis / WaveGlow denoiser on gfx950.
//
// Replaces src/common/stft.py:79-138 (STFT.transform / inverse as conv1d / conv_transpose1d
// with a windowed DFT basis), src/common/layers.py:96-112 (TacotronSTFT.mel_spectrogram),
// src/common/audio_processing.py:39-88,110-116 (window_sumsquare, log compression) and
// src/waveglow/denoiser.py:63-68 (Denoiser.forward).
//
// Structure: the strided framing conv is turned into a dense GEMM by first materialising the
// frame matrix X[1024][F] (reflect padding applied while gathering), so the DFT, the mel
// filterbank and the inverse DFT are all exact-fp32 MFMA GEMMs (facppg_gemm); the inverse
// transform's overlap-add, the window-sum-square normalisation (recomputed on the HOST with
// NumPy on every call in the reference, stft.py:119-130) and the trim are one gather kernel.
#include <cstring>
#include <new>

#include "facppg_gemm.h"

using namespace facppg;

struct facppg_stft {
  int fl, hop, cutoff, n_mel, device;
  char* arena;
  float4 *fwd, *inv_t, *melb;  // packed A operands
  float* win_sq;               // [fl]
};

namespace {

// frames: X[b][k][f] = xpad[f*hop + k], xpad = reflect-pad(x, fl/2)   (stft.py:86-97)
__global__ void k_frames(const float* __restrict__ x, float* __restrict__ X, const int* __restrict__ n_valid, int N, int fl,
                         int hop, int F, int* __restrict__ f_valid) {
  const int b = blockIdx.z, k = blockIdx.y;
  const int Nb = n_valid ? n_valid[b] : N;
  const int Fb = Nb / hop + 1;
  if (f_valid && blockIdx.x == 0 && k == 0 && threadIdx.x == 0) f_valid[b] = Fb;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= Fb) return;
  int j = f * hop + k - fl / 2;
  if (j < 0) j = -j;
  if (j >= Nb) j = 2 * (Nb - 1) - j;
  X[((size_t)b * fl + k) * F + f] = x[(size_t)b * N + j];
}

// S[b][2*cutoff][F] (re rows then im rows) -> magnitude, phase   (stft.py:99-105)
__global__ void k_magphase(const float* __restrict__ S, float* __restrict__ mag, float* __restrict__ phase,
                           const int* __restrict__ f_valid, int cutoff, int F) {
  const int b = blockIdx.z, c = blockIdx.y, f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= (f_valid ? f_valid[b] : F)) return;
  const float re = S[((size_t)b * 2 * cutoff + c) * F + f], im = S[((size_t)b * 2 * cutoff + cutoff + c) * F + f];
  const size_t o = ((size_t)b * cutoff + c) * F + f;
  mag[o] = sqrtf(re * re + im * im);
  if (phase) phase[o] = atan2f(im, re);
}

// R = [mag*cos(phase); mag*sin(phase)]   (stft.py:110-111)
__global__ void k_recombine(const float* __restrict__ mag, const float* __restrict__ phase, float* __restrict__ R,
                            const int* __restrict__ f_valid, int cutoff, int F) {
  const int b = blockIdx.z, c = blockIdx.y, f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= (f_valid ? f_valid[b] : F)) return;
  const size_t o = ((size_t)b * cutoff + c) * F + f;
  float sn, cs;
  sincosf(phase[o], &sn, &cs);
  R[((size_t)b * 2 * cutoff + c) * F + f] = mag[o] * cs;
  R[((size_t)b * 2 * cutoff + cutoff + c) * F + f] = mag[o] * sn;
}

// Denoiser core (denoiser.py:64-67): mag' = max(mag - bias*strength, 0), same phase.  With
// phase = atan2(im, re): mag'*cos(phase) = mag' * re/mag; atan2(0,0) = 0 -> (mag', 0).
__global__ void k_spectral_subtract(float* __restrict__ S, const float* __restrict__ bias, float strength,
                                    const int* __restrict__ f_valid, int cutoff, int F) {
  const int b = blockIdx.z, c = blockIdx.y, f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= (f_valid ? f_valid[b] : F)) return;
  const size_t ore = ((size_t)b * 2 * cutoff + c) * F + f, oim = ore + (size_t)cutoff * F;
  const float re = S[ore], im = S[oim];
  const float mag = sqrtf(re * re + im * im);
  const float m2 = fmaxf(mag - bias[c] * strength, 0.0f);
  if (mag > 0.0f) {
    const float sc = m2 / mag;
    S[ore] = re * sc; S[oim] = im * sc;
  } else {
    S[ore] = m2; S[oim] = 0.0f;
  }
}

// Overlap-add of Zt[b][f][k] (conv_transpose1d, stft.py:113-117), window-sum-square normalisation
// where > tiny, * fl/hop, trim fl/2 both sides (stft.py:119-136; audio_processing.py:76-88).
__global__ void k_overlap_add(const float* __restrict__ Zt, const float* __restrict__ win_sq, float* __restrict__ y,
                              const int* __restrict__ f_valid, int fl, int hop, int F, long y_bs) {
  const int b = blockIdx.y;
  const int Fb = f_valid ? f_valid[b] : F;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // output index, n = i + fl/2
  if (i >= hop * (Fb - 1)) return;
  const int n = i + fl / 2;
  int f_hi = n / hop;
  if (f_hi > Fb - 1) f_hi = Fb - 1;
  int f_lo = (n - fl + hop) / hop;  // ceil((n - fl + 1)/hop)
  if (f_lo < 0) f_lo = 0;
  float acc = 0.0f, ws = 0.0f;
  for (int f = f_lo; f <= f_hi; ++f) {
    const int k = n - f * hop;
    acc += Zt[((size_t)b * F + f) * fl + k];
    ws += win_sq[k];
  }
  if (ws > 1.17549435e-38f) acc /= ws;
  y[(size_t)b * y_bs + i] = acc * ((float)fl / (float)hop);
}

struct Ws {
  int F;
  size_t X, S, Z, fv, sk, sk_bytes, total;
};
Ws ws_layout(const facppg_stft* h, int B, int N) {
  Ws w;
  w.F = N / h->hop + 1;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  w.X = take((size_t)B * (h->fl + 2) * w.F * 4);   // frames, reused as R
  w.S = take((size_t)B * (h->fl + 2) * w.F * 4);
  w.Z = take((size_t)B * w.F * h->fl * 4);
  w.fv = take((size_t)B * 4);
  // split-K partial sums of the two DFT products (K = filter_length resp. 2 * cutoff: gemm_launch splits by K alone, so an
  // utterance is summed in the same order in any batch): a short utterance's [1026 x 1024] x [1024 x 201] product is 36
  // workgroups with a 16-chunk serial K loop otherwise -- 70 us for 3 us of MFMA work
  w.sk_bytes = (size_t)4 * B * (h->fl + 2) * w.F * 4;
  w.sk = take(w.sk_bytes);
  w.total = off;
  return w;
}

int run_forward(facppg_stft* h, const float* audio, const int* n_valid, int B, int N, char* ws, const Ws& w, hipStream_t s) {
  float* X = (float*)(ws + w.X);
  float* S = (float*)(ws + w.S);
  int* fv = (int*)(ws + w.fv);
  dim3 g((w.F + 255) / 256, h->fl, B);
  k_frames<<<g, 256, 0, s>>>(audio, X, n_valid, N, h->fl, h->hop, w.F, fv);
  GemmArgs a;
  a.A = h->fwd; a.M = 2 * h->cutoff; a.Cin = h->fl; a.X = X; a.x_bs = (long)h->fl * w.F; a.ldx = w.F; a.N = w.F;
  a.n_valid = fv; a.C = S; a.c_bs = (long)2 * h->cutoff * w.F; a.ldc = w.F; a.B = B;
  a.splitk_ws = (float*)(ws + w.sk); a.splitk_ws_bytes = w.sk_bytes;
  return gemm_launch(a, s);
}

int run_inverse(facppg_stft* h, const float* R, int B, char* ws, const Ws& w, const int* fv, float* out, long out_bs, hipStream_t s) {
  float* Z = (float*)(ws + w.Z);
  GemmArgs a;
  a.A = h->inv_t; a.M = h->fl; a.Cin = 2 * h->cutoff; a.X = R; a.x_bs = (long)2 * h->cutoff * w.F; a.ldx = w.F; a.N = w.F;
  a.n_valid = fv; a.C = Z; a.c_bs = (long)w.F * h->fl; a.ldc = h->fl; a.c_transposed = 1; a.B = B;
  a.splitk_ws = (float*)(ws + w.sk); a.splitk_ws_bytes = w.sk_bytes;
  if (int rc = gemm_launch(a, s)) return rc;
  const int n_out = h->hop * (w.F - 1);
  if (n_out > 0) {
    dim3 g((n_out + 255) / 256, B);
    k_overlap_add<<<g, 256, 0, s>>>(Z, h->win_sq, out, fv, h->fl, h->hop, w.F, out_bs);
  }
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

int check_common(const facppg_stft* h, const void* a, const void* b, const void* ws, int B, int N, size_t ws_bytes, Ws* w) {
  FACPPG_REQUIRE(h && a && b && ws, FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(B > 0 && N > h->fl / 2, FACPPG_EINVAL, "need B > 0 and N > filter_length/2 (reflect padding), got B=%d N=%d", B, N);
  *w = ws_layout(h, B, N);
  FACPPG_REQUIRE(ws_bytes >= w->total, FACPPG_EWORKSPACE, "workspace has %zu bytes, need %zu", ws_bytes, w->total);
  return FACPPG_OK;
}

}  // namespace

extern "C" int facppg_stft_create(int filter_length, int hop_length, const float* fwd_basis_dev, const float* inv_basis_t_dev,
                                  const float* win_sq_dev, const float* mel_basis_dev, int n_mel, int device, void* stream_,
                                  facppg_stft** out) {
  FACPPG_REQUIRE(fwd_basis_dev && inv_basis_t_dev && win_sq_dev && out, FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(filter_length >= 16 && filter_length % 2 == 0 && hop_length > 0 && hop_length <= filter_length, FACPPG_EINVAL,
                 "bad filter_length/hop_length %d/%d", filter_length, hop_length);
  FACPPG_REQUIRE(!mel_basis_dev || n_mel > 0, FACPPG_EINVAL, "n_mel must be positive with a mel basis");
  hipStream_t s = (hipStream_t)stream_;
  FACPPG_HIP_CHECK(hipSetDevice(device));
  facppg_stft* h = new (std::nothrow) facppg_stft();
  FACPPG_REQUIRE(h, FACPPG_EINVAL, "out of host memory");
  h->fl = filter_length; h->hop = hop_length; h->cutoff = filter_length / 2 + 1; h->n_mel = mel_basis_dev ? n_mel : 0;
  h->device = device;
  const int R = 2 * h->cutoff;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t o_fwd = take(packed_a_float4s(R, h->fl) * 16), o_inv = take(packed_a_float4s(h->fl, R) * 16);
  const size_t o_mel = take(h->n_mel ? packed_a_float4s(h->n_mel, h->cutoff) * 16 : 16), o_win = take((size_t)h->fl * 4);
  if (hipMalloc((void**)&h->arena, off) != hipSuccess) {
    set_error("hipMalloc(%zu) failed", off);
    delete h;
    return FACPPG_EHIP;
  }
  h->fwd = (float4*)(h->arena + o_fwd); h->inv_t = (float4*)(h->arena + o_inv); h->melb = (float4*)(h->arena + o_mel);
  h->win_sq = (float*)(h->arena + o_win);
  int rc = pack_a(fwd_basis_dev, R, h->fl, 1, h->fwd, s);
  if (!rc) rc = pack_a(inv_basis_t_dev, h->fl, R, 1, h->inv_t, s);
  if (!rc && h->n_mel) rc = pack_a(mel_basis_dev, h->n_mel, h->cutoff, 1, h->melb, s);
  if (!rc && hipMemcpyAsync(h->win_sq, win_sq_dev, (size_t)h->fl * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) rc = FACPPG_EHIP;
  if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = FACPPG_EHIP;
  if (rc) {
    (void)hipFree(h->arena);
    delete h;
    return rc;
  }
  *out = h;
  return FACPPG_OK;
}

extern "C" void facppg_stft_destroy(facppg_stft* h) {
  if (!h) return;
  (void)hipFree(h->arena);
  delete h;
}

extern "C" size_t facppg_stft_workspace_bytes(const facppg_stft* h, int B, int N) {
  if (!h || B <= 0 || N <= 0) return 0;
  return ws_layout(h, B, N).total;
}

extern "C" int facppg_stft_transform(facppg_stft* h, const float* audio_dev, const int32_t* n_valid_dev, int B, int N,
                                     float* mag_dev, float* phase_dev, void* ws_, size_t ws_bytes, void* stream_) {
  Ws w;
  if (int rc = check_common(h, audio_dev, mag_dev, ws_, B, N, ws_bytes, &w)) return rc;
  hipStream_t s = (hipStream_t)stream_;
  char* ws = (char*)ws_;
  if (int rc = run_forward(h, audio_dev, n_valid_dev, B, N, ws, w, s)) return rc;
  dim3 g((w.F + 255) / 256, h->cutoff, B);
  k_magphase<<<g, 256, 0, s>>>((float*)(ws + w.S), mag_dev, phase_dev, (int*)(ws + w.fv), h->cutoff, w.F);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

extern "C" int facppg_stft_inverse(facppg_stft* h, const float* mag_dev, const float* phase_dev, int B, int F, float* out_dev,
                                   void* ws_, size_t ws_bytes, void* stream_) {
  FACPPG_REQUIRE(h && mag_dev && phase_dev && out_dev && ws_, FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(B > 0 && F > 1, FACPPG_EINVAL, "need B > 0 and at least 2 frames");
  const int N = (F - 1) * h->hop;
  const Ws w = ws_layout(h, B, N);
  FACPPG_REQUIRE(ws_bytes >= w.total, FACPPG_EWORKSPACE, "workspace has %zu bytes, need %zu", ws_bytes, w.total);
  hipStream_t s = (hipStream_t)stream_;
  char* ws = (char*)ws_;
  float* R = (float*)(ws + w.X);
  dim3 g((F + 255) / 256, h->cutoff, B);
  k_recombine<<<g, 256, 0, s>>>(mag_dev, phase_dev, R, nullptr, h->cutoff, F);
  return run_inverse(h, R, B, ws, w, nullptr, out_dev, (long)h->hop * (F - 1), s);
}

extern "C" int facppg_stft_mel(facppg_stft* h, const float* audio_dev, const int32_t* n_valid_dev, int B, int N, float* mel_dev,
                               void* ws_, size_t ws_bytes, void* stream_) {
  Ws w;
  if (int rc = check_common(h, audio_dev, mel_dev, ws_, B, N, ws_bytes, &w)) return rc;
  FACPPG_REQUIRE(h->n_mel > 0, FACPPG_EINVAL, "handle was created without a mel basis");
  hipStream_t s = (hipStream_t)stream_;
  char* ws = (char*)ws_;
  if (int rc = run_forward(h, audio_dev, n_valid_dev, B, N, ws, w, s)) return rc;
  float* mag = (float*)(ws + w.X);  // frames no longer needed
  int* fv = (int*)(ws + w.fv);
  dim3 g((w.F + 255) / 256, h->cutoff, B);
  k_magphase<<<g, 256, 0, s>>>((float*)(ws + w.S), mag, nullptr, fv, h->cutoff, w.F);
  GemmArgs a;
  a.A = h->melb; a.M = h->n_mel; a.Cin = h->cutoff; a.X = mag; a.x_bs = (long)h->cutoff * w.F; a.ldx = w.F; a.N = w.F;
  a.n_valid = fv; a.act = ACT_LOG_CLAMP; a.C = mel_dev; a.c_bs = (long)h->n_mel * w.F; a.ldc = w.F; a.B = B;
  return gemm_launch(a, s);
}

extern "C" int facppg_denoise(facppg_stft* h, const float* audio_dev, const int32_t* n_valid_dev, const float* bias_spec_dev,
                              float strength, int B, int N, float* out_dev, void* ws_, size_t ws_bytes, void* stream_) {
  Ws w;
  if (int rc = check_common(h, audio_dev, out_dev, ws_, B, N, ws_bytes, &w)) return rc;
  FACPPG_REQUIRE(bias_spec_dev, FACPPG_EINVAL, "bias_spec is NULL");
  hipStream_t s = (hipStream_t)stream_;
  char* ws = (char*)ws_;
  if (int rc = run_forward(h, audio_dev, n_valid_dev, B, N, ws, w, s)) return rc;
  float* S = (float*)(ws + w.S);
  int* fv = (int*)(ws + w.fv);
  dim3 g((w.F + 255) / 256, h->cutoff, B);
  k_spectral_subtract<<<g, 256, 0, s>>>(S, bias_spec_dev, strength, fv, h->cutoff, w.F);
  return run_inverse(h, S, B, ws, w, fv, out_dev, (long)h->hop * (w.F - 1), s);
}
