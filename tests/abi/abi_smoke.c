/* Plain-C client of libfacppg_hip.so: proves include/facppg.h is a self-contained C header and that the
 * library can be driven without Python or torch (device memory through the HIP runtime C API only).
 * Runs WaveGlow.infer twice on random weights with the same seed and checks: error codes, determinism,
 * finite output, the EINVAL / EWORKSPACE paths, and the seeded path of ONE utterance (facppg_wg_mel_pad -> facppg_wg_cond_seed ->
 * facppg_wg_infer_seeded: all-seeded and with an unseeded tail) against facppg_wg_infer, bit for bit.
 * Built and run by tests/test_gpu_abi_c.py. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "facppg.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_RC(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, facppg_last_error()); return 3; } } while (0)

int main(void) {
  facppg_wg_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.n_mel_channels = 80; cfg.hop_length = 256; cfg.n_flows = 4; cfg.n_group = 8; cfg.n_early_every = 2; cfg.n_early_size = 2;
  cfg.wn_layers = 8; cfg.wn_channels = 256; cfg.wn_kernel_size = 3; cfg.upsample_kernel = 1024; cfg.alternate_halves = 0;
  const size_t nw = facppg_wg_weight_count(&cfg);
  if (nw == 0) { fprintf(stderr, "weight_count: %s\n", facppg_last_error()); return 1; }
  float* hw = (float*)malloc(nw * sizeof(float));
  unsigned s = 12345u;
  for (size_t i = 0; i < nw; ++i) {   /* small random weights: keeps the flow well conditioned */
    s = s * 1664525u + 1013904223u;
    hw[i] = ((float)(s >> 8) / 16777216.0f - 0.5f) * 0.02f;
  }
  /* the blob ends each flow with W_inverse then W (include/facppg.h): make those identities */
  {
    size_t off = (size_t)80 * 80 * 1024 + 80;
    int n_half = 4, n_rem = 8;
    for (int k = 0; k < cfg.n_flows; ++k) {
      if (k % cfg.n_early_every == 0 && k > 0) { n_half -= 1; n_rem -= 2; }
      const size_t cc = 2 * (size_t)n_half;
      off += (size_t)256 * n_half + 256;
      for (int i = 0; i < cfg.wn_layers; ++i) {
        const size_t rs = i < cfg.wn_layers - 1 ? 512 : 256;
        off += (size_t)512 * 256 * 3 + 512 + (size_t)512 * 640 + 512 + rs * 256 + rs;
      }
      off += cc * 256 + cc;
      for (int m = 0; m < 2; ++m, off += cc * cc)
        for (size_t i = 0; i < cc; ++i)
          for (size_t j = 0; j < cc; ++j) hw[off + i * cc + j] = i == j ? 1.0f : 0.0f;
    }
    (void)n_rem;
    if (off != nw) { fprintf(stderr, "blob walk %zu != %zu\n", off, nw); return 1; }
  }
  float* dw = NULL;
  CHECK_HIP(hipMalloc((void**)&dw, nw * sizeof(float)));
  CHECK_HIP(hipMemcpy(dw, hw, nw * sizeof(float), hipMemcpyHostToDevice));
  facppg_wg* h = NULL;
  CHECK_RC(facppg_wg_create(&cfg, dw, nw, 0, NULL, &h));

  const int B = 2, T = 24;
  const size_t n_mel = (size_t)B * 80 * T, n_audio = (size_t)B * T * cfg.hop_length;
  float* hmel = (float*)malloc(n_mel * sizeof(float));
  for (size_t i = 0; i < n_mel; ++i) { s = s * 1664525u + 1013904223u; hmel[i] = -5.0f + 2.0f * ((float)(s >> 8) / 16777216.0f - 0.5f); }
  float *dmel = NULL, *daud = NULL;
  void* ws = NULL;
  const size_t wsb = facppg_wg_workspace_bytes(h, B, T);
  CHECK_HIP(hipMalloc((void**)&dmel, n_mel * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&daud, n_audio * sizeof(float)));
  CHECK_HIP(hipMalloc(&ws, wsb));
  CHECK_HIP(hipMemcpy(dmel, hmel, n_mel * sizeof(float), hipMemcpyHostToDevice));
  float* a0 = (float*)malloc(n_audio * sizeof(float));
  float* a1 = (float*)malloc(n_audio * sizeof(float));
  for (int rep = 0; rep < 2; ++rep) {
    CHECK_RC(facppg_wg_infer(h, dmel, NULL, NULL, 777u, 0.6f, B, T, daud, ws, wsb, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(rep ? a1 : a0, daud, n_audio * sizeof(float), hipMemcpyDeviceToHost));
  }
  double rms = 0.0;
  for (size_t i = 0; i < n_audio; ++i) {
    if (!isfinite(a0[i])) { fprintf(stderr, "non-finite sample %zu\n", i); return 4; }
    if (a0[i] != a1[i]) { fprintf(stderr, "run-to-run mismatch at %zu\n", i); return 4; }
    rms += (double)a0[i] * a0[i];
  }
  rms = sqrt(rms / (double)n_audio);
  if (!(rms > 1e-3 && rms < 1e3)) { fprintf(stderr, "implausible rms %g\n", rms); return 4; }
  /* error paths */
  if (facppg_wg_infer(h, dmel, NULL, NULL, 1u, 0.6f, B, T, daud, ws, wsb / 2, NULL) != FACPPG_EWORKSPACE) { fprintf(stderr, "expected EWORKSPACE\n"); return 5; }
  if (facppg_wg_infer(h, NULL, NULL, NULL, 1u, 0.6f, B, T, daud, ws, wsb, NULL) != FACPPG_EINVAL) { fprintf(stderr, "expected EINVAL\n"); return 5; }
  /* a stand-alone entry point with plain structs: sums of strided segments (WaveGlowLoss's reductions) */
  {
    const int n = 1000;
    float* hv = (float*)malloc(2 * n * sizeof(float));
    double want0 = 0.0, want1 = 0.0;
    for (int i = 0; i < 2 * n; ++i) hv[i] = (float)((i * 37) % 11) - 5.0f;
    for (int i = 0; i < n; ++i) { want0 += (double)hv[i] * hv[i]; }
    for (int o = 0; o < 4; ++o) for (int i = 0; i < 100; ++i) want1 += hv[n + o * 250 + i];
    float *dv = NULL, *dout = NULL;
    void* sws = NULL;
    CHECK_HIP(hipMalloc((void**)&dv, 2 * n * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&dout, 2 * sizeof(float)));
    CHECK_HIP(hipMalloc(&sws, 2 * 64 * sizeof(double)));
    CHECK_HIP(hipMemcpy(dv, hv, 2 * n * sizeof(float), hipMemcpyHostToDevice));
    facppg_sum_segment segs[2] = {{dv, 0, 1, n, 1}, {dv + n, 250, 4, 100, 0}};
    CHECK_RC(facppg_segment_sums(segs, 2, sws, 2 * 64 * sizeof(double), dout, NULL));
    float got[2];
    CHECK_HIP(hipMemcpy(got, dout, sizeof(got), hipMemcpyDeviceToHost));
    if (fabs(got[0] - want0) > 1e-3 * fabs(want0) || fabs(got[1] - want1) > 1e-3 + 1e-5 * fabs(want1)) {
      fprintf(stderr, "segment sums %g %g, expected %g %g\n", got[0], got[1], want0, want1);
      return 6;
    }
    if (facppg_segment_sums(segs, 2, sws, 8, dout, NULL) != FACPPG_EINVAL) { fprintf(stderr, "expected EINVAL for a short workspace\n"); return 6; }
    free(hv);
  }
  /* one utterance through the seeded entry points: the same samples as facppg_wg_infer, bit for bit */
  {
    const int T1 = 72;
    const size_t n1 = (size_t)T1 * cfg.hop_length;
    int tqp = 0, margin = 0;
    size_t seed_bytes = 0;
    CHECK_RC(facppg_wg_seed_layout(h, T1, &tqp, &margin, &seed_bytes));
    float *dmel1 = NULL, *dmelp = NULL, *dseeds = NULL, *daud1 = NULL;
    void* ws1 = NULL;
    const size_t wsb1 = facppg_wg_workspace_bytes(h, 1, T1);
    CHECK_HIP(hipMalloc((void**)&dmel1, (size_t)80 * T1 * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&dmelp, (size_t)80 * tqp * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&dseeds, seed_bytes));
    CHECK_HIP(hipMalloc((void**)&daud1, n1 * sizeof(float)));
    CHECK_HIP(hipMalloc(&ws1, wsb1));
    float* hm1 = (float*)malloc((size_t)80 * T1 * sizeof(float));
    for (size_t i = 0; i < (size_t)80 * T1; ++i) { s = s * 1664525u + 1013904223u; hm1[i] = -5.0f + 2.0f * ((float)(s >> 8) / 16777216.0f - 0.5f); }
    CHECK_HIP(hipMemcpy(dmel1, hm1, (size_t)80 * T1 * sizeof(float), hipMemcpyHostToDevice));
    float* r0 = (float*)malloc(n1 * sizeof(float));
    float* r1 = (float*)malloc(n1 * sizeof(float));
    CHECK_RC(facppg_wg_infer(h, dmel1, NULL, NULL, 4242u, 0.6f, 1, T1, daud1, ws1, wsb1, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(r0, daud1, n1 * sizeof(float), hipMemcpyDeviceToHost));
    CHECK_RC(facppg_wg_mel_pad(h, dmel1, T1, T1, dmelp, NULL));
    for (int mode = 0; mode < 2; ++mode) {   /* 0: every frame seeded (96 >= 72); 1: 64 seeded frames + an unseeded tail */
      const int seeded = mode ? 64 : 96;
      CHECK_HIP(hipMemset(dseeds, 0xff, seed_bytes));
      CHECK_RC(facppg_wg_cond_seed(h, dmelp, T1, 0, seeded, 1, 2, 0, 0, dseeds, seed_bytes, NULL, 0, NULL, NULL));
      CHECK_RC(facppg_wg_infer_seeded(h, dmelp, T1, T1, dseeds, seeded, NULL, 4242u, 0.6f, daud1, ws1, wsb1, NULL, NULL));
      CHECK_HIP(hipDeviceSynchronize());
      CHECK_HIP(hipMemcpy(r1, daud1, n1 * sizeof(float), hipMemcpyDeviceToHost));
      for (size_t i = 0; i < n1; ++i)
        if (r0[i] != r1[i]) { fprintf(stderr, "seeded path (mode %d) differs from facppg_wg_infer at sample %zu: %g vs %g\n", mode, i, r1[i], r0[i]); return 7; }
    }
    if (facppg_wg_cond_seed(h, dmelp, T1, 16, 32, 1, 2, 0, 0, dseeds, seed_bytes, NULL, 0, NULL, NULL) != FACPPG_EINVAL) { fprintf(stderr, "expected EINVAL for a block that does not start on a tile\n"); return 7; }
    free(hm1); free(r0); free(r1);
  }
  facppg_wg_destroy(h);
  printf("abi_smoke ok: version %d, %zu weights, %zu samples, rms %.4f, workspace %zu bytes\n", facppg_version(), nw, n_audio, rms, wsb);
  return 0;
}
