/* facppg.h -- C ABI of libfacppg_hip.so: the MI355X (gfx950) implementation of the
 * PPG -> mel -> waveform synthesis hot path of guanlongzhao/fac-via-ppg.
 *
 * The reference has no FFI/plugin layer (it is 100 % Python on torch ops); its boundary is
 * the Python surface listed in SURVEY.md section 8b.  This header is the C ABI that sits
 * directly beneath that surface: every entry point names the reference function it
 * replaces (file:line under /root/reference).  The Python modules in fac-via-ppg_amd/
 * (common.*, waveglow.*, script.*) keep the reference's names and call these through ctypes.
 *
 * Conventions
 *   - return 0 on success, a negative FACPPG_E* code otherwise; nothing throws across the ABI;
 *     facppg_last_error() gives a thread-local message for the last failure.
 *   - every pointer named *_dev is DEVICE memory owned by the caller (e.g. a torch tensor);
 *     a handle owns only its packed weights.  All work is enqueued on the caller's
 *     hipStream_t (passed as void*); no entry point synchronises unless it says so.
 *   - a handle is bound to one device and is not thread-safe; handles on different devices
 *     are independent (one process per GPU).
 *   - all tensors are fp32, dense, row-major in the layouts given per function.
 */
#ifndef FACPPG_H
#define FACPPG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FACPPG_VERSION 100 /* 0.1.0 */

#define FACPPG_OK 0
#define FACPPG_EINVAL (-1)       /* bad argument (NULL pointer, non-positive size, ...) */
#define FACPPG_EUNSUPPORTED (-2) /* configuration outside what the kernels are built for */
#define FACPPG_EHIP (-3)         /* a HIP runtime call failed */
#define FACPPG_EWORKSPACE (-4)   /* workspace smaller than facppg_*_workspace_bytes() */

int facppg_version(void);
/* Thread-local, valid until the next failing call on this thread. */
const char* facppg_last_error(void);

/* ------------------------------------------------------------------------------------
 * WaveGlow vocoder (src/waveglow/glow.py)
 * ---------------------------------------------------------------------------------- */

/* Constructor arguments of WaveGlow.__init__ (glow.py:179-206) / config.json:29-41. */
typedef struct facppg_wg_config {
  int32_t n_mel_channels; /* 80 */
  int32_t hop_length;     /* 160 (reference) or 256; must be a multiple of n_group */
  int32_t n_flows;        /* 12 */
  int32_t n_group;        /* 8 */
  int32_t n_early_every;  /* 4 */
  int32_t n_early_size;   /* 2 */
  int32_t wn_layers;      /* 8  (WN_config.n_layers)   */
  int32_t wn_channels;    /* 256 (WN_config.n_channels) */
  int32_t wn_kernel_size; /* 3  (WN_config.kernel_size) */
  int32_t upsample_kernel; /* 1024: ConvTranspose1d kernel size, glow.py:184-186 */
} facppg_wg_config;

typedef struct facppg_wg facppg_wg;

/* Number of fp32 values in the plain weight blob facppg_wg_create() expects.  The blob is
 * the post-remove_weightnorm state dict (glow.py:295-311) flattened
 * in THIS order, each tensor dense row-major:
 *   upsample.weight [n_mel][n_mel][upsample_kernel], upsample.bias [n_mel]
 *   for k in 0..n_flows-1   (h_k = n_half of flow k, c_k = 2*h_k):
 *     WN.k.start.weight [C][h_k], WN.k.start.bias [C]
 *     for i in 0..wn_layers-1:
 *       WN.k.in_layers.i.weight [2C][C][ks], .bias [2C]
 *       WN.k.cond_layers.i.weight [2C][n_mel*n_group], .bias [2C]
 *       WN.k.res_skip_layers.i.weight [2C or C (last i)][C], .bias
 *     WN.k.end.weight [c_k][C], WN.k.end.bias [c_k]
 *     convinv.k  W_inverse [c_k][c_k]   (= conv.weight.squeeze().inverse(), glow.py:88-95)
 */
size_t facppg_wg_weight_count(const facppg_wg_config* cfg);

/* Replaces: WaveGlow.__init__ + load_waveglow_model's weight preparation
 * (glow.py:179-206, common/utils.py:177-181).  Packs the plain blob into the MFMA operand
 * layouts on `device` using `stream`; synchronises that stream before returning, so the
 * caller may free `weights_dev` afterwards. */
int facppg_wg_create(const facppg_wg_config* cfg, const float* weights_dev, size_t n_floats,
                     int device, void* stream, facppg_wg** out);
void facppg_wg_destroy(facppg_wg* h);

/* Scratch bytes facppg_wg_infer needs for a batch of B mels of (max) T frames. */
size_t facppg_wg_workspace_bytes(const facppg_wg* h, int B, int T);

/* Replaces: WaveGlow.infer(spect, sigma) (glow.py:252-293), i.e. upsample + regroup, the
 * 12 x (WN, affine-coupling inverse, inverse 1x1 conv) flows, early-z concatenation and the
 * final group->time interleave.
 *   mel_dev   [B][n_mel][T]
 *   T_valid_dev  NULL, or [B] int32 frame counts <= T: utterance b is synthesised exactly as
 *             a batch-1 call on mel[b, :, :T_valid[b]] would be (frames beyond it are ignored,
 *             audio beyond T_valid[b]*hop is left untouched).
 *   z_dev     NULL -> noise is generated on the device from `seed` (Philox4x32-10 + Box-Muller);
 *             else the reference's three normal_() draws in call order, concatenated flat:
 *             [B][n_remaining][L] ++ [B][n_early_size][L] (flow 8) ++ [B][n_early_size][L] (flow 4),
 *             L = T*hop/n_group  (glow.py:261-270, 285-290).
 *   audio_dev [B][T*hop]
 */
int facppg_wg_infer(facppg_wg* h, const float* mel_dev, const int32_t* T_valid_dev,
                    const float* z_dev, uint64_t seed, float sigma, int B, int T,
                    float* audio_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* Average device time (ms) of the dominant kernel (the fused WN layer) over the launches of
 * the most recent facppg_wg_infer on this handle, measured with hipEvents on the stream the
 * kernels ran on when profiling was enabled with facppg_wg_set_profiling(h, 1).  Synchronises
 * the recorded events.  *n_launches receives the number of launches averaged. */
int facppg_wg_set_profiling(facppg_wg* h, int enable);
int facppg_wg_last_layer_ms(facppg_wg* h, float* avg_ms, int* n_launches);

#ifdef __cplusplus
}
#endif
#endif /* FACPPG_H */
