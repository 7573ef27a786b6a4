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

#define FACPPG_VERSION 103 /* 0.1.2 */

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
  int32_t alternate_halves; /* 0; 1 = legacy glow_old.py layout: odd flows condition on the second half */
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
 *     convinv.k  W [c_k][c_k]           (conv.weight.squeeze(), used by the training direction)
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

/* The three normal_() draws of WaveGlow.infer (glow.py:261-270, 285-290) as PER-UTTERANCE streams:
 * writes z_dev in the flat injected-z layout of facppg_wg_infer for a batch of B mels of T frames, where
 * every value of utterance b is a function of seeds_dev[b] (uint64 [B]) and its own (draw, channel,
 * position) only.  A padded batch synthesised with this z therefore equals B independent batch-1 runs
 * with the same seeds, whatever the batch composition, padding or sharding over GPUs. */
int facppg_wg_draw_noise(const facppg_wg* h, const uint64_t* seeds_dev, int B, int T, float* z_dev,
                         void* stream);

/* Replaces: WaveGlow.forward((spect, audio)) (glow.py:208-250), the training direction
 * audio -> z: upsample + crop to the audio length, 8-sample regroup, per flow the forward 1x1
 * mixing conv, WN, and a1 = exp(log_s)*a1 + b, with early outputs split off every n_early_every
 * flows.
 *   mel_dev [B][n_mel][F], audio_dev [B][N] (N a multiple of n_group; (F-1)*hop + kernel >= N)
 *   z_dev   [B][n_group][N/n_group]   = cat(early outputs..., final) along channels (glow.py:249)
 *   log_s_dev  the per-flow log_s tensors [B][h_k][N/n_group], flow 0 first, concatenated flat
 *              (facppg_wg_log_s_count values); log|det W_k| is a property of the weights alone and
 *              is left to the caller (glow.py:100).
 * Workspace: facppg_wg_workspace_bytes(h, B, ceil(N / hop)). */
size_t facppg_wg_log_s_count(const facppg_wg* h, int B, int N);
int facppg_wg_forward(facppg_wg* h, const float* mel_dev, const float* audio_dev, int B, int F,
                      int N, float* z_dev, float* log_s_dev, void* workspace_dev,
                      size_t workspace_bytes, void* stream);

/* Replaces Invertible1x1Conv.forward (glow.py:82-102) as a stand-alone op: out[b][i][l] = sum_j W[i][j] z[b][j][l]
 * for the c x c mixing matrix (c in {2,4,6,8}; z, out [B][c][L], not in place).  transpose_w != 0 applies W^T, which
 * is the op's data gradient; the reverse direction is the same call with W^-1.  log|det W| is a property of the
 * weights alone and is left to the caller (glow.py:100). */
int facppg_conv1x1(const float* w_dev, const float* z_dev, float* out_dev, int B, int c, int L,
                   int transpose_w, void* stream);
/* Its weight gradient dW[i][j] = sum_{b,l} dout[b][i][l] z[b][j][l], summed in a fixed order (bit-reproducible). */
size_t facppg_conv1x1_wgrad_workspace_bytes(int c);
int facppg_conv1x1_wgrad(const float* dout_dev, const float* z_dev, float* dw_dev, int B, int c, int L,
                         void* workspace_dev, size_t workspace_bytes, void* stream);

/* Replaces torch.logdet(W) of Invertible1x1Conv.forward (glow.py:100) and, through winv_t = W^-T, its gradient
 * (d logdet / dW = W^-T): LU with partial pivoting of one c x c matrix (c <= 8) in one launch.  logdet_dev [1],
 * winv_t_dev [c][c].  NaN for a negative determinant, -inf for a singular matrix, as torch.logdet. */
int facppg_logdet(const float* w_dev, int c, float* logdet_dev, float* winv_t_dev, void* stream);

/* Replaces torch.nn.utils.weight_norm's per-conv recomputation (glow.py:118-146: w = g * v / ||v|| per output row) for
 * ALL weight-normed convs of the model in one launch, and its backward in one more.  table_dev: n_tensors entries of
 * 6 x 8 bytes {const float* v; const float* g; float* w; float* norm; int64 row0; int32 rows; int32 len} -- row0 = first
 * global row of the tensor (prefix sum of rows), len = elements per row; forward writes w and norm [rows].
 * Backward: same table with w := dw (read), norm (read); out_table_dev entries' v := dv, g := dg (written). */
int facppg_weight_norm_forward(const void* table_dev, int n_tensors, long total_rows, void* stream);
int facppg_weight_norm_backward(const void* table_dev, const void* out_table_dev, int n_tensors,
                                long total_rows, void* stream);

/* Replaces the reductions of WaveGlowLoss.forward (glow.py:43-59: torch.sum(z*z) and one torch.sum(log_s) per flow, 13 reduction
 * launches + 12 adds) by two launches: out[e] = sum over segment e, a segment being `outer` runs of `inner` contiguous floats
 * `outer_stride` elements apart (a flow's log_s is the upper half of every batch item of its WN output), squared first if
 * `square`.  Double accumulation, fixed summation order (bit-reproducible).  workspace: n * 64 doubles. */
#define FACPPG_MAX_SUM_SEGMENTS 16
typedef struct facppg_sum_segment {
  const float* data_dev;
  long outer_stride;
  int outer, inner;
  int square;
} facppg_sum_segment;
int facppg_segment_sums(const facppg_sum_segment* segs, int n, void* workspace_dev, size_t workspace_bytes, float* out_dev,
                        void* stream);

/* Replaces the affine coupling of WaveGlow.forward (glow.py:240-245): x = [x0 | x1] and wn_out = [b | log_s], all
 * [B][2h][L] fp32 -> y = cat(x0, exp(log_s) * x1 + b); and its backward: dx = [dy0 | dy1 exp(log_s)],
 * dwn_out = [dy1 | dy1 exp(log_s) x1] (the loss's own -sum(log_s) term reaches log_s through autograd separately). */
int facppg_affine_forward(const float* x_dev, const float* wn_out_dev, float* y_dev, int B, int h, int L, void* stream);
int facppg_affine_backward(const float* x_dev, const float* wn_out_dev, const float* dy_dev, float* dx_dev,
                           float* dwn_out_dev, int B, int h, int L, void* stream);

/* ---- WN training primitives (one flow's WN stack; WaveGlow training step, glow.py:154-175 +
 * its autograd backward).  Plain (un-packed, weight-norm already applied) device weights: */
typedef struct facppg_wn_weights {
  const float* start_w; /* [256][n_in] */
  const float* start_b; /* [256] */
  const float* in_w[8];   /* [512][256][3] */
  const float* in_b[8];   /* [512] */
  const float* cond_w[8]; /* [512][640] */
  const float* cond_b[8]; /* [512] */
  const float* rs_w[8];   /* [512][256], last layer [256][256] */
  const float* rs_b[8];
  const float* end_w; /* [2*n_in][256] */
  const float* end_b; /* [2*n_in] */
} facppg_wn_weights;
size_t facppg_wn_train_workspace_bytes(int n_layers, int B, int L);
/* Lr = round_up(L, 64), Lp = 128 + Lr + 128.  Forward of WN keeping what the backward needs:
 * h_all [n_layers+1][B][256][Lp], ts_all [n_layers][B][512][Lr], skip [B][256][Lr]. */
int facppg_wn_forward_save(const facppg_wn_weights* w, int n_in, int n_layers, const float* a0_dev,
                           const float* spect_pad_dev /*[B][640][Lr]*/, int B, int L, float* out_dev,
                           float* h_all_dev, float* ts_all_dev, float* skip_dev, void* workspace_dev,
                           size_t workspace_bytes, void* stream);
/* Backward w.r.t. data; keeps dpre_all [n_layers][B][512][Lr], dh_all [n_layers+1][B][256][Lr] and
 * dskip [B][256][Lr] for the caller's weight-gradient matrix products; dspect [B][640][Lr], da0 [B][n_in][L]. */
int facppg_wn_backward_data(const facppg_wn_weights* w, int n_in, int n_layers, const float* dout_dev,
                            const float* ts_all_dev, int B, int L, float* dpre_all_dev,
                            float* dh_all_dev, float* dskip_dev, float* dspect_dev, float* da0_dev,
                            void* workspace_dev, size_t workspace_bytes, void* stream);

/* Weight and bias gradients of the stack from what facppg_wn_forward_save and facppg_wn_backward_data kept (the parameter half of
 * the autograd backward of src/waveglow/glow.py:154-175; reference: torch autograd over nn.Conv1d): every weight gradient
 * is an NT product over batch and positions on the exact-fp32 MFMA, every bias gradient a row sum, all in two launches,
 * deterministic.  g: fp32 outputs with the shapes of facppg_wn_weights. */
struct facppg_wn_grads;
size_t facppg_wn_weight_grads_workspace_bytes(int n_layers);
int facppg_wn_weight_grads(int n_in, int n_layers, const float* a0_dev, const float* spect_pad_dev, const float* h_all_dev,
                           const float* ts_all_dev, const float* skip_dev, const float* dout_dev, const float* dpre_all_dev,
                           const float* dh_all_dev, const float* dskip_dev, int B, int L, const struct facppg_wn_grads* g,
                           void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- the same stack with bf16 MFMA operands (BASELINE config 5: bf16 training, fp32 accumulation, fp32 master
 * weights and fp32 gradients).  Activations are kept POSITION-major ([B][Lr][channels] bf16, Lr =
 * facppg_wn_bf16_padded_len(L)) because the bf16 MFMA takes 8 consecutive reduction entries per lane; everything --
 * forward, data gradients AND the weight / bias gradients (NT products over the positions) -- runs in this library. */
typedef struct facppg_wn_grads {   /* fp32 outputs, same shapes as facppg_wn_weights */
  float* start_w; float* start_b;
  float* in_w[8]; float* in_b[8];
  float* cond_w[8]; float* cond_b[8];
  float* rs_w[8]; float* rs_b[8];
  float* end_w; float* end_b;
} facppg_wn_grads;
int facppg_wn_bf16_padded_len(int L);
/* ---- the whole model's training direction on the bf16 stack, GROUPS of consecutive flows per call (src/waveglow/glow.py:228-247 and
 * its autograd backward; src/script/train_waveglow.py:126-133).  Between two WN stacks everything is per-position arithmetic on <= 8
 * channels -- the end conv + affine coupling of the flow below (glow.py:175, 240-245), the early-output split (glow.py:231-233), the
 * 1x1 mixing conv (glow.py:98-102) and the start conv (glow.py:156) of the flow above: one launch per flow boundary and direction.
 * Per step: facppg_glow_bf16_begin once (packed bf16 weight images of every stack, zeroed margins of every flow's saved state), then
 * facppg_glow_bf16_group_forward per group in flow order, facppg_glow_bf16_group_backward per group in reverse order. */
#define FACPPG_GLOW_MAX_FLOWS 12
#define FACPPG_GLOW_PART_FLOATS 3400
typedef struct facppg_glow_bf16_sizes {
  size_t packed_bytes_per_flow;  /* bf16 operand images + summed gate biases of one flow's stack */
  size_t state_bytes_per_flow;   /* saved activations of one flow's stack (stride between flows in `states`) */
  size_t work_bytes;             /* gradients in flight of one flow's backward + reduction partials (shared by all flows) */
  int n_parts;                   /* workgroups of a backward edge launch: facppg_glow_flow.part is [n_parts][part_floats] */
  int part_floats;
} facppg_glow_bf16_sizes;
typedef struct facppg_glow_flow {
  const facppg_wn_weights* w;    /* effective (weight-normed) fp32 weights of the flow's WN, n_in = c / 2 */
  const facppg_wn_grads* g;      /* backward: where their gradients go */
  const float* conv_w;           /* [c][c] mixing matrix (Invertible1x1Conv, glow.py:62-102) */
  float* d_conv_w;               /* backward: its gradient, data term + g_logdet * ld_scale * W^-T */
  float* logdet;                 /* [1 + c*c]: ld_scale * log|det W| and W^-T, written by the forward, read by the backward */
  const float* g_logdet;         /* backward: upstream gradient of the flow's log_det_W output (one float), or NULL */
  float ld_scale;                /* B * L (glow.py:100) */
  int c;                         /* channels through the flow */
  int early;                     /* channels split off in front of it as an early output (glow.py:231-233), 0 if none */
  float* early_io; long early_bs; /* [B][early][L], batch stride early_bs floats: forward = the early output, backward = its gradient */
  float* u; float* z; float* wn_out; float* dzp;   /* [B][c][L] each: conv input, conv output, stack output (b | log_s), backward temp */
  const float* dlog_s; long dls_b, dls_j, dls_n;   /* backward: gradient of log_s = wn_out[:, c/2:] and its element strides (NULL: none) */
  float* part;                   /* backward: [n_parts][part_floats] partial sums */
  void* packed; void* state;                       /* this flow's slices of the step's `packed` and `states` buffers */
} facppg_glow_flow;
int facppg_glow_bf16_layout(int n_layers, int B, int L, facppg_glow_bf16_sizes* out);
int facppg_glow_bf16_begin(const facppg_wn_weights* wts /* [n_flows] */, int n_flows, int n_layers, int B, int L, void* packed_dev,
                           void* states_dev, void* work_dev, void* stream);
/* audio_in [B][flows[0].c + flows[0].early][L] -> audio_out [B][flows[n-1].c][L]; *_bs: batch strides in floats (rows are L apart) */
int facppg_glow_bf16_group_forward(const facppg_glow_flow* flows, int n, int n_layers, const float* audio_in_dev, long in_bs,
                                   float* audio_out_dev, long out_bs, const void* spect_pm_dev, int B, int L, void* stream);
int facppg_glow_bf16_group_backward(const facppg_glow_flow* flows, int n, int n_layers, const float* d_audio_out_dev, long d_out_bs,
                                    float* d_audio_in_dev, long d_in_bs, const void* spect_pm_dev, float* dspect_pm_dev, int accumulate_dspect,
                                    void* work_dev, int B, int L, void* stream);

size_t facppg_wn_bf16_state_bytes(int n_layers, int B, int L);    /* saved activations of one stack (forward -> backward) */
size_t facppg_wn_bf16_scratch_bytes(int n_layers, int B, int L);  /* per-call scratch (packed bf16 weight images, gradients in flight) */
/* Which launches facppg_wn_forward_bf16 / facppg_wn_backward_bf16 use for B items of L positions (glow.py:154-175 and its autograd
 * backward), after the FACPPG_TRAIN_FUSED_FWD / _BWD, FACPPG_TRAIN_TILE and FACPPG_WGRAD_TILE overrides -- a diagnostic, host-side only:
 * bit 0: one launch per layer forward (k_wn_fwd) instead of two; bit 1: one launch per layer in the backward chain (k_wn_bwd);
 * bit 2: 256 x 256 weight-gradient tiles (k_wgrad2) for the tap and conditioning products; bits 8..15: positions per tile of the fused
 * launches (64 or 32).  Negative: bad argument. */
int facppg_wn_bf16_launch_plan(int n_layers, int B, int L);
/* fp32 channel-major [B][channels][ld] (first L columns) -> bf16 position-major [B][Lr][channels], rows >= L zero */
int facppg_spect_to_bf16(const float* spect_dev, int B, int channels, int L, int ld, void* out_dev, void* stream);
/* fp32 position-major [B][Lr][channels] -> fp32 channel-major [B][channels][ld] (first L columns) */
int facppg_posmajor_to_f32(const float* src_dev, int B, int channels, int L, float* out_dev, int ld, void* stream);
/* Replaces WN.forward (glow.py:154-175): a0 [B][n_in][L] fp32, spect_pm bf16 [B][Lr][640] -> out [B][2*n_in][L] fp32. */
int facppg_wn_forward_bf16(const facppg_wn_weights* w, int n_in, int n_layers, const float* a0_dev,
                           const void* spect_pm_dev, int B, int L, float* out_dev, void* state_dev,
                           size_t state_bytes, void* scratch_dev, size_t scratch_bytes, void* stream);
/* Its autograd backward, complete: da0 [B][n_in][L], dspect_pm fp32 [B][Lr][640] (overwritten, or added to when
 * accumulate_dspect != 0: the flows of one step share the buffer) and all of `grads`. */
int facppg_wn_backward_bf16(const facppg_wn_weights* w, const facppg_wn_grads* grads, int n_in, int n_layers,
                            const float* a0_dev, const void* spect_pm_dev, const float* dout_dev, int B, int L,
                            const void* state_dev, size_t state_bytes, float* da0_dev, float* dspect_pm_dev,
                            int accumulate_dspect, void* scratch_dev, size_t scratch_bytes, void* stream);
/* Replaces WaveGlow.upsample + crop + regroup in the training direction (glow.py:184-186, 214-222), straight into
 * the bf16 position-major conditioning operand: mel [B][80][T] fp32, up_w [80][80][ksize], up_b [80] ->
 * spect_pm bf16 [B][Lr][640] for L = N/8 group positions (rows >= L zero). */
size_t facppg_upsample_forward_workspace_bytes(int B, int T, int n_mel, int hop, int ksize, int L);
/* workspace (may be NULL: the scalar kernels run): the transposed convolution as an exact-fp32 MFMA matrix product */
int facppg_upsample_regroup_bf16(const float* mel_dev, const float* up_w_dev, const float* up_b_dev, int B,
                                 int T, int n_mel, int hop, int ksize, int L, void* spect_pm_dev,
                                 void* workspace_dev, size_t workspace_bytes, void* stream);
/* Its backward w.r.t. the parameters from the accumulated conditioning gradient dspect_pm fp32 [B][Lr][640]:
 * d_up_w [80][80][ksize], d_up_b [80]. */
size_t facppg_upsample_backward_workspace_bytes(int B, int T, int n_mel, int hop, int ksize, int L);
int facppg_upsample_regroup_backward(const float* mel_dev, const float* dspect_pm_dev, int B, int T, int n_mel,
                                     int hop, int ksize, int L, float* d_up_w_dev, float* d_up_b_dev,
                                     void* workspace_dev, size_t workspace_bytes, void* stream);

/* Average device time (ms) of the dominant kernel (the fused WN layer) over the launches of
 * the most recent facppg_wg_infer on this handle, measured with hipEvents on the stream the
 * kernels ran on when profiling was enabled with facppg_wg_set_profiling(h, 1).  One event pair brackets the
 * back-to-back layer launches of a flow (nothing else runs between them) and the figure is that time / the layers:
 * a pair per launch costs a 78 us launch ~9 us of its own, and the summary of rocprofv3 --kernel-trace would not agree.  enable = n > 1 accumulates
 * over the next n facppg_wg_infer calls instead (their events are created by the set_profiling call itself, so
 * that a timed region creates none); any set_profiling call starts a new accumulation.  Synchronises
 * the recorded events.  *n_launches receives the number of launches averaged. */
int facppg_wg_set_profiling(facppg_wg* h, int enable);
int facppg_wg_last_layer_ms(facppg_wg* h, float* avg_ms, int* n_launches);

/* Shape of the fused-WN-layer launches of the most recent facppg_wg_infer on this handle: frames (phase-major
 * path) or group positions (FACPPG_WG_UNFOLDED=1) per tile, waves per workgroup, workgroups per launch.  The tile
 * width is picked per call from measured round costs; FACPPG_WN_TILE=16|32|64|128 in the environment forces one.
 * Lets the parity tests assert WHICH instantiation of k_wn_layer (glow.py:154-175) they compared with the oracle. */
int facppg_wg_last_launch_shape(const facppg_wg* h, int* tile_frames, int* waves, int* n_tiles);

/* ---- ONE utterance whose mel frames arrive over time (the metric's "batch = 1" case): the conditioning part of every WN
 * layer's gate GEMM ahead of the layers.  No reference counterpart as code -- the reference computes cond_layers[i](spect)
 * inside WN.forward (glow.py:154-175) on the upsampled mel (glow.py:253-259) -- but the same arithmetic: the fused layer
 * kernel accumulates bias + the folded conditioning chunks BEFORE any dilated tap, and that part depends on the mel frames
 * alone.  facppg_wg_cond_seed forms it for a block of frames and every (flow, layer, phase) with the layer kernel's own MFMA
 * sequence and parks the accumulators in `seeds`; facppg_wg_infer_seeded starts the layers from them (12 K chunks instead
 * of 17; first layers 1 instead of 6).  Same samples as facppg_wg_infer, bit for bit (tests/test_gpu_stream.py).
 *
 * facppg_wg_seed_layout: for an utterance of at most T frames, *Tqp = row length of the zero-margined mel buffer
 *   melp [n_mel][Tqp] (frame q at column *margin + q; the margins and every frame that is not (yet) there must be zero),
 *   *seed_bytes = size of the seed buffer.
 * facppg_wg_mel_pad: mel_dev [n_mel][ld] (T frames) -> melp_dev (zero margins written too).
 * facppg_wg_cond_seed: seeds of frames [frame0, frame0 + nframes) (frame0 a multiple of 32; the frames and the 3 before them
 *   must be final in melp), for flows [flow0, flow0 + nflows) (nflows <= 0: all).  block_tiles (1..4) 32-frame tiles share one
 *   pass over a weight image; layers_per_workgroup layers share one staging of the mel window.  *skip_dev != 0 (device, may be
 *   NULL): the launch does nothing.  max_workgroups > 0 bounds the launch to that many workgroups, which take the remaining
 *   work items from *counter_dev (a zeroed int32 on the device; NULL: strided) -- the CUs it leaves stay free for other streams.
 * facppg_wg_infer_seeded: WaveGlow.infer (glow.py:252-293) of the T frames in melp_dev, laid out (melp, seeds) for T_layout >= T
 *   frames.  Frames [0, seeded_frames) start from the seeds (seeded_frames a multiple of 32); the frames behind them run as
 *   unseeded 16-frame tiles of the same launches.  flow_events: NULL or n_flows hipEvent_t (NULL entries allowed): the launches
 *   of flow k first wait for flow_events[k] on `stream` (its seeds are still being formed on another stream).  z_dev / seed /
 *   sigma / audio_dev [T*hop] / workspace (facppg_wg_workspace_bytes(h, 1, T_layout)) as in facppg_wg_infer. */
int facppg_wg_seed_layout(const facppg_wg* h, int T, int* Tqp, int* margin, size_t* seed_bytes);
int facppg_wg_mel_pad(const facppg_wg* h, const float* mel_dev, int T, int ld, float* melp_dev, void* stream);
int facppg_wg_cond_seed(facppg_wg* h, const float* melp_dev, int T, int frame0, int nframes, int block_tiles,
                        int layers_per_workgroup, int flow0, int nflows, float* seeds_dev, size_t seed_bytes,
                        const int32_t* skip_dev, int max_workgroups, int32_t* counter_dev, void* stream);
int facppg_wg_infer_seeded(facppg_wg* h, const float* melp_dev, int T_layout, int T, const float* seeds_dev,
                           int seeded_frames, const float* z_dev, uint64_t seed, float sigma, float* audio_dev,
                           void* workspace_dev, size_t workspace_bytes, void* const* flow_events, void* stream);

/* ------------------------------------------------------------------------------------
 * STFT / mel analysis / denoiser (src/common/stft.py, src/common/layers.py,
 * src/waveglow/denoiser.py)
 * ---------------------------------------------------------------------------------- */
typedef struct facppg_stft facppg_stft;

/* Replaces STFT.__init__ (stft.py:46-77) [+ TacotronSTFT.__init__ layers.py:75-87 when a mel
 * basis is given].  The caller supplies the constant tables (built on the host exactly as the
 * reference builds them) as device arrays:
 *   fwd_basis_dev   [filter_length+2][filter_length]  windowed DFT basis (real rows, then imag rows)
 *   inv_basis_t_dev [filter_length][filter_length+2]  TRANSPOSE of the windowed pinv basis
 *   win_sq_dev      [filter_length]  squared, centre-padded window (audio_processing.py:79-82)
 *   mel_basis_dev   [n_mel][filter_length/2+1] or NULL
 * Synchronises `stream` before returning. */
int facppg_stft_create(int filter_length, int hop_length, const float* fwd_basis_dev,
                       const float* inv_basis_t_dev, const float* win_sq_dev,
                       const float* mel_basis_dev, int n_mel, int device, void* stream,
                       facppg_stft** out);
void facppg_stft_destroy(facppg_stft* h);
size_t facppg_stft_workspace_bytes(const facppg_stft* h, int B, int N);

/* Replaces STFT.transform (stft.py:79-107): audio [B][N] -> magnitude, phase
 * [B][filter_length/2+1][N/hop+1] (phase_dev may be NULL).  n_valid_dev: NULL or [B] sample
 * counts <= N for padded batches (reflect padding is applied at each utterance's own end). */
int facppg_stft_transform(facppg_stft* h, const float* audio_dev, const int32_t* n_valid_dev,
                          int B, int N, float* mag_dev, float* phase_dev, void* workspace_dev,
                          size_t workspace_bytes, void* stream);
/* Replaces STFT.inverse (stft.py:109-138): magnitude, phase [B][cutoff][F] -> [B][hop*(F-1)]. */
int facppg_stft_inverse(facppg_stft* h, const float* mag_dev, const float* phase_dev, int B,
                        int F, float* out_dev, void* workspace_dev, size_t workspace_bytes,
                        void* stream);
/* Replaces TacotronSTFT.mel_spectrogram (layers.py:96-112) without its host-side range assert:
 * audio [B][N] in [-1,1] -> log(clamp(mel_basis . |STFT|, 1e-5)) [B][n_mel][N/hop+1]. */
int facppg_stft_mel(facppg_stft* h, const float* audio_dev, const int32_t* n_valid_dev, int B,
                    int N, float* mel_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
/* Replaces Denoiser.forward (denoiser.py:63-68): STFT, magnitude - bias_spec*strength clamped
 * at 0, inverse STFT with the original phase.  bias_spec_dev [filter_length/2+1];
 * out [B][hop*(N/hop)]. */
int facppg_denoise(facppg_stft* h, const float* audio_dev, const int32_t* n_valid_dev,
                   const float* bias_spec_dev, float strength, int B, int N, float* out_dev,
                   void* workspace_dev, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Tacotron2-style PPG -> mel model (src/common/model.py)
 * ---------------------------------------------------------------------------------- */
/* The model hyper-parameters of create_hparams_stage() (hparams.py:161-241). */
typedef struct facppg_taco_config {
  int32_t n_symbols;                      /* 5816 (full PPG) or 40 (monophone) */
  int32_t symbols_embedding_dim;          /* 600 */
  int32_t encoder_kernel_size;            /* 5 */
  int32_t encoder_n_convolutions;         /* 3 */
  int32_t encoder_embedding_dim;          /* 600 */
  int32_t n_acoustic_feat_dims;           /* 80 */
  int32_t prenet_dim;                     /* 300 */
  int32_t attention_rnn_dim;              /* 300 */
  int32_t decoder_rnn_dim;                /* 300 */
  int32_t attention_dim;                  /* 150 */
  int32_t attention_location_n_filters;   /* 32 */
  int32_t attention_location_kernel_size; /* 31 */
  int32_t attention_window_size;          /* 20; -1 = None (no window) */
  int32_t postnet_embedding_dim;          /* 512 */
  int32_t postnet_kernel_size;            /* 5 */
  int32_t postnet_n_convolutions;         /* 5 */
  float gate_threshold;                   /* 0.5 */
  float bn_eps;                           /* 1e-5 (torch BatchNorm1d default) */
} facppg_taco_config;

typedef struct facppg_taco facppg_taco;

/* fp32 values in the plain weight blob, flattened in THIS order (names: SURVEY.md Appendix B):
 *   encoder.prenet.layers.{0,1}.linear_layer.weight
 *   encoder.convolutions.j: conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var
 *   encoder.lstm: weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0, then the same four *_reverse
 *   decoder.prenet.layers.{0,1}.linear_layer.weight
 *   decoder.attention_rnn: weight_ih, weight_hh, bias_ih, bias_hh
 *   decoder.attention_layer: query_layer.W, memory_layer.W, v.W, location_conv.W, location_dense.W
 *   decoder.decoder_rnn: weight_ih, weight_hh, bias_ih, bias_hh
 *   decoder.linear_projection: W, b ; decoder.gate_layer: W, b
 *   postnet.convolutions.j: conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var */
size_t facppg_taco_weight_count(const facppg_taco_config* cfg);
/* Replaces Tacotron2.__init__ + load_state_dict's weight preparation (model.py:539-546). */
int facppg_taco_create(const facppg_taco_config* cfg, const float* weights_dev, size_t n_floats,
                       int device, void* stream, facppg_taco** out);
void facppg_taco_destroy(facppg_taco* h);
size_t facppg_taco_workspace_bytes(const facppg_taco* h, int B, int Tin);
size_t facppg_taco_decode_workspace_bytes(const facppg_taco* h, int B, int max_steps);
size_t facppg_taco_postnet_workspace_bytes(const facppg_taco* h, int B, int T);
/* Bounds the workgroups (= CUs: a decoder workgroup holds a CU's whole LDS) later facppg_taco_decode calls on this handle may
 * occupy; 0 = no bound beyond the device's.  No reference counterpart (the reference decodes one utterance at a time on
 * the whole device, model.py:489-535): it exists for callers that run the latency-bound decoder of the NEXT batch on a second
 * stream under the MFMA-bound vocoder of the current one (facppg.pipeline.synthesize_stream) and want it to take few CUs
 * away from the vocoder.  A tighter bound selects wider weight slices per workgroup, which cuts the LSTM sums differently:
 * results agree with the unbounded launch to rounding (1e-6 relative on the mel), not bit for bit. */
int facppg_taco_set_decoder_workgroups(facppg_taco* h, int max_workgroups);
/* Which decoder kernel the most recent facppg_taco_decode launched (tests assert the shape they mean to cover, as
 * facppg_wg_last_launch_shape does for the vocoder): *mode = 0 one workgroup per utterance (k_decoder), 1 cooperative
 * slices (k_decoder_coop), 2 split (k_decoder_split: one attention workgroup per utterance + register-resident dense-layer
 * workers); *workgroups = workgroups of that launch.  The reference has one code path (model.py:489-535). */
int facppg_taco_last_decoder_launch(const facppg_taco* h, int* mode, int* workgroups);

/* ---- the decoder's frames while it is still producing them (B = 1, split decoder).  The reference's decoder appends a frame per
 * step to a Python list (model.py:516-531) and the postnet runs on the finished spectrogram (model.py:604-605); here a
 * consumer on another stream may start on the frames as they appear.
 * facppg_taco_set_frame_stream: words_dev = [frames * n_feat] 8-byte words, ZEROED by the caller before every decode, or NULL
 *   to switch publishing off.  The next facppg_taco_decode with B = 1 whose launch is the split decoder and whose max_steps <=
 *   frames then stores every mel value of frame t ALSO as the word {value, t + 1} at words[t * n_feat + row] with an
 *   agent-scope store the moment it exists (the plain mel_dev stores are only guaranteed visible once the launch has ended).
 *   facppg_taco_last_decode_streamed tells whether the most recent decode did.
 * facppg_taco_collect_frames: waits (bounded in wall-clock time, FACPPG_POLL_LIMIT) for frames [frame_a, frame_b) and writes
 *   them channel-major into mel_dev [n_feat][ld].  If the decoder stops short of frame_b (out_length_dev, the decode call's
 *   output, becomes > 0 and <= a wanted frame) the block is VOID: *void_flag_dev = 1 and nothing more is waited for; a block
 *   whose predecessor is void (*prev_flag_dev != 0) is void at once.  Either flag may be NULL.
 * facppg_taco_postnet_range: Postnet.forward + residual (model.py:178-184, 604-605) as a streaming convolution stack.  With f
 *   frames of mel known, layer j (1-based) is final up to f - j*pad columns; the call extends every layer from the f_prev-frame
 *   frontier to the f_new-frame one (its inputs: mel_dev [n_feat][ld], the layers' own earlier columns in the workspace) and
 *   writes the new final columns of mel_post_dev [n_feat][ld_post].  final_T > 0 (= f_new): the utterance has ended there --
 *   every layer runs up to final_T with the convolutions' zero padding behind it.  Every output column is the sum, in the
 *   order, of facppg_taco_postnet's: same bits.  The workspace (facppg_taco_postnet_stream_workspace_bytes(h, max_frames))
 *   carries the layers' columns from call to call.  *skip_dev != 0 (device, may be NULL): the call's launches do nothing. */
int facppg_taco_set_frame_stream(facppg_taco* h, void* words_dev, int frames);
int facppg_taco_last_decode_streamed(const facppg_taco* h, int* streamed);
int facppg_taco_collect_frames(const facppg_taco* h, const void* words_dev, const int32_t* out_length_dev, int frame_a,
                               int frame_b, float* mel_dev, int ld, int32_t* void_flag_dev, const int32_t* prev_flag_dev,
                               void* stream);
size_t facppg_taco_postnet_stream_workspace_bytes(const facppg_taco* h, int max_frames);
int facppg_taco_postnet_range(facppg_taco* h, const float* mel_dev, int ld, int f_prev, int f_new, int final_T,
                              float* mel_post_dev, int ld_post, void* workspace_dev, size_t workspace_bytes,
                              int max_frames, const int32_t* skip_dev, void* stream);

/* Replaces Encoder.inference (model.py:237-249) and the memory_layer projection
 * (model.py:334).  ppg_dev [B][n_symbols][Tin]; lengths_dev NULL or [B] valid frame counts
 * (each utterance is encoded exactly as its own batch-1 run); masks_dev NULL (dropout keep-masks
 * are drawn on the device from `seed`) or uint8 {0,1} [2][B][symbols_embedding_dim][Tin]
 * (the prenet's two always-on p=0.5 dropouts, model.py:132-135).
 * Outputs: memory_dev [B][Tin][E] and processed-memory pm_dev [B][attention_dim][Tin] (an opaque
 * intermediate handed to facppg_taco_decode; positions contiguous). */
int facppg_taco_encode(facppg_taco* h, const float* ppg_dev, const int32_t* lengths_dev,
                       const uint8_t* masks_dev, uint64_t seed, int B, int Tin, float* memory_dev,
                       float* pm_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
/* Replaces Decoder.inference (model.py:489-535) including the attention window mask
 * (utils.py:46-78) and the stop rule (sigmoid(gate) > gate_threshold after appending the frame,
 * else stop at max_steps).  step_limits_dev NULL, or [B] int32: utterance b stops after
 * min(step_limits[b], max_steps) frames at the latest (a padded batch whose utterances each have their
 * own max_decoder_steps, e.g. = their PPG length).  masks_dev NULL or uint8 [max_steps][2][B][prenet_dim].
 * Outputs: mel_dev [B][n_feat][max_steps], gate_dev [B][max_steps], align_dev NULL or
 * [B][max_steps][Tin], out_lengths_dev [B] (= Tout per utterance; columns beyond it untouched).
 * One launch for the whole loop.  The launch shape follows B: up to 9 utterances run as one attention
 * workgroup each plus register-resident dense-layer workers shared by 1-3 of them, up to 120 as 2-38 cooperating workgroups
 * (larger batches in chunks of 120), one workgroup per utterance as the fallback; the first two use hipLaunchCooperativeKernel (the workgroups
 * of an utterance must be co-resident) and every shape returns the same values to fp32 round-off.
 * FACPPG_DECODER_MODE=split|coop|single forces one (FACPPG_EUNSUPPORTED if B does not allow it). */
int facppg_taco_decode(facppg_taco* h, const float* memory_dev, const float* pm_dev,
                       const int32_t* lengths_dev, const int32_t* step_limits_dev,
                       const uint8_t* masks_dev, uint64_t seed, int B,
                       int Tin, int max_steps, float* mel_dev, float* gate_dev, float* align_dev,
                       int32_t* out_lengths_dev, void* workspace_dev, size_t workspace_bytes,
                       void* stream);
/* The always-on p=0.5 dropout draws of both prenets (Prenet.forward, model.py:132-135) as PER-UTTERANCE
 * streams keyed by seeds_dev[b] (uint64 [B]): enc_masks_dev uint8 [2][B][symbols_embedding_dim][Tin] and
 * dec_masks_dev uint8 [max_steps][2][B][prenet_dim] in the layouts facppg_taco_encode / _decode accept
 * (either may be NULL).  Bit (layer, channel, frame) of utterance b does not depend on B, Tin or max_steps. */
int facppg_taco_draw_dropout(const facppg_taco* h, const uint64_t* seeds_dev, int B, int Tin,
                             int max_steps, uint8_t* enc_masks_dev, uint8_t* dec_masks_dev,
                             void* stream);
/* Replaces get_mask_from_lengths_window_and_time_step(memory_lengths, attention_window_size,
 * time_step) (src/common/utils.py:46-78), the integer part of the attention: mask_dev [B][Tmax]
 * uint8, 0 = keep for index in [min(max(0, t-W), len-1), min(t+W, len-1)], 1 = masked (so the last
 * frame stays unmasked once t-W has passed it, utils.py:65-69); window < 0 = no window.  lengths_dev [B]
 * int32, each in [1, Tmax].  Uses the very range function the decoder kernels evaluate the attention on. */
int facppg_attention_window_mask(const int32_t* lengths_dev, int B, int Tmax, int window,
                                 int time_step, uint8_t* mask_dev, void* stream);
/* Replaces Postnet.forward + the residual add (model.py:178-184, 604-605):
 * mel_dev [B][n_feat][ld] (first T columns used) -> mel_post_dev, same layout. */
int facppg_taco_postnet(facppg_taco* h, const float* mel_dev, const int32_t* out_lengths_dev,
                        int B, int T, int ld, float* mel_post_dev, void* workspace_dev,
                        size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * PPG front-end, the part that needs no acoustic-model blob (src/ppg/compute_ppg.py, src/common/feat.py; the Kaldi
 * feature code behind pykaldi): the nnet3 input features and the senone -> monophone reduction.
 * ---------------------------------------------------------------------------------- */
typedef struct facppg_mfcc facppg_mfcc;
/* Replaces kaldi.feat.mfcc.Mfcc(opts) (feat.py:74-100).  The caller folds Kaldi's per-frame linear steps (DC removal,
 * pre-emphasis, window, zero padding, DFT) into basis_dev [2*nbins][frame_length] (real rows, then imaginary rows) and
 * supplies the mel bank mel_dev [n_mel][nbins] and the liftered DCT dct_dev [n_ceps][n_mel].  Synchronises `stream`. */
int facppg_mfcc_create(int frame_length, int frame_shift, int nbins, const float* basis_dev,
                       const float* mel_dev, int n_mel, const float* dct_dev, int n_ceps, int device,
                       void* stream, facppg_mfcc** out);
void facppg_mfcc_destroy(facppg_mfcc* h);
int facppg_mfcc_num_frames(const facppg_mfcc* h, int n_samples);   /* (n + shift/2) / shift: snip_edges = false */
size_t facppg_mfcc_workspace_bytes(const facppg_mfcc* h, int n_samples);
/* Replaces Mfcc.compute_features (feat.py:98): wav_dev [n_samples] (int16-range floats) -> mfcc_dev [T][n_ceps];
 * frames are cut with Kaldi's snip_edges=false reflection; dither is not applied (deterministic); use_energy != 0
 * puts the log frame energy in coefficient 0. */
int facppg_mfcc_compute(facppg_mfcc* h, const float* wav_dev, int n_samples, int use_energy,
                        float* mfcc_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
/* Replaces Kaldi's DownsampleWaveForm / LinearResample (what frame_opts.allow_downsample = True triggers for inputs above
 * 16 kHz, feat.py:85-87): windowed-sinc low-pass at 0.99 * min(fs) / 2, 6 zero crossings.  out_dev holds
 * facppg_resample_num_samples(n_in, fs_in, fs_out) samples. */
int facppg_resample_num_samples(int n_in, int fs_in, int fs_out);
int facppg_resample(const float* wav_dev, int n_in, int fs_in, int fs_out, float* out_dev, void* stream);
/* Replaces apply_cepstral_mean_norm (feat.py:103-118), kaldi.feat.functions.splice_frames(left, right) (edge frames
 * replicated) and apply_feat_transform (feat.py:121-156) in one pass: feats_dev [T][D] -> out_dev [T][M] with
 * transform_dev [M][cols], cols = (left+right+1)*D or one more (affine offset column); transform_dev NULL -> the spliced
 * (and, with do_cmn, mean-normalised) features [T][(left+right+1)*D].  mean_ws_dev: D floats of scratch. */
int facppg_cmn_splice_transform(const float* feats_dev, int T, int D, int do_cmn, int left, int right,
                                const float* transform_dev, int M, int cols, float* out_dev,
                                float* mean_ws_dev, void* stream);
/* Replaces reduce_ppg_dim (compute_ppg.py:73-94): out[t][m] = sum_k ppg[t][k] * transform_t[k][m]
 * (transform_t = the densified pdf -> monophone matrix, TRANSPOSED: [K][M], M <= 64). */
int facppg_reduce_ppg(const float* ppg_dev, const float* transform_t_dev, int T, int K, int M,
                      float* out_dev, void* stream);

/* Replaces torch.optim.Adam(model.parameters(), lr).step() of the training loop (src/script/train_waveglow.py:83,134) for
 * every parameter in ONE streaming launch.  table_dev: n_tensors entries {float* param, const float* grad, float* exp_avg,
 * float* exp_avg_sq, int64 numel} (40 bytes each); chunks_dev: n_chunks pairs {tensor index, chunk index} covering every
 * tensor in chunks of facppg_adam_chunk_elems() elements; step_dev: the step count (a float, as torch's fused Adam keeps
 * it), incremented on the device BEFORE the update so a captured launch advances it at every replay.  Arithmetic of
 * torch._fused_adam_ (amsgrad off, maximize off, L2-style weight_decay). */
int facppg_adam_chunk_elems(void);
int facppg_adam_step(const void* table_dev, int n_tensors, const int32_t* chunks_dev, int n_chunks, float* step_dev,
                     float lr, double beta1, double beta2, float eps, float weight_decay, void* stream);

/* ------------------------------------------------------------------------------------
 * nnet3 TDNN acoustic model: feature frames -> senone posteriors (the "full PPG")
 * (src/ppg/compute_ppg.py:42-70 compute_full_ppg = Kaldi nnet3::DecodableNnetSimple through PyKaldi;
 *  src/common/decode.py:23-38 read_nnet3_model).  The model file (data/am/final.raw) is not shipped by the reference and
 * no Kaldi runtime exists in this image: PARITY UNPINNED at the Kaldi boundary (see common/nnet3.py, oracle/nnet3.py).
 * ---------------------------------------------------------------------------------- */
typedef struct facppg_tdnn facppg_tdnn;

/* One fused layer: y[:, t] = act(W . [x[:, t + first + j*dil], j = 0..taps-1] + b), optionally followed by Kaldi's
 * NormalizeComponent (renorm_target_rms > 0).  Test-mode BatchNorm / FixedAffine layers are folded into W, b by the
 * host (common.nnet3.plan_layers, as nnet3::CollapseModel does at compute_ppg.py:57).  first <= 0 <= first + (taps-1)*dil. */
typedef struct facppg_tdnn_layer {
  int32_t out_dim, in_dim, taps, dil, first, relu;
  float renorm_target_rms;
} facppg_tdnn_layer;

/* Values in the weight blob: per layer W [out_dim][taps*in_dim] (tap-major = Kaldi's Append order) then b [out_dim]. */
size_t facppg_tdnn_weight_count(const facppg_tdnn_layer* layers, int n_layers);
/* final_op: 0 = the last layer's output, 1 = softmax (posteriors), 2 = log-softmax.  weights_dev may be freed on return. */
int facppg_tdnn_create(const facppg_tdnn_layer* layers, int n_layers, int final_op, const float* weights_dev,
                       size_t n_weights, int device, void* stream, facppg_tdnn** out);
void facppg_tdnn_destroy(facppg_tdnn* h);
/* Frames of input context the network needs before / after a frame (nnet3::ComputeSimpleNnetContext). */
int facppg_tdnn_context(const facppg_tdnn* h, int* left, int* right);
size_t facppg_tdnn_workspace_bytes(const facppg_tdnn* h, int T);
/* feats_dev [T][in_dim] row-major (what compute_feat_for_nnet returns) -> out_dev [T][out_dim] row-major; frames beyond
 * the utterance's ends are the first / last frame repeated (DecodableNnetSimple's edge handling). */
int facppg_tdnn_forward(facppg_tdnn* h, const float* feats_dev, int T, float* out_dev, void* workspace_dev,
                        size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FACPPG_H */
