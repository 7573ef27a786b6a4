"""ctypes binding of libfacppg_hip.so (C ABI: include/facppg.h).

The library is the product: if it is missing, or there is no GPU, the synthesis path raises --
there is no CPU fallback (the CPU restatement lives in oracle/ and is test infrastructure).
``import torch`` happens before ``CDLL`` on purpose: libfacppg_hip.so needs ``libamdhip64.so.7``
and must bind to the copy PyTorch-ROCm already loaded, so that torch's streams and device
pointers are valid inside the kernels' launches.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede CDLL, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libfacppg_hip.so")

_lib = None


class FacppgError(RuntimeError):
    pass


class WgConfig(ctypes.Structure):
    """facppg_wg_config (include/facppg.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "n_mel_channels", "hop_length", "n_flows", "n_group", "n_early_every", "n_early_size",
        "wn_layers", "wn_channels", "wn_kernel_size", "upsample_kernel", "alternate_halves")]


class TacoConfig(ctypes.Structure):
    """facppg_taco_config (include/facppg.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "n_symbols", "symbols_embedding_dim", "encoder_kernel_size", "encoder_n_convolutions", "encoder_embedding_dim",
        "n_acoustic_feat_dims", "prenet_dim", "attention_rnn_dim", "decoder_rnn_dim", "attention_dim",
        "attention_location_n_filters", "attention_location_kernel_size", "attention_window_size",
        "postnet_embedding_dim", "postnet_kernel_size", "postnet_n_convolutions")] + \
        [("gate_threshold", ctypes.c_float), ("bn_eps", ctypes.c_float)]


class WnWeights(ctypes.Structure):
    """facppg_wn_weights (include/facppg.h)."""
    _fields_ = [("start_w", ctypes.c_void_p), ("start_b", ctypes.c_void_p),
                ("in_w", ctypes.c_void_p * 8), ("in_b", ctypes.c_void_p * 8),
                ("cond_w", ctypes.c_void_p * 8), ("cond_b", ctypes.c_void_p * 8),
                ("rs_w", ctypes.c_void_p * 8), ("rs_b", ctypes.c_void_p * 8),
                ("end_w", ctypes.c_void_p), ("end_b", ctypes.c_void_p)]


class WnGrads(ctypes.Structure):
    """facppg_wn_grads (include/facppg.h): fp32 gradient outputs, same shapes as WnWeights."""
    _fields_ = WnWeights._fields_


class GlowBf16Sizes(ctypes.Structure):
    """facppg_glow_bf16_sizes (include/facppg.h)."""
    _fields_ = [("packed_bytes_per_flow", ctypes.c_size_t), ("state_bytes_per_flow", ctypes.c_size_t), ("work_bytes", ctypes.c_size_t),
                ("n_parts", ctypes.c_int), ("part_floats", ctypes.c_int)]


class GlowFlow(ctypes.Structure):
    """facppg_glow_flow (include/facppg.h): one flow of a group call of the bf16 training direction."""
    _fields_ = [("w", ctypes.POINTER(WnWeights)), ("g", ctypes.POINTER(WnGrads)),
                ("conv_w", ctypes.c_void_p), ("d_conv_w", ctypes.c_void_p), ("logdet", ctypes.c_void_p), ("g_logdet", ctypes.c_void_p),
                ("ld_scale", ctypes.c_float), ("c", ctypes.c_int), ("early", ctypes.c_int),
                ("early_io", ctypes.c_void_p), ("early_bs", ctypes.c_long),
                ("u", ctypes.c_void_p), ("z", ctypes.c_void_p), ("wn_out", ctypes.c_void_p), ("dzp", ctypes.c_void_p),
                ("dlog_s", ctypes.c_void_p), ("dls_b", ctypes.c_long), ("dls_j", ctypes.c_long), ("dls_n", ctypes.c_long),
                ("part", ctypes.c_void_p), ("packed", ctypes.c_void_p), ("state", ctypes.c_void_p)]


class TdnnLayer(ctypes.Structure):
    """facppg_tdnn_layer (include/facppg.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in ("out_dim", "in_dim", "taps", "dil", "first", "relu")] + [("renorm_target_rms", ctypes.c_float)]


class SumSegment(ctypes.Structure):
    """facppg_sum_segment (include/facppg.h)."""
    _fields_ = [("data_dev", ctypes.c_void_p), ("outer_stride", ctypes.c_long), ("outer", ctypes.c_int), ("inner", ctypes.c_int),
                ("square", ctypes.c_int)]


def _declare(lib):
    c = ctypes
    vp, i32, u64, f32, sz = c.c_void_p, c.c_int32, c.c_uint64, c.c_float, c.c_size_t
    sigs = {
        "facppg_version": (c.c_int, []),
        "facppg_last_error": (c.c_char_p, []),
        "facppg_wg_weight_count": (sz, [c.POINTER(WgConfig)]),
        "facppg_wg_create": (c.c_int, [c.POINTER(WgConfig), vp, sz, c.c_int, vp, c.POINTER(vp)]),
        "facppg_wg_destroy": (None, [vp]),
        "facppg_wg_workspace_bytes": (sz, [vp, c.c_int, c.c_int]),
        "facppg_wg_infer": (c.c_int, [vp, vp, vp, vp, u64, f32, c.c_int, c.c_int, vp, vp, sz, vp]),
        "facppg_wg_log_s_count": (sz, [vp, c.c_int, c.c_int]),
        "facppg_wg_forward": (c.c_int, [vp, vp, vp, c.c_int, c.c_int, c.c_int, vp, vp, vp, sz, vp]),
        "facppg_wn_train_workspace_bytes": (sz, [c.c_int, c.c_int, c.c_int]),
        "facppg_wn_forward_save": (c.c_int, [c.POINTER(WnWeights), c.c_int, c.c_int, vp, vp, c.c_int, c.c_int, vp, vp, vp, vp,
                                             vp, sz, vp]),
        "facppg_wn_backward_data": (c.c_int, [c.POINTER(WnWeights), c.c_int, c.c_int, vp, vp, c.c_int, c.c_int, vp, vp, vp, vp,
                                              vp, vp, sz, vp]),
        "facppg_wn_weight_grads_workspace_bytes": (sz, [c.c_int]),
        "facppg_wn_weight_grads": (c.c_int, [c.c_int, c.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, c.c_int, c.c_int, c.POINTER(WnGrads),
                                             vp, sz, vp]),
        "facppg_conv1x1": (c.c_int, [vp, vp, vp, c.c_int, c.c_int, c.c_int, c.c_int, vp]),
        "facppg_logdet": (c.c_int, [vp, c.c_int, vp, vp, vp]),
        "facppg_conv1x1_wgrad_workspace_bytes": (sz, [c.c_int]),
        "facppg_conv1x1_wgrad": (c.c_int, [vp, vp, vp, c.c_int, c.c_int, c.c_int, vp, sz, vp]),
        "facppg_wn_bf16_padded_len": (c.c_int, [c.c_int]),
        "facppg_wn_bf16_state_bytes": (sz, [c.c_int, c.c_int, c.c_int]),
        "facppg_wn_bf16_scratch_bytes": (sz, [c.c_int, c.c_int, c.c_int]),
        "facppg_wn_bf16_launch_plan": (c.c_int, [c.c_int, c.c_int, c.c_int]),
        "facppg_spect_to_bf16": (c.c_int, [vp, c.c_int, c.c_int, c.c_int, c.c_int, vp, vp]),
        "facppg_posmajor_to_f32": (c.c_int, [vp, c.c_int, c.c_int, c.c_int, vp, c.c_int, vp]),
        "facppg_wn_forward_bf16": (c.c_int, [c.POINTER(WnWeights), c.c_int, c.c_int, vp, vp, c.c_int, c.c_int, vp, vp, sz, vp, sz, vp]),
        "facppg_wn_backward_bf16": (c.c_int, [c.POINTER(WnWeights), c.POINTER(WnGrads), c.c_int, c.c_int, vp, vp, vp, c.c_int, c.c_int,
                                              vp, sz, vp, vp, c.c_int, vp, sz, vp]),
        "facppg_glow_bf16_layout": (c.c_int, [c.c_int, c.c_int, c.c_int, c.POINTER(GlowBf16Sizes)]),
        "facppg_glow_bf16_begin": (c.c_int, [c.POINTER(WnWeights), c.c_int, c.c_int, c.c_int, c.c_int, vp, vp, vp, vp]),
        "facppg_glow_bf16_group_forward": (c.c_int, [c.POINTER(GlowFlow), c.c_int, c.c_int, vp, c.c_long, vp, c.c_long, vp, c.c_int, c.c_int, vp]),
        "facppg_glow_bf16_group_backward": (c.c_int, [c.POINTER(GlowFlow), c.c_int, c.c_int, vp, c.c_long, vp, c.c_long, vp, vp, c.c_int, vp,
                                                      c.c_int, c.c_int, vp]),
        "facppg_upsample_forward_workspace_bytes": (sz, [c.c_int] * 6),
        "facppg_upsample_regroup_bf16": (c.c_int, [vp, vp, vp, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, vp, vp, sz, vp]),
        "facppg_upsample_backward_workspace_bytes": (sz, [c.c_int] * 6),
        "facppg_upsample_regroup_backward": (c.c_int, [vp, vp, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, vp, vp, vp, sz, vp]),
        "facppg_weight_norm_forward": (c.c_int, [vp, c.c_int, c.c_long, vp]),
        "facppg_weight_norm_backward": (c.c_int, [vp, vp, c.c_int, c.c_long, vp]),
        "facppg_affine_forward": (c.c_int, [vp, vp, vp, c.c_int, c.c_int, c.c_int, vp]),
        "facppg_segment_sums": (c.c_int, [c.POINTER(SumSegment), c.c_int, vp, sz, vp, vp]),
        "facppg_affine_backward": (c.c_int, [vp, vp, vp, vp, vp, c.c_int, c.c_int, c.c_int, vp]),
        "facppg_wg_set_profiling": (c.c_int, [vp, c.c_int]),
        "facppg_wg_last_layer_ms": (c.c_int, [vp, c.POINTER(f32), c.POINTER(c.c_int)]),
        "facppg_wg_last_launch_shape": (c.c_int, [vp, c.POINTER(c.c_int), c.POINTER(c.c_int), c.POINTER(c.c_int)]),
        "facppg_wg_seed_layout": (c.c_int, [vp, c.c_int, c.POINTER(c.c_int), c.POINTER(c.c_int), c.POINTER(sz)]),
        "facppg_wg_cond_seed": (c.c_int, [vp, vp, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, vp, sz, vp, c.c_int, vp, vp]),
        "facppg_wg_mel_pad": (c.c_int, [vp, vp, c.c_int, c.c_int, vp, vp]),
        "facppg_wg_infer_seeded": (c.c_int, [vp, vp, c.c_int, c.c_int, vp, c.c_int, vp, c.c_uint64, f32, vp, vp, sz, vp, vp]),
        "facppg_stft_create": (c.c_int, [c.c_int, c.c_int, vp, vp, vp, vp, c.c_int, c.c_int, vp, c.POINTER(vp)]),
        "facppg_stft_destroy": (None, [vp]),
        "facppg_stft_workspace_bytes": (sz, [vp, c.c_int, c.c_int]),
        "facppg_stft_transform": (c.c_int, [vp, vp, vp, c.c_int, c.c_int, vp, vp, vp, sz, vp]),
        "facppg_stft_inverse": (c.c_int, [vp, vp, vp, c.c_int, c.c_int, vp, vp, sz, vp]),
        "facppg_stft_mel": (c.c_int, [vp, vp, vp, c.c_int, c.c_int, vp, vp, sz, vp]),
        "facppg_denoise": (c.c_int, [vp, vp, vp, vp, f32, c.c_int, c.c_int, vp, vp, sz, vp]),
        "facppg_taco_weight_count": (sz, [c.POINTER(TacoConfig)]),
        "facppg_taco_create": (c.c_int, [c.POINTER(TacoConfig), vp, sz, c.c_int, vp, c.POINTER(vp)]),
        "facppg_taco_destroy": (None, [vp]),
        "facppg_taco_workspace_bytes": (sz, [vp, c.c_int, c.c_int]),
        "facppg_taco_decode_workspace_bytes": (sz, [vp, c.c_int, c.c_int]),
        "facppg_taco_set_decoder_workgroups": (c.c_int, [vp, c.c_int]),
        "facppg_taco_last_decoder_launch": (c.c_int, [vp, c.POINTER(c.c_int), c.POINTER(c.c_int)]),
        "facppg_taco_set_frame_stream": (c.c_int, [vp, vp, c.c_int]),
        "facppg_taco_last_decode_streamed": (c.c_int, [vp, c.POINTER(c.c_int)]),
        "facppg_taco_collect_frames": (c.c_int, [vp, vp, vp, c.c_int, c.c_int, vp, c.c_int, vp, vp, vp]),
        "facppg_taco_postnet_stream_workspace_bytes": (sz, [vp, c.c_int]),
        "facppg_taco_postnet_range": (c.c_int, [vp, vp, c.c_int, c.c_int, c.c_int, c.c_int, vp, c.c_int, vp, sz, c.c_int, vp, vp]),
        "facppg_taco_postnet_workspace_bytes": (sz, [vp, c.c_int, c.c_int]),
        "facppg_taco_encode": (c.c_int, [vp, vp, vp, vp, u64, c.c_int, c.c_int, vp, vp, vp, sz, vp]),
        "facppg_taco_decode": (c.c_int, [vp, vp, vp, vp, vp, vp, u64, c.c_int, c.c_int, c.c_int, vp, vp, vp, vp, vp, sz, vp]),
        "facppg_taco_draw_dropout": (c.c_int, [vp, vp, c.c_int, c.c_int, c.c_int, vp, vp, vp]),
        "facppg_wg_draw_noise": (c.c_int, [vp, vp, c.c_int, c.c_int, vp, vp]),
        "facppg_taco_postnet": (c.c_int, [vp, vp, vp, c.c_int, c.c_int, c.c_int, vp, vp, sz, vp]),
        "facppg_mfcc_create": (c.c_int, [c.c_int, c.c_int, c.c_int, vp, vp, c.c_int, vp, c.c_int, c.c_int, vp, c.POINTER(vp)]),
        "facppg_mfcc_destroy": (None, [vp]),
        "facppg_mfcc_num_frames": (c.c_int, [vp, c.c_int]),
        "facppg_mfcc_workspace_bytes": (sz, [vp, c.c_int]),
        "facppg_mfcc_compute": (c.c_int, [vp, vp, c.c_int, c.c_int, vp, vp, sz, vp]),
        "facppg_cmn_splice_transform": (c.c_int, [vp, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, vp, c.c_int, c.c_int, vp, vp, vp]),
        "facppg_resample_num_samples": (c.c_int, [c.c_int, c.c_int, c.c_int]),
        "facppg_resample": (c.c_int, [vp, c.c_int, c.c_int, c.c_int, vp, vp]),
        "facppg_reduce_ppg": (c.c_int, [vp, vp, c.c_int, c.c_int, c.c_int, vp, vp]),
        "facppg_adam_chunk_elems": (c.c_int, []),
        "facppg_adam_step": (c.c_int, [vp, c.c_int, vp, c.c_int, vp, f32, c.c_double, c.c_double, f32, f32, vp]),
        "facppg_tdnn_weight_count": (sz, [c.POINTER(TdnnLayer), c.c_int]),
        "facppg_tdnn_create": (c.c_int, [c.POINTER(TdnnLayer), c.c_int, c.c_int, vp, sz, c.c_int, vp, c.POINTER(vp)]),
        "facppg_tdnn_destroy": (None, [vp]),
        "facppg_tdnn_context": (c.c_int, [vp, c.POINTER(c.c_int), c.POINTER(c.c_int)]),
        "facppg_tdnn_workspace_bytes": (sz, [vp, c.c_int]),
        "facppg_tdnn_forward": (c.c_int, [vp, vp, c.c_int, vp, vp, sz, vp]),
        "facppg_attention_window_mask": (c.c_int, [vp, c.c_int, c.c_int, c.c_int, c.c_int, vp, vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)      # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = res, args
    return sigs


def exported_symbols():
    """Names include/facppg.h declares and the binding uses (checked by the CPU tests)."""
    return sorted(_declare(load()))


def load():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise FacppgError(
                "%s is not built. Build it with `python -c \"import __graft_entry__ as g; g.build()\"` "
                "(or `make -C fac-via-ppg_amd/csrc`). There is no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        _declare(lib)
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise FacppgError("libfacppg_hip: error %d: %s" % (rc, load().facppg_last_error().decode()))


def require_cuda(t, what):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise FacppgError(
            "%s must be a GPU tensor: this build runs only on the HIP kernels of libfacppg_hip.so "
            "(MI355X / gfx950); there is no CPU path." % what)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def upload(values, dtype, device):
    """A small host list -> device tensor WITHOUT stalling the host behind the work already queued on the stream: a copy
    from pageable memory waits for the stream to reach it (with a WaveGlow.infer in front, > 100 ms during which nothing
    else gets enqueued); from a pinned staging tensor it is just another queued command."""
    return torch.tensor(values, dtype=dtype).pin_memory().to(device, non_blocking=True)


def current_stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class WeightIdentity(object):
    """What a packed-weight handle was built from, cheap to re-check on every call: the identity of every parameter, buffer
    AND submodule of ``module`` (re-assigning a Parameter, or replacing a submodule -- ``model.WN[k] = ...`` --, changes an
    id in its parent's dict), and per tensor its in-place version (optimizer steps, any in-place op) and its storage address
    (``p.data = other`` keeps both the id and the version: EMA swaps and the like).  The module tree is walked once; a check
    reads the live dicts' values, and the versions / addresses of the cached flat tensor list only while the ids still match."""

    def __init__(self, module):
        self.dicts = [d for m in module.modules() for d in (m._parameters, m._buffers, m._modules)]
        self.ids = self._ids()
        self.tensors = [v for d in self.dicts for v in d.values() if torch.is_tensor(v)]
        self.state = self._state()

    def _ids(self):
        return tuple(id(v) for d in self.dicts for v in d.values())

    def _state(self):
        return tuple(t._version for t in self.tensors), tuple(t.data_ptr() for t in self.tensors)

    def unchanged(self):
        return self._ids() == self.ids and self._state() == self.state

