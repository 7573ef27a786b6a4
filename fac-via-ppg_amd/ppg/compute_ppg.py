"""PPG extraction -- the part of src/ppg/compute_ppg.py that needs no acoustic-model blob, on libfacppg_hip:

  DependenciesPPG          loads the LDA matrix (data/feats/final.mat), the pdf -> monophone reduction
                           (data/feats/reduce_dim.mat) and the splice options; the nnet3 model (data/am/final.raw) is NOT
                           shipped by the reference: ``nnet`` is None when the file is absent
  compute_feat_for_nnet[_internal]   wav -> MFCC -> CMN -> splice -> LDA, the acoustic model's input (compute_ppg.py:97-158)
  reduce_ppg_dim           full (senone) PPG [T, 5816] -> monophone PPG [T, 40] (compute_ppg.py:73-94)
  compute_full_ppg         nnet3 TDNN inference on the exact-fp32 MFMA GEMM (csrc/facppg_tdnn.hip): acoustic-model input
                           features [T, 40] -> senone posteriors [T, K] (compute_ppg.py:42-70).  The model is read by
                           common.decode.read_nnet3_model / common.nnet3 -- PARITY UNPINNED: the reference ships neither
                           the model (data/am/final.raw) nor anything Kaldi computed from it

Matrices are float32 GPU tensors; file paths default to ``<repo data dir>/...`` like the reference's module constants and
can be overridden (FACPPG_DATA_DIR, or the constructor arguments)."""
import logging
import os
import re

import numpy as np
import torch

from common import decode, feat, kaldi_io, nnet3
from facppg import lib as _lib

DATA_DIR = os.environ.get("FACPPG_DATA_DIR", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "data"))
NNET_PATH = os.path.join(DATA_DIR, 'am', 'final.raw')
LDA_PATH = os.path.join(DATA_DIR, 'feats', 'final.mat')
REDUCE_DIM_PATH = os.path.join(DATA_DIR, 'feats', 'reduce_dim.mat')
SPLICE_OPTS_PATH = os.path.join(DATA_DIR, 'feats', 'splice_opts')


class _TdnnHandle(object):
    """The packed HIP form of one nnet3 model on one device (kept on the Nnet object)."""

    def __init__(self, nnet, dev):
        L = _lib.load()
        layers, final = nnet3.plan_layers(nnet)
        table = (_lib.TdnnLayer * len(layers))()
        blob = []
        for i, l in enumerate(layers):
            out_dim, k = l["W"].shape
            table[i].out_dim, table[i].in_dim, table[i].taps = out_dim, k // l["taps"], l["taps"]
            table[i].dil, table[i].first, table[i].relu = l["dil"], l["first"], int(l["act"] == "relu")
            table[i].renorm_target_rms = float(l["renorm"] or 0.0)
            blob += [l["W"].reshape(-1), l["b"].reshape(-1)]
        flat = torch.from_numpy(np.concatenate(blob).astype(np.float32)).to(dev)
        self.handle = _lib.ctypes.c_void_p()
        self.dev, self.out_dim, self.in_dim = dev, layers[-1]["W"].shape[0], layers[0]["W"].shape[1] // layers[0]["taps"]
        with torch.cuda.device(dev):
            _lib.check(L.facppg_tdnn_create(table, len(layers), {"none": 0, "softmax": 1, "log-softmax": 2}[final], _lib.ptr(flat),
                                            flat.numel(), dev.index, _lib.current_stream(dev), _lib.ctypes.byref(self.handle)))

    def __del__(self):
        try:
            _lib.load().facppg_tdnn_destroy(self.handle)
        except Exception:
            pass


def compute_full_ppg(nnet, feats):
    """compute_ppg.py:42-70: nnet (common.nnet3.Nnet) x feats [T, D] (GPU tensor or numpy) -> raw PPGs [T, K] on the GPU,
    K = number of senones.  Batch-norm in test mode, frames beyond the utterance = the edge frames repeated, acoustic
    scale 1, no priors -- what the reference configures on DecodableNnetSimple."""
    if nnet is None:
        raise _lib.FacppgError("compute_full_ppg: no acoustic model -- the reference does not ship data/am/final.raw; pass a model read "
                               "with common.decode.read_nnet3_model, or use precomputed PPGs (common.data_utils.get_ppg)")
    L = _lib.load()
    feats = torch.as_tensor(feats)
    if not feats.is_cuda:
        feats = feats.cuda()
    feats = feats.float().contiguous()
    dev = feats.device
    h = getattr(nnet, "_facppg_tdnn", None)
    if h is None or h.dev != dev:
        h = nnet._facppg_tdnn = _TdnnHandle(nnet, dev)
    T, D = feats.shape
    if D != h.in_dim:
        raise _lib.FacppgError("compute_full_ppg: features have %d dims, the model's input node has %d" % (D, h.in_dim))
    out = torch.empty(T, h.out_dim, device=dev)
    ws = torch.empty(L.facppg_tdnn_workspace_bytes(h.handle, T), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.facppg_tdnn_forward(h.handle, _lib.ptr(feats), T, _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.current_stream(dev)))
    return out


def reduce_ppg_dim(ppgs, transform):
    """ppgs [T, D] (GPU tensor or numpy) x transform [d, D] (dense; read_sparse_mat) -> [T, d] on the GPU."""
    L = _lib.load()
    ppgs = torch.as_tensor(ppgs)
    if not ppgs.is_cuda:
        ppgs = ppgs.cuda()
    ppgs = ppgs.float().contiguous()
    dev = ppgs.device
    tr_t = torch.as_tensor(transform).to(dev).float().t().contiguous()            # [D, d]
    T, D = ppgs.shape
    if tr_t.shape[0] != D:
        raise _lib.FacppgError("reduce_ppg_dim: PPG has %d dims, the transform expects %d" % (D, tr_t.shape[0]))
    out = torch.empty(T, tr_t.shape[1], device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.facppg_reduce_ppg(_lib.ptr(ppgs), _lib.ptr(tr_t), T, D, tr_t.shape[1], _lib.ptr(out), _lib.current_stream(dev)))
    return out


def compute_feat_for_nnet_internal(wav, lda, **kwargs):
    """compute_ppg.py:97-134, same options and defaults."""
    options = {"is_use_energy": False, "is_downsample": True, "frame_shift": 10, "is_snip_edges": False, "left_context": 3,
               "right_context": 3}
    for key, val in kwargs.items():
        if key in options:
            options[key] = val
        else:
            logging.error("Option %s not allowed!" % (key))
    mfcc_opts = feat.MfccOptions()
    mfcc_opts.use_energy = options["is_use_energy"]
    mfcc_opts.frame_opts.allow_downsample = options["is_downsample"]
    mfcc_opts.frame_opts.frame_shift_ms = options["frame_shift"]
    mfcc_opts.frame_opts.snip_edges = options["is_snip_edges"]
    mfccs = feat.compute_mfcc_feats(wav, mfcc_opts)
    return feat.cmn_splice_transform(mfccs, options["left_context"], options["right_context"], lda)


def compute_feat_for_nnet(wav_path, lda_path):
    """compute_ppg.py:137-158"""
    if not os.path.exists(wav_path):
        logging.error("File %s does not exist." % (wav_path))
    if not os.path.exists(lda_path):
        logging.error("Transform file %s does not exist." % (lda_path))
    return compute_feat_for_nnet_internal(feat.read_wav_kaldi(wav_path), torch.from_numpy(kaldi_io.read_matrix(lda_path)))


def compute_monophone_ppg(wav, nnet, lda, transform, shift=10):
    """compute_ppg.py:161-181: wav -> features -> full PPG -> monophone PPG, as a numpy array."""
    feats = compute_feat_for_nnet_internal(wav, lda, frame_shift=shift)
    return reduce_ppg_dim(compute_full_ppg(nnet, feats), transform).cpu().numpy()


def compute_full_ppg_wrapper(wav, nnet, lda, shift=10):
    """compute_ppg.py:184-202"""
    feats = compute_feat_for_nnet_internal(wav, lda, frame_shift=shift)
    return compute_full_ppg(nnet, feats).cpu().numpy()


class DependenciesPPG(object):
    """compute_ppg.py:205-256.  The reference does not ship the acoustic model: ``nnet`` is None (and ``precomputed_only``
    True) when ``nnet_path`` does not exist, else the model read by common.decode.read_nnet3_model."""

    def __init__(self, nnet_path=NNET_PATH, lda_path=LDA_PATH, reduce_dim_path=REDUCE_DIM_PATH, splice_opts_path=SPLICE_OPTS_PATH):
        self.nnet_path, self.lda_path, self.reduce_dim_path, self.splice_opts_path = nnet_path, lda_path, reduce_dim_path, splice_opts_path
        self.precomputed_only = not os.path.isfile(nnet_path)
        self.nnet = None if self.precomputed_only else decode.read_nnet3_model(nnet_path)
        self.context_parser = re.compile(r"--left-context=(\d+) --right-context=(\d+)")
        self.lda = self.monophone_trans = None
        self.splice_opts, self.left_context, self.right_context = "", None, None
        if os.path.isfile(lda_path):
            self.lda = torch.from_numpy(kaldi_io.read_matrix(lda_path))
        if os.path.isfile(reduce_dim_path):
            self.monophone_trans = feat.read_sparse_mat(reduce_dim_path)
        if os.path.isfile(splice_opts_path):
            with open(splice_opts_path, 'r') as reader:
                self.splice_opts = reader.readline()
            context = self.context_parser.match(self.splice_opts) if self.splice_opts else None
            if context:
                self.left_context, self.right_context = context.groups()
            else:
                logging.warning("Splice options are empty.")
