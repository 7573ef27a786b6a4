"""PPG extraction -- the part of src/ppg/compute_ppg.py that needs no acoustic-model blob, on libfacppg_hip:

  DependenciesPPG          loads the LDA matrix (data/feats/final.mat), the pdf -> monophone reduction
                           (data/feats/reduce_dim.mat) and the splice options; the nnet3 model (data/am/final.raw) is NOT
                           shipped by the reference: ``nnet`` is None when the file is absent
  compute_feat_for_nnet[_internal]   wav -> MFCC -> CMN -> splice -> LDA, the acoustic model's input (compute_ppg.py:97-158)
  reduce_ppg_dim           full (senone) PPG [T, 5816] -> monophone PPG [T, 40] (compute_ppg.py:73-94)
  compute_full_ppg         raises: it needs a Kaldi nnet3 runtime and the missing blob (compute_ppg.py:42-70)

Matrices are float32 GPU tensors; file paths default to ``<repo data dir>/...`` like the reference's module constants and
can be overridden (FACPPG_DATA_DIR, or the constructor arguments)."""
import logging
import os
import re

import torch

from common import feat, kaldi_io
from facppg import lib as _lib

DATA_DIR = os.environ.get("FACPPG_DATA_DIR", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "data"))
NNET_PATH = os.path.join(DATA_DIR, 'am', 'final.raw')
LDA_PATH = os.path.join(DATA_DIR, 'feats', 'final.mat')
REDUCE_DIM_PATH = os.path.join(DATA_DIR, 'feats', 'reduce_dim.mat')
SPLICE_OPTS_PATH = os.path.join(DATA_DIR, 'feats', 'splice_opts')


def compute_full_ppg(nnet, feats):
    raise _lib.FacppgError(
        "compute_full_ppg needs the Kaldi nnet3 acoustic model (data/am/final.raw), which the reference does not ship, and an nnet3 "
        "runtime; this build stops at the model's input features (compute_feat_for_nnet) and reads PPGs from precomputed .npy files")


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


class DependenciesPPG(object):
    """compute_ppg.py:205-256, minus the acoustic model: ``nnet`` is None unless a loader for the blob exists."""

    def __init__(self, nnet_path=NNET_PATH, lda_path=LDA_PATH, reduce_dim_path=REDUCE_DIM_PATH, splice_opts_path=SPLICE_OPTS_PATH):
        self.nnet_path, self.lda_path, self.reduce_dim_path, self.splice_opts_path = nnet_path, lda_path, reduce_dim_path, splice_opts_path
        self.precomputed_only = not os.path.isfile(nnet_path)
        self.nnet = None
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
