"""Kaldi-decoding helpers of the reference's src/common/decode.py that the PPG path uses: ``read_nnet3_model``
(decode.py:23-38).  The transition-model / FST readers (decode.py:41-66) serve Kaldi decoding, which is not on the
PPG -> wav path, and are not built."""
from common import nnet3


def read_nnet3_model(model_path):
    """decode.py:23-38: a raw nnet3 model file -> common.nnet3.Nnet (no Kaldi: see common/nnet3.py for the restated
    file grammar; parity unpinned -- the reference ships no model file)."""
    return nnet3.read_nnet3(model_path)
