"""Placeholder for the reference's ``ppg`` package (src/ppg/compute_ppg.py).

PPG extraction (Kaldi MFCC -> LDA -> nnet3 acoustic model, needs pykaldi and the missing
data/am/final.raw blob) sits UPSTREAM of the synthesis hot path and is out of scope for this
build.  ``DependenciesPPG`` exists so the CLI surface of
generate_synthesis.py:86 is preserved; PPGs are read from precomputed ``.npy`` files instead
(see common.data_utils.get_ppg)."""


class DependenciesPPG(object):
    """Stand-in for compute_ppg.DependenciesPPG (compute_ppg.py:205-256): holds nothing."""

    def __init__(self, *args, **kwargs):
        self.precomputed_only = True
