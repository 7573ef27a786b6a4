"""The reference's ``ppg`` package (src/ppg): see compute_ppg.py for what is built."""
from ppg.compute_ppg import (DependenciesPPG, compute_feat_for_nnet, compute_feat_for_nnet_internal, compute_full_ppg,  # noqa: F401
                             compute_full_ppg_wrapper, compute_monophone_ppg, reduce_ppg_dim)
