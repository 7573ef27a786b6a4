"""The reference's ``ppg`` package (src/ppg): see compute_ppg.py for what is built (everything around the acoustic
model, whose nnet3 blob the reference does not ship)."""
from ppg.compute_ppg import (DependenciesPPG, compute_feat_for_nnet, compute_feat_for_nnet_internal, compute_full_ppg,  # noqa: F401
                             reduce_ppg_dim)
