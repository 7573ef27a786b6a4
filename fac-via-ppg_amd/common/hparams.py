"""Hyper-parameter surface of the PPG->mel model.

Drop-in for the reference's ``common.hparams`` (src/common/hparams.py:35-241): the
two factories return an attribute bag whose keys and defaults equal the reference's,
keyword overrides replace defaults, and an unknown key raises ``ValueError`` with the
reference's message (hparams.py:150-154, 233-237).  The defaults are kept as one table
with a column per factory instead of two literal dicts; tests/golden/hparams.json (dumped
from the reference) pins both columns.
"""

_ABSENT = object()  # key does not exist in that factory's view


class HParamsView(object):
    """Attribute view over a dict (hparams.py:35-37)."""

    def __init__(self, d):
        self.__dict__ = d


# name -> (create_hparams default, create_hparams_stage default)
_TABLE = {
    # experiment / bookkeeping
    "epochs": (1000, 1000),
    "iters_per_checkpoint": (200, 100),
    "seed": (16807, 16807),
    "dynamic_loss_scaling": (True, True),
    "fp16_run": (False, False),
    "distributed_run": (False, False),
    "dist_backend": ("nccl", "nccl"),
    "dist_url": ("tcp://localhost:54321", "tcp://localhost:54321"),
    "cudnn_enabled": (True, True),
    "cudnn_benchmark": (False, False),
    "output_directory": (None, ""),
    "log_directory": ("log", "log"),
    "checkpoint_path": ("", None),
    "warm_start": (False, False),
    "n_gpus": (1, 1),
    "rank": (0, 0),
    "group_name": ("group_name", "group_name"),
    # data
    "training_files": ("", ""),
    "validation_files": ("", ""),
    "is_full_ppg": (True, True),
    "is_append_f0": (False, False),
    "ppg_subsampling_factor": (1, 1),
    "load_feats_from_disk": (False, True),
    "is_cache_feats": (False, False),
    "feats_cache_path": ("", ""),
    "is_large_set": (_ABSENT, False),
    "is_skip_sil": (_ABSENT, False),
    "mvn_stats_file": (_ABSENT, ""),
    "sequence_level": (_ABSENT, "sentence"),
    # audio (16 kHz / hop 160 / 1024-point window, 80 mel bins)
    "max_wav_value": (32768.0, 32768.0),
    "sampling_rate": (16000, 16000),
    "n_acoustic_feat_dims": (80, 80),
    "filter_length": (1024, 1024),
    "hop_length": (160, 160),
    "win_length": (1024, 1024),
    "mel_fmin": (0.0, 0.0),
    "mel_fmax": (8000.0, 8000.0),
    # model: encoder
    "n_symbols": (5816, 5816),
    "symbols_embedding_dim": (600, 600),
    "encoder_kernel_size": (5, 5),
    "encoder_n_convolutions": (3, 3),
    "encoder_embedding_dim": (600, 600),
    # model: decoder
    "decoder_rnn_dim": (300, 300),
    "prenet_dim": (300, 300),
    "max_decoder_steps": (1000, 1000),
    "gate_threshold": (0.5, 0.5),
    "p_attention_dropout": (0.1, 0.1),
    "p_decoder_dropout": (0.1, 0.1),
    # model: attention (window = +-20 encoder steps; None disables the window)
    "attention_rnn_dim": (300, 300),
    "attention_dim": (150, 150),
    "attention_window_size": (20, 20),
    "attention_location_n_filters": (32, 32),
    "attention_location_kernel_size": (31, 31),
    # model: postnet
    "postnet_embedding_dim": (512, 512),
    "postnet_kernel_size": (5, 5),
    "postnet_n_convolutions": (5, 5),
    # optimisation
    "use_saved_learning_rate": (False, False),
    "learning_rate": (1e-5, 0.0001),
    "weight_decay": (1e-6, 1e-06),
    "grad_clip_thresh": (1.0, 1.0),
    "batch_size": (6, 6),
    "mask_padding": (True, True),
    "mel_weight": (1, 1),
    "gate_weight": (0.005, 0.005),
}


def _build(column, overrides):
    values = {k: v[column] for k, v in _TABLE.items() if v[column] is not _ABSENT}
    for key, val in overrides.items():
        if key not in values:
            raise ValueError('The hyper-parameter %s is not supported.' % key)
        values[key] = val
    return HParamsView(values)


def create_hparams(**kwargs):
    """Training-time defaults (hparams.py:40-158)."""
    return _build(0, kwargs)


def create_hparams_stage(**kwargs):
    """The values used by ``generate_synthesis.py`` (hparams.py:161-241)."""
    return _build(1, kwargs)
