"""Legacy WaveGlow layout -- drop-in for src/waveglow/glow_old.py (NVIDIA's first release):
upsampling stride fixed at 256, and odd flows condition on the SECOND half of the channels
(glow_old.py:224-240).  Same HIP kernels as waveglow.glow; the handle is created with
``alternate_halves = 1``.  Like the reference's file, ``forward`` is disabled (glow_old.py:150-151).
"""
from waveglow import glow
from waveglow.glow import Invertible1x1Conv, WN, WaveGlowLoss, fused_add_tanh_sigmoid_multiply, remove  # noqa: F401


class WaveGlow(glow.WaveGlow):
    def __init__(self, n_mel_channels, n_flows, n_group, n_early_every, n_early_size, WN_config):
        super(WaveGlow, self).__init__(n_mel_channels, 256, n_flows, n_group, n_early_every, n_early_size, WN_config)
        self._alternate_halves = True

    def __setstate__(self, state):            # unpickled legacy checkpoints carry no flag
        self.__dict__.update(state)
        self._alternate_halves = True

    def forward(self, forward_input):
        return None
