"""WaveGlow bias denoiser -- drop-in for src/waveglow/denoiser.py (runs on libfacppg_hip)."""
import torch

from common.layers import STFT


class Denoiser(torch.nn.Module):
    """Removes model bias from audio produced with waveglow (denoiser.py:35-68)."""

    def __init__(self, waveglow, filter_length=1024, hop_length=160, win_length=1024, mode='zeros'):
        super(Denoiser, self).__init__()
        self.stft = STFT(filter_length=filter_length, hop_length=hop_length, win_length=win_length).cuda()
        w = waveglow.upsample.weight
        if mode == 'zeros':
            mel_input = torch.zeros((1, 80, 88), dtype=w.dtype, device=w.device)
        elif mode == 'normal':
            mel_input = torch.randn((1, 80, 88), dtype=w.dtype, device=w.device)
        else:
            raise Exception("Mode {} if not supported".format(mode))
        with torch.no_grad():
            bias_audio = waveglow.infer(mel_input, sigma=0.0).float()
            bias_spec, _ = self.stft.transform(bias_audio)
        self.register_buffer('bias_spec', bias_spec[:, :, 0][:, :, None])

    def forward(self, audio, strength=0.1, lengths=None):
        """audio [B, N] -> denoised [B, 1, hop*(N//hop)]  (denoiser.py:63-68), one fused pass."""
        return self.stft.denoise(audio.cuda().float(), self.bias_spec, strength, lengths)
