"""Glue helpers of the synthesis path -- drop-in for src/common/utils.py.

Device-agnostic where the reference hard-wires ``torch.cuda.*Tensor`` types (utils.py:41,61);
masks are bool (uint8 masks are rejected by current PyTorch, cf. README.md:30).
"""
import numpy as np
import torch
from scipy import signal
from scipy.io.wavfile import read


def get_mask_from_lengths(lengths):
    """True for valid positions (utils.py:39-43)."""
    max_len = int(torch.max(lengths).item())
    ids = torch.arange(0, max_len, device=lengths.device)
    return ids < lengths.unsqueeze(1)


def get_mask_from_lengths_window_and_time_step(lengths, attention_window_size, time_step):
    """Attention window mask, True = masked (utils.py:46-78).  Keeps, per utterance of length n,
    the index range [min(max(0, t-W), n-1), min(t+W, n-1)] -- including the reference's documented
    quirk that the last frame stays unmasked once t-W has passed it.

    ``lengths`` on the GPU (what the reference's decoder passes, model.py:424-427): the mask comes
    from ``facppg_attention_window_mask``, i.e. from the same device range function the decoder
    kernels evaluate the attention on.  Host lengths (lists / CPU tensors): pure integer host logic."""
    if torch.is_tensor(lengths) and lengths.is_cuda:
        from facppg import lib as _lib
        L = _lib.load()
        lt = lengths.to(torch.int32).contiguous()
        B, t_max = lt.numel(), int(lt.max())
        mask = torch.empty(B, t_max, dtype=torch.uint8, device=lt.device)
        with torch.cuda.device(lt.device):
            _lib.check(L.facppg_attention_window_mask(_lib.ptr(lt), B, t_max, -1 if attention_window_size is None
                                                      else int(attention_window_size), int(time_step), _lib.ptr(mask),
                                                      _lib.current_stream(lt.device)))
        return mask.bool()
    lens = [int(v) for v in lengths]
    mask = torch.ones(len(lens), max(lens), dtype=torch.bool)
    for row, n in enumerate(lens):
        first = min(max(0, time_step - attention_window_size), n - 1)
        last = min(time_step + attention_window_size, n - 1)
        if first <= last:
            mask[row, first:last + 1] = False
    return mask


def load_wav_to_torch(full_path):
    sampling_rate, data = read(full_path)
    return torch.FloatTensor(data.astype(np.float32)), sampling_rate


def load_filepaths_and_text(filename, split="|"):
    with open(filename, encoding='utf-8') as f:
        return [tuple(line.strip().split(split)) for line in f]


def load_filepaths(filename):
    """One path per line (utils.py:92-104)."""
    with open(filename) as f:
        return [line.strip() for line in f]


def to_gpu(x):
    """utils.py:107-112"""
    x = x.contiguous()
    if torch.cuda.is_available():
        x = x.cuda(non_blocking=True)
    return x


def notch_filtering(wav, fs, w0, Q):
    """Band-stop filter (utils.py:115-129)."""
    b, a = signal.iirnotch(2 * w0 / fs, Q)
    return signal.lfilter(b, a, wav)


def get_mel(wav, stft):
    """int16-range wav (numpy) -> log-mel [1, n_mel, T]  (utils.py:132-139); runs on the GPU STFT."""
    audio_norm = (torch.FloatTensor(wav.astype(np.float32)) / 32768).unsqueeze(0)
    return stft.mel_spectrogram(to_gpu(audio_norm))


def waveglow_audio(mel, waveglow, sigma, is_cuda_output=False):
    """utils.py:142-152"""
    mel = mel.cuda()
    with torch.no_grad():
        audio = waveglow.infer(mel, sigma=sigma)
    if is_cuda_output:
        return audio
    return (32768 * audio[0]).cpu().numpy().astype('int16')


def get_inference(seq, model, is_clip=False):
    """Tacotron inference on a T*D numpy PPG (utils.py:155-174)."""
    seq = to_gpu(torch.from_numpy(seq).float().transpose(0, 1).unsqueeze(0))
    mel_outputs, mel_outputs_postnet, _, alignments = model.inference(seq)
    if is_clip:
        return mel_outputs_postnet[:, :, 10:(seq.size(2) - 10)]
    return mel_outputs_postnet


def load_waveglow_model(path):
    """utils.py:177-181: checkpoints pickle the whole module, hence weights_only=False."""
    model = torch.load(path, weights_only=False)['model']
    model = model.remove_weightnorm(model)
    model.cuda().eval()
    return model
