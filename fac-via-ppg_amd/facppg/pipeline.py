"""Batched PPG -> mel -> wav synthesis (BASELINE configs 3 and 4).

The reference synthesises one utterance at a time (generate_synthesis.py:90-95).  ``synthesize``
runs the same three stages -- Tacotron2.inference, WaveGlow.infer, Denoiser -- on a padded batch
with per-utterance lengths threaded through every HIP call, so each utterance's result equals
its own batch-1 run; everything stays on the device until the final copy of the audio.
"""
import numpy as np
import torch


def pad_ppgs(ppgs, device=None):
    """list of [Tin_i, D] arrays -> ([B, D, Tmax] float32 tensor, lengths list).

    With ``device`` the frames are uploaded as they are (time-major, one contiguous copy each) and
    transposed into the channel-major batch on the GPU; the host-side transpose of a 2 s utterance
    ([200, 5816] floats) costs more than the whole encoder."""
    lens = [int(p.shape[0]) for p in ppgs]
    D = int(ppgs[0].shape[1])
    x = torch.zeros(len(ppgs), D, max(lens), dtype=torch.float32, device=device)
    for b, p in enumerate(ppgs):
        t = torch.as_tensor(np.ascontiguousarray(p, dtype=np.float32))
        if device is not None:
            t = t.to(device, non_blocking=True)
        x[b, :, :lens[b]] = t.t()
    return x, lens


def synthesize(ppgs, tacotron, waveglow, denoiser=None, sigma=0.6, strength=0.005, seed=None, dropout_masks=None, z=None,
               return_device=False):
    """Returns (list of float32 waveforms [N_i], list of mel lengths).  Models must be on the GPU."""
    dev = next(tacotron.parameters()).device
    x, lens = pad_ppgs(ppgs, device=dev)
    hop = waveglow.upsample.stride[0]
    with torch.no_grad():
        _, mel_post, _, _ = tacotron.inference(x, lengths=lens if len(lens) > 1 else None,
                                               dropout_masks=dropout_masks, seed=seed)
        tout = [int(v) for v in tacotron.last_output_lengths]
        multi = len(tout) > 1
        audio = waveglow.infer(mel_post.contiguous(), sigma=sigma, z=z, lengths=tout if multi else None, seed=seed)
        if denoiser is not None:
            audio = denoiser(audio, strength=strength, lengths=[t * hop for t in tout] if multi else None)[:, 0]
    if return_device:
        return [audio[b, :tout[b] * hop] for b in range(len(tout))], tout
    host = audio.cpu().numpy()
    return [host[b, :tout[b] * hop].copy() for b in range(len(tout))], tout
