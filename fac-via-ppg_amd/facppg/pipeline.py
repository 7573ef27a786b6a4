"""Batched PPG -> mel -> wav synthesis (BASELINE configs 3 and 4).

The reference synthesises one utterance at a time (generate_synthesis.py:90-95).  ``synthesize``
runs the same three stages -- Tacotron2.inference, WaveGlow.infer, Denoiser -- on a padded batch
with per-utterance lengths threaded through every HIP call, so each utterance's result equals
its own batch-1 run; everything stays on the device until the final copy of the audio.
"""
import os

import numpy as np
import torch


class Synthesizer(object):
    """The three models of the synthesis path loaded from the reference's checkpoint formats and kept on the GPU:
    the PPG->mel state dict (``{'state_dict': ...}``, train_ppg2mel.py:113-119), the pickled WaveGlow module
    (``{'model': ...}``, train_waveglow.py:56-64) once with weight norm removed for synthesis
    (utils.py:177-181) and once as pickled for the denoiser's bias estimate (generate_synthesis.py:58-61)."""

    def __init__(self, ppg2mel_path, waveglow_path, hparams=None, denoiser_mode='zeros'):
        from common.hparams import create_hparams_stage
        from common.utils import load_waveglow_model
        from script.train_ppg2mel import load_model
        from waveglow.denoiser import Denoiser
        self.hparams = hparams if hparams is not None else create_hparams_stage()
        self.tacotron = load_model(self.hparams)
        self.tacotron.load_state_dict(torch.load(ppg2mel_path, weights_only=False)['state_dict'])
        self.tacotron.eval()
        self.denoiser = Denoiser(torch.load(waveglow_path, weights_only=False)['model'].cuda(), mode=denoiser_mode)
        self.waveglow = load_waveglow_model(waveglow_path)

    def __call__(self, ppgs, sigma=0.6, strength=0.005, **kw):
        return synthesize(ppgs, self.tacotron, self.waveglow, self.denoiser, sigma=sigma, strength=strength, **kw)

    def stream(self, jobs, sigma=0.6, strength=0.005, **kw):
        """synthesize_stream over this synthesizer's models: batch i+1's acoustic model under batch i's vocoder."""
        return synthesize_stream(jobs, self.tacotron, self.waveglow, self.denoiser, sigma=sigma, strength=strength, **kw)

    @staticmethod
    def has_utterance(utterance_path):
        """The reference checks the teacher wav itself (generate_synthesis.py:89); here a precomputed PPG next to
        it (or given directly) counts as well, since that is what this build reads."""
        from common.data_utils import ppg_candidates
        return any(os.path.isfile(c) for c in [utterance_path] + ppg_candidates(utterance_path))

    def synthesize_file(self, utterance_path, wav_path, fs, sigma=0.6, strength=0.005, ppg_deps=None):
        """One teacher utterance -> ``wav_path`` (float32 [N, 1] at ``fs``, generate_synthesis.py:90-98)."""
        from common.data_utils import get_ppg
        from scipy.io import wavfile
        wavs, tout = self([get_ppg(utterance_path, ppg_deps)], sigma=sigma, strength=strength)
        wavfile.write(wav_path, fs, wavs[0].astype(np.float32)[:, None])
        return tout[0]


class StageTimer(object):
    """hipEvent timestamps on the launch stream between the stages of one synthesize() call (the kernels run on
    torch's current stream, so torch.cuda.Event brackets them).  ``mark(name)`` closes stage ``name``."""

    def __init__(self):
        self.marks = []
        self.mark("start")

    def mark(self, name):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.marks.append((name, ev))

    def stages_ms(self):
        """{stage: ms} in order, plus 'total'; synchronises the last event."""
        self.marks[-1][1].synchronize()
        out = {}
        for (_, a), (name, b) in zip(self.marks[:-1], self.marks[1:]):
            out[name] = out.get(name, 0.0) + a.elapsed_time(b)
        out["total"] = self.marks[0][1].elapsed_time(self.marks[-1][1])
        return out


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


def _acoustic(ppgs, tacotron, seed, dropout_masks, utterance_seeds, step_limits, timer=None, while_decoding=None):
    """PPG upload + Tacotron2.inference on the current stream -> (mel_post [B, 80, Tout], [Tout_i]).  Blocks the host once,
    for the decoder's output lengths."""
    dev = next(tacotron.parameters()).device
    x, lens = pad_ppgs(ppgs, device=dev)
    if timer is not None:
        timer.mark("ppg_upload")
    _, mel_post, _, _ = tacotron.inference(x, lengths=lens if len(lens) > 1 else None, dropout_masks=dropout_masks,
                                           seed=seed, utterance_seeds=utterance_seeds, step_limits=step_limits,
                                           while_decoding=while_decoding, **({"timer": timer} if timer is not None else {}))
    return mel_post.contiguous(), [int(v) for v in tacotron.last_output_lengths]


def _vocode(mel_post, tout, waveglow, denoiser, sigma, strength, seed, z, utterance_seeds, timer=None):
    """WaveGlow.infer + Denoiser on the current stream -> audio [B, Tout_max * hop] on the device; no host waits beyond the
    small uploads (lengths, seeds)."""
    hop = waveglow.upsample.stride[0]
    multi = len(tout) > 1
    wg_seeds = None if utterance_seeds is None else [int(v) + 1 for v in utterance_seeds]
    audio = waveglow.infer(mel_post, sigma=sigma, z=z, lengths=tout if multi else None, seed=seed, utterance_seeds=wg_seeds)
    if timer is not None:
        timer.mark("waveglow")
    if denoiser is not None:
        audio = denoiser(audio, strength=strength, lengths=[t * hop for t in tout] if multi else None)[:, 0]
        if timer is not None:
            timer.mark("denoiser")
    return audio


def synthesize(ppgs, tacotron, waveglow, denoiser=None, sigma=0.6, strength=0.005, seed=None, dropout_masks=None, z=None,
               return_device=False, utterance_seeds=None, step_limits=None, timer=None, decoder_heaters=-1):
    """Returns (list of float32 waveforms [N_i], list of mel lengths).  Models must be on the GPU.

    decoder_heaters: this is the latency path -- one batch, the vocoder right behind the acoustic model -- so the small-batch
    decoder launch carries heater workgroups on the CUs it leaves empty (Tacotron2.decoder_heaters; -1 = all that fit, 0 = none;
    same samples either way): measured, the vocoder otherwise starts ~10 % slower behind the decoder's milliseconds of low
    activity (0.4-0.6 ms of a 14 ms utterance; cause not identified -- the GPU's reported clocks do not move).  A model that already carries its own setting keeps it.

    utterance_seeds: one integer per utterance -- its dropout and noise streams then depend on that seed alone,
    so the result for an utterance is the same whatever batch, batch size or GPU it is synthesised in.
    step_limits: per-utterance max_decoder_steps (e.g. its PPG length)."""
    if timer is not None:
        timer.__init__()
    hop = waveglow.upsample.stride[0]
    with torch.no_grad():
        dev = next(tacotron.parameters()).device
        own = getattr(tacotron, "decoder_heaters", 0)
        if not own:
            tacotron.decoder_heaters = int(decoder_heaters)
        try:
            mel_post, tout = _acoustic(ppgs, tacotron, seed, dropout_masks, utterance_seeds, step_limits, timer,
                                       while_decoding=lambda: waveglow.prepare(dev))     # host work under the decoder's milliseconds
        finally:
            tacotron.decoder_heaters = own
        audio = _vocode(mel_post, tout, waveglow, denoiser, sigma, strength, seed, z, utterance_seeds, timer)
    if return_device:
        return [audio[b, :tout[b] * hop] for b in range(len(tout))], tout
    host = audio.cpu().numpy()
    return [host[b, :tout[b] * hop].copy() for b in range(len(tout))], tout


_ACOUSTIC_STREAMS = {}


def _acoustic_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _ACOUSTIC_STREAMS:
        _ACOUSTIC_STREAMS[key] = torch.cuda.Stream(device=torch.device("cuda", key))
    return _ACOUSTIC_STREAMS[key]


def synthesize_stream(jobs, tacotron, waveglow, denoiser=None, sigma=0.6, strength=0.005, return_device=True, overlap=True,
                      acoustic_workgroups=32):
    """A sequence of batches, software-pipelined: generator of (waveforms, mel lengths), one per job, in order.

    jobs: iterable of dicts with ``ppgs`` and optionally ``seed``, ``utterance_seeds``, ``step_limits`` (as synthesize()).
    The acoustic model of batch i+1 (PPG upload, encoder, the latency-bound autoregressive decoder, postnet -- a few
    percent of the chip for ~25 ms) runs on its own HIP stream UNDER the vocoder of batch i (MFMA-bound, ~120 ms for 16
    utterances) instead of in front of it.  The host enqueues vocoder i first, then walks through acoustic i+1, whose one
    blocking read (the decoder's output lengths) it would otherwise spend idle.  Every utterance's samples are those of
    synthesize() on the same job: the stages, their inputs and their random streams are the same, only their placement in
    time differs (tests/test_gpu_e2e.py).  overlap=False runs the same jobs back to back on the caller's stream.

    acoustic_workgroups: while overlapped, the decoder is held to this many CUs (Tacotron2.decoder_workgroups, unless the model
    already carries a bound of its own).  A decoder workgroup owns its CU's LDS, so the vocoder loses every CU the decoder
    sits on; on the whole chip (240 workgroups for 16 utterances) the two stages merely take turns (146.7 -> 140.2 ms per
    batch), on 32 CUs the decoder takes 60 instead of 18 ms -- still hidden -- and the vocoder keeps 7/8 of the chip
    (128.3 ms; profiles/r03_experiments.txt).  The slice width that goes with the bound cuts the decoder's LSTM sums
    differently: samples then equal those of synthesize() with the same Tacotron2.decoder_workgroups bit for bit, and those
    of the unbounded decoder to rounding."""
    dev = next(tacotron.parameters()).device
    hop = waveglow.upsample.stride[0]
    main = torch.cuda.current_stream(dev)
    side = _acoustic_stream(dev) if overlap else main

    def acoustic(job):
        with torch.no_grad(), torch.cuda.stream(side):
            # (the inputs are host arrays: nothing on the caller's stream to wait for -- in particular not the previous vocoder)
            bound = tacotron.decoder_workgroups
            if overlap and not bound and acoustic_workgroups:
                tacotron.decoder_workgroups = int(acoustic_workgroups)
            try:
                mel_post, tout = _acoustic(job["ppgs"], tacotron, job.get("seed"), None, job.get("utterance_seeds"), job.get("step_limits"))
            finally:
                tacotron.decoder_workgroups = bound
            done = torch.cuda.Event()
            done.record(side)
        mel_post.record_stream(main)
        return job, mel_post, tout, done

    def finish(audio, tout):
        if return_device:
            return [audio[b, :tout[b] * hop] for b in range(len(tout))], tout
        host = audio.cpu().numpy()
        return [host[b, :tout[b] * hop].copy() for b in range(len(tout))], tout

    it = iter(jobs)
    first = next(it, None)
    if first is None:
        return
    ready = acoustic(first)
    previous_vocoder = None
    while ready is not None:
        job, mel_post, tout, done = ready
        main.wait_event(done)
        with torch.no_grad():
            audio = _vocode(mel_post, tout, waveglow, denoiser, sigma, strength, job.get("seed"), None, job.get("utterance_seeds"))
        vocoder_done = torch.cuda.Event()
        vocoder_done.record(main)
        nxt = next(it, None)
        if nxt is not None:
            # The acoustic model (~25-60 ms) is several times faster than the vocoder it hides under: left alone, the host would run
            # the acoustic models of ALL remaining jobs under the first vocoders and queue every vocoder call -- each with its output,
            # noise and workspace buffers -- far ahead of its execution.  One job of look-ahead is all the overlap needs: acoustic
            # i+1 is enqueued when vocoder i-1 has finished, i.e. as vocoder i starts.
            if overlap and previous_vocoder is not None:
                previous_vocoder.synchronize()
            ready = acoustic(nxt)       # enqueued behind nothing on its own stream: runs while the vocoder above does
        else:
            ready = None
        previous_vocoder = vocoder_done
        yield finish(audio, tout)
