"""GPU: the streamed batch-1 path (facppg.pipeline.ConditioningStream: the postnet and the conditioning part of every WaveNet
layer's gate GEMM run on a second stream WHILE the split decoder is still producing frames) against the path it replaces --
the same utterance with FACPPG_STREAM=0 -- bit for bit, at lengths that end on and off block boundaries, with the decoder
running to its step limit and stopping early on its gate (blocks that never become final are void)."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

from helpers import masks_from_seed
from facppg import synth

pytestmark = pytest.mark.gpu

HOP = 256


@pytest.fixture(scope="module")
def vocoder():
    from waveglow.denoiser import Denoiser
    from waveglow.glow import WaveGlow
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=HOP)
    wg = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
    wg.load_state_dict(synth.waveglow_state_dict(cfg))
    wg = wg.cuda().eval()
    return cfg, wg, Denoiser(wg, hop_length=HOP, mode="zeros")


def acoustic(steps, gate_bias):
    from common.hparams import create_hparams_stage
    from script.train_ppg2mel import load_model
    hp = create_hparams_stage(max_decoder_steps=steps)
    with contextlib.redirect_stdout(io.StringIO()):
        taco = load_model(hp)
    taco.load_state_dict(synth.tacotron_state_dict(hp, gate_bias=gate_bias))
    taco.eval()
    return hp, taco


def run(taco, wg, den, ppg, em, dm, zs, stream, monkeypatch):
    from facppg import pipeline
    monkeypatch.setenv("FACPPG_STREAM", "1" if stream else "0")
    monkeypatch.setenv("FACPPG_STREAM_MIN_FRAMES", "64")     # (short utterances are not streamed by default: they do not gain)
    seen = {}
    inference = taco.inference

    def spy(*a, **kw):
        out = inference(*a, **kw)
        seen["mel_post"] = out[1].detach().clone()
        seen["streamed"] = kw.get("frame_consumer") is not None and kw["frame_consumer"].active
        return out
    taco.inference = spy
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            wavs, tout = pipeline.synthesize([ppg], taco, wg, den, sigma=0.6, strength=0.005, dropout_masks=(em, dm), z=zs)
    finally:
        del taco.inference
    return wavs[0], tout[0], seen


@pytest.mark.parametrize("Tin,steps,gate_bias", [(200, 200, -10.0), (170, 170, -10.0), (96, 96, -10.0), (75, 75, -10.0),
                                                 (130, 400, -10.0), (150, 200, -10.0), (150, 1000, -0.02), (64, 64, -10.0)])
def test_streamed_utterance_equals_the_unstreamed_path_bit_for_bit(vocoder, Tin, steps, gate_bias, monkeypatch):
    from facppg.pipeline import ConditioningStream
    cfg, wg, den = vocoder
    if Tin == 96:                  # (the optional mode of the stream: unbounded seed passes)
        monkeypatch.setenv("FACPPG_STREAM_SPARE_CUS", "-1")
    hp, taco = acoustic(steps, gate_bias)
    ppg = synth.synthetic_ppg(Tin, 5816, seed=Tin, alpha=0.002)
    em = masks_from_seed(21, (2, 1, Tin, hp.symbols_embedding_dim))
    dm = masks_from_seed(22, (steps, 2, 1, hp.prenet_dim))
    ref, t_ref, seen_ref = run(taco, wg, den, ppg, em, dm, None, False, monkeypatch)
    zs = synth.synthetic_z(1, t_ref * HOP // 8, cfg, seed=23)
    ref, t_ref, seen_ref = run(taco, wg, den, ppg, em, dm, zs, False, monkeypatch)
    out, t_out, seen = run(taco, wg, den, ppg, em, dm, zs, True, monkeypatch)
    # the buffers are laid out for the PPG's length + SLACK frames: an utterance the decoder takes further (130 frames in, 400 out)
    # is finished by the one-shot postnet and the ordinary vocoder
    cap = min(steps, -(-(Tin + ConditioningStream.SLACK) // 32) * 32)
    assert not seen_ref["streamed"] and seen["streamed"] == (min(steps, Tin) >= 64 and t_ref <= cap)
    assert taco.last_decoder_launch()[0] == "split"
    assert t_out == t_ref and (gate_bias > -1 or t_ref == steps)
    cs = wg.__dict__["_facppg_cond_stream"]
    print("Tin %d steps %d: Tout %d, streamed %s, blocks %s, %.2f GB held" % (Tin, steps, t_out, seen["streamed"],
                                                                             cs.cuts if seen["streamed"] else None, cs.footprint_bytes() / 1e9))
    assert torch.equal(seen["mel_post"], seen_ref["mel_post"])                     # the streaming postnet: same bits
    assert out.shape == ref.shape == (t_ref * HOP,) and np.array_equal(out, ref)    # ... and so the samples


def test_blocks_that_time_out_are_redone_behind_the_decoder(vocoder, monkeypatch):
    """k_collect_frames gives a block up when its frames do not arrive within FACPPG_STREAM_WAIT_MS (a profiler that serialises
    kernels, a cooperative launch queued behind another process): the block and every block behind it is void -- no postnet
    columns, no seeds.  The host reads the flags back with the length and redoes those frames behind the decoder: late, not wrong.
    With a 1 us limit every block times out; the samples must still equal the unstreamed path's bit for bit."""
    cfg, wg, den = vocoder
    Tin = steps = 200
    hp, taco = acoustic(steps, -10.0)
    ppg = synth.synthetic_ppg(Tin, 5816, seed=5, alpha=0.002)
    em = masks_from_seed(61, (2, 1, Tin, hp.symbols_embedding_dim))
    dm = masks_from_seed(62, (steps, 2, 1, hp.prenet_dim))
    zs = synth.synthetic_z(1, steps * HOP // 8, cfg, seed=63)
    ref, t_ref, _ = run(taco, wg, den, ppg, em, dm, zs, False, monkeypatch)
    out, t_out, seen = run(taco, wg, den, ppg, em, dm, zs, True, monkeypatch)
    cs = wg.__dict__["_facppg_cond_stream"]
    assert seen["streamed"] and cs.void_blocks == 0 and np.array_equal(out, ref)
    monkeypatch.setenv("FACPPG_STREAM_WAIT_MS", "0.001")
    out, t_out, seen = run(taco, wg, den, ppg, em, dm, zs, True, monkeypatch)
    print("blocks", cs.cuts, "void", cs.void_blocks, "seeded frames", cs.seeded)
    assert seen["streamed"] and cs.void_blocks > 0 and cs.seeded < 160
    assert t_out == t_ref and np.array_equal(out, ref)


def test_stream_footprint_follows_the_utterance_not_the_step_limit(vocoder, monkeypatch):
    """The default max_decoder_steps is 1000: the stream must not hold 6.4 GB of seeds for a 2 s utterance.  Its buffers are laid
    out for the PPG's length + 64 frames, grow only, and the side streams are created once."""
    cfg, wg, den = vocoder
    hp, taco = acoustic(1000, -10.0)
    wg.__dict__.pop("_facppg_cond_stream", None)
    held = []
    for Tin in (136, 200, 150):
        ppg = synth.synthetic_ppg(Tin, 5816, seed=Tin, alpha=0.002)
        em = masks_from_seed(71, (2, 1, Tin, hp.symbols_embedding_dim))
        dm = masks_from_seed(72, (Tin, 2, 1, hp.prenet_dim))
        from facppg import pipeline
        monkeypatch.setenv("FACPPG_STREAM", "1")
        with contextlib.redirect_stdout(io.StringIO()):
            wavs, tout = pipeline.synthesize([ppg], taco, wg, den, sigma=0.6, strength=0.005, dropout_masks=(em, dm), step_limits=[Tin], seed=3)
        cs = wg.__dict__["_facppg_cond_stream"]
        held.append((cs.footprint_bytes(), cs.side, cs.cap))
        assert tout[0] == Tin and cs.cap == Tin                      # (step_limits bounds the layout too)
    print("held GB:", ["%.2f" % (h[0] / 1e9) for h in held])
    assert held[1][0] <= 2.0e9 and held[2][0] == held[1][0] and held[2][1] is held[0][1]


def test_stream_buffers_are_reused_across_utterances(vocoder, monkeypatch):
    """The stream's buffers (frame words, void flags, work counters, mel buffer, seeds) live with the models and are reused by
    every utterance of the same step limit: three different utterances in a row, each streamed and unstreamed, and the first one
    again at the end -- every streamed result equals its unstreamed one bit for bit (nothing of an earlier utterance leaks)."""
    cfg, wg, den = vocoder
    steps = 136
    hp, taco = acoustic(steps, -10.0)
    cases = []
    for i, Tin in enumerate((136, 90, 120)):
        ppg = synth.synthetic_ppg(Tin, 5816, seed=70 + i, alpha=0.002)
        em = masks_from_seed(31 + i, (2, 1, Tin, hp.symbols_embedding_dim))
        dm = masks_from_seed(41 + i, (steps, 2, 1, hp.prenet_dim))
        zs = synth.synthetic_z(1, steps * HOP // 8, cfg, seed=51 + i)
        cases.append((ppg, em, dm, zs))
    refs = [run(taco, wg, den, *c, False, monkeypatch)[0] for c in cases]
    stream_obj = None
    for i in (0, 1, 2, 0):
        out, t_out, seen = run(taco, wg, den, *cases[i], True, monkeypatch)
        assert seen["streamed"] and t_out == steps
        cs = wg.__dict__["_facppg_cond_stream"]
        assert stream_obj is None or cs is stream_obj            # the same object, the same buffers
        stream_obj = cs
        assert np.array_equal(out, refs[i]), i
