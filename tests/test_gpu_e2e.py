"""GPU: end-to-end PPG -> mel -> wav (BASELINE configs 1 and 3) through the drop-in surface:
the generate_synthesis CLI on synthetic checkpoints, and the batched pipeline with injected
dropout masks / z against the CPU oracle of the whole path."""
import os

import numpy as np
import pytest
import torch
from scipy.io import wavfile

from helpers import masks_from_seed, rms
from facppg import synth

pytestmark = pytest.mark.gpu


def weightnorm_state_dict(sd):
    """Plain synthetic WaveGlow weights expressed as weight-norm (g, v) pairs: g = ||v||, v = w."""
    out = {}
    for k, v in sd.items():
        if k.startswith("WN.") and k.endswith(".weight") and ".end." not in k:
            out[k[:-6] + "weight_v"] = v
            out[k[:-6] + "weight_g"] = v.flatten(1).norm(dim=1).view(-1, 1, 1)
        else:
            out[k] = v
    return out


@pytest.fixture(scope="module")
def checkpoints(tmp_path_factory):
    from common.hparams import create_hparams_stage
    from waveglow.glow import WaveGlow
    d = tmp_path_factory.mktemp("ckpt")
    cfg = dict(synth.WAVEGLOW_CONFIG)
    wg = WaveGlow(**cfg)
    wg.load_state_dict(weightnorm_state_dict(synth.waveglow_state_dict(cfg)), strict=True)
    # the reference pickles the whole module (train_waveglow.py:56-64)
    torch.save({"model": wg, "iteration": 0, "optimizer": None, "learning_rate": 1e-4}, d / "waveglow.pt")
    hp = create_hparams_stage()
    torch.save({"state_dict": synth.tacotron_state_dict(hp, gate_bias=-0.02), "iteration": 0}, d / "tacotron.pt")
    return d


def test_generate_synthesis_cli(checkpoints, tmp_path, capsys):
    from script import generate_synthesis
    ppg = synth.synthetic_ppg(60, 5816, seed=3, alpha=0.002)
    utt = tmp_path / "teacher.wav"
    np.save(str(utt) + ".ppg.npy", ppg)
    out = tmp_path / "out"
    torch.manual_seed(0)
    generate_synthesis.main(["--ppg2mel_model", str(checkpoints / "tacotron.pt"), "--waveglow_model",
                             str(checkpoints / "waveglow.pt"), "--teacher_utterance_path", str(utt), "--output_dir", str(out)])
    sr, wav = wavfile.read(out / "ac.wav")
    assert sr == 16000 and wav.dtype == np.float32 and wav.ndim == 1     # written [N, 1] float32 mono
    assert wav.shape[0] % 160 == 0 and wav.shape[0] > 0 and np.isfinite(wav).all()
    log = open(out / "debug.log").read()
    for needle in ("Output dir:", "Sigma: 0.600000", "Denoiser strength: 0.005000", "Denoiser mode: zeros",
                   "Perform AC on", "Done!"):
        assert needle in log
    # missing teacher utterance: warning in the log, exit 0, no wav (generate_synthesis.py:99-100)
    out2 = tmp_path / "out2"
    generate_synthesis.main(["--ppg2mel_model", str(checkpoints / "tacotron.pt"), "--waveglow_model",
                             str(checkpoints / "waveglow.pt"), "--teacher_utterance_path", str(tmp_path / "nope.wav"),
                             "--output_dir", str(out2)])
    assert not os.path.exists(out2 / "ac.wav") and "Missing" in open(out2 / "debug.log").read()


def test_pipeline_matches_oracle_and_batches_equal_singles(checkpoints):
    from common.hparams import create_hparams_stage
    from common.utils import load_waveglow_model
    from facppg import pipeline
    from oracle import dsp, tacotron as otac, waveglow as owg
    from script.train_ppg2mel import load_model
    from waveglow.denoiser import Denoiser
    steps = 48
    hp = create_hparams_stage(max_decoder_steps=steps)
    tsd = synth.tacotron_state_dict(hp, gate_bias=-0.02)
    taco = load_model(hp)
    taco.load_state_dict(tsd)
    taco.eval()
    den = Denoiser(torch.load(checkpoints / "waveglow.pt", weights_only=False)["model"].cuda(), mode="zeros")
    wg = load_waveglow_model(str(checkpoints / "waveglow.pt"))
    cfg = dict(synth.WAVEGLOW_CONFIG)
    wsd = synth.waveglow_state_dict(cfg)

    lens = [33, 20, 27]
    ppgs = [synth.synthetic_ppg(n, 5816, seed=60 + i, alpha=0.002) for i, n in enumerate(lens)]
    B, Tin = len(lens), max(lens)
    em = masks_from_seed(5, (2, B, Tin, 600))
    dm = masks_from_seed(6, (steps, 2, B, 300))
    # run the batch once to learn Tout per utterance, then build per-utterance z of that size
    wavs0, tout = pipeline.synthesize(ppgs, taco, wg, den, sigma=0.6, strength=0.01, dropout_masks=(em, dm), seed=1)
    Lmax = max(tout) * 20
    zs = synth.synthetic_z(B, Lmax, cfg, seed=77)
    wavs, tout2 = pipeline.synthesize(ppgs, taco, wg, den, sigma=0.6, strength=0.01, dropout_masks=(em, dm), z=zs)
    assert tout2 == tout and [len(w) for w in wavs] == [t * 160 for t in tout]
    with torch.no_grad():
        bias = owg.infer(wsd, cfg, torch.zeros(1, 80, 88), 0.0, [torch.zeros(1, 4, 1760), torch.zeros(1, 2, 1760), torch.zeros(1, 2, 1760)])
    oden = dsp.DenoiserOracle(bias)
    for b, n in enumerate(lens):
        emb, dmb = em[:, b:b + 1, :n], dm[:, :, b:b + 1]
        zb = [z[b:b + 1, :, :tout[b] * 20].contiguous() for z in zs]
        single, t1 = pipeline.synthesize([ppgs[b]], taco, wg, den, sigma=0.6, strength=0.01, dropout_masks=(emb, dmb), z=zb)
        assert t1 == [tout[b]] and np.array_equal(single[0], wavs[b])       # batch == independent run, bit-exact
        x = torch.from_numpy(ppgs[b]).t().unsqueeze(0)
        mel, mel_post, gate, align = otac.inference(tsd, hp, x, torch.from_numpy(emb.astype(np.float32)),
                                                    torch.from_numpy(dmb.astype(np.float32)))
        assert mel_post.shape[2] == tout[b]                                  # same stop step as the oracle
        with torch.no_grad():
            ref = oden(owg.infer(wsd, cfg, mel_post, 0.6, zb), 0.01)[0, 0].numpy()
        e = wavs[b] - ref
        print("utt %d: Tout %d, wav rms err %.2e (rms ref %.3f)" % (b, tout[b], rms(e), rms(ref)))
        assert rms(e) <= 1e-3


def test_waveglow_inference_cli_and_mel2samp(checkpoints, tmp_path):
    """waveglow.inference (mel .pt list -> int16 wavs) fed by waveglow.mel2samp's GPU mel analysis."""
    from waveglow import inference
    from waveglow.mel2samp import Mel2Samp
    g = np.random.Generator(np.random.PCG64(2))
    wavs = []
    for i, n in enumerate((4800, 3200)):
        path = tmp_path / ("a%d.wav" % i)
        wavfile.write(path, 16000, (3000 * np.sin(np.arange(n) * 0.05 * (i + 1)) + 50 * g.standard_normal(n)).astype(np.int16))
        wavs.append(str(path))
    (tmp_path / "wavs.txt").write_text("\n".join(wavs) + "\n")
    ds = Mel2Samp(str(tmp_path / "wavs.txt"), segment_length=1600, filter_length=1024, hop_length=160, win_length=1024,
                  sampling_rate=16000, mel_fmin=0.0, mel_fmax=8000.0)
    mel, audio = ds[0]
    assert mel.shape == (80, 1600 // 160 + 1) and audio.shape == (1600,) and float(audio.abs().max()) <= 1.0
    mel_paths = []
    for i, w in enumerate(wavs):
        sr, data = wavfile.read(w)
        m = ds.get_mel(torch.from_numpy(data).float()).cpu()
        assert m.shape == (80, len(data) // 160 + 1)
        torch.save(m, tmp_path / ("m%d.pt" % i))
        mel_paths.append(str(tmp_path / ("m%d.pt" % i)))
    (tmp_path / "mels.txt").write_text("\n".join(mel_paths) + "\n")
    out = tmp_path / "syn"
    torch.manual_seed(3)
    inference.main(str(tmp_path / "mels.txt"), str(checkpoints / "waveglow.pt"), 0.6, str(out), 16000, False)
    for i, w in enumerate(wavs):
        sr, syn = wavfile.read(out / ("m%d_synthesis.wav" % i))
        n_frames = wavfile.read(w)[1].shape[0] // 160 + 1
        assert sr == 16000 and syn.dtype == np.int16 and syn.shape == (n_frames * 160,)
    with pytest.raises(NotImplementedError):
        inference.main(str(tmp_path / "mels.txt"), str(checkpoints / "waveglow.pt"), 0.6, str(out), 16000, True)
