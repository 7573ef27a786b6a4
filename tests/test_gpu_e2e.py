"""GPU: end-to-end PPG -> mel -> wav (BASELINE configs 1 and 3) through the drop-in surface:
the generate_synthesis CLI on synthetic checkpoints, and the batched pipeline with injected
dropout masks / z against the CPU oracle of the whole path."""
import contextlib
import io
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


def test_generate_synthesis_cli_output_matches_reference_golden(tmp_path, monkeypatch):
    """a20: the CLI's ac.wav against tests/golden/e2e_cli.npz -- the body of the reference's generate_synthesis.py
    (get_inference -> waveglow_audio -> Denoiser) run on the imported reference with the same synthetic checkpoints.
    The only intervention is INJECTION of the random draws the fixture captured (prenet dropout masks, z), done by
    wrapping the two model entry points the CLI reaches; everything else (checkpoint loading incl. weight-norm
    removal, hparams, the denoiser's bias estimate, the WAV writer) is the CLI's own code."""
    from common.hparams import create_hparams_stage
    from common.model import Tacotron2
    from helpers import golden
    from script import generate_synthesis
    from waveglow.glow import WaveGlow
    d = golden("e2e_cli.npz")
    Tin, Tout = int(d["Tin"]), int(d["Tout"])
    cfg = dict(synth.WAVEGLOW_CONFIG)
    wg = WaveGlow(**cfg)
    wg.load_state_dict(weightnorm_state_dict(synth.waveglow_state_dict(cfg)), strict=True)
    torch.save({"model": wg, "iteration": 0, "optimizer": None, "learning_rate": 1e-4}, tmp_path / "waveglow.pt")
    hp = create_hparams_stage()
    torch.save({"state_dict": synth.tacotron_state_dict(hp, gate_bias=float(d["gate_bias"])), "iteration": 0}, tmp_path / "tacotron.pt")
    ppg = synth.synthetic_ppg(Tin, int(d["n_symbols"]), seed=int(d["ppg_seed"]), alpha=float(d["ppg_alpha"]))
    np.save(str(tmp_path / "teacher.wav") + ".ppg.npy", ppg)
    em = masks_from_seed(int(d["enc_mask_seed"]), (2, 1, Tin, hp.symbols_embedding_dim))
    dm = masks_from_seed(int(d["dec_mask_seed"]), (hp.max_decoder_steps, 2, 1, hp.prenet_dim))
    taco_inference, wg_infer = Tacotron2.inference, WaveGlow.infer
    seen = {}

    def inference_injected(self, x, **kw):
        out = taco_inference(self, x, dropout_masks=(em, dm))
        seen["mel_post"] = out[1].cpu().numpy()
        return out

    def infer_injected(self, spect, sigma=1.0, **kw):
        if sigma == 0.0:                       # the denoiser's bias estimate: z is multiplied by 0
            return wg_infer(self, spect, sigma=sigma)
        seen["audio"] = wg_infer(self, spect, sigma=sigma, z=synth.synthetic_z(1, spect.shape[2] * 20, cfg, seed=int(d["z_seed"])))
        return seen["audio"]

    monkeypatch.setattr(Tacotron2, "inference", inference_injected)
    monkeypatch.setattr(WaveGlow, "infer", infer_injected)
    out = tmp_path / "out"
    generate_synthesis.main(["--ppg2mel_model", str(tmp_path / "tacotron.pt"), "--waveglow_model", str(tmp_path / "waveglow.pt"),
                             "--teacher_utterance_path", str(tmp_path / "teacher.wav"), "--output_dir", str(out)])
    sr, wav = wavfile.read(out / "ac.wav")
    assert sr == 16000 and wav.dtype == np.float32
    assert seen["mel_post"].shape == d["mel_post"].shape == (1, 80, Tout)                 # same stop frame
    assert np.abs(seen["mel_post"] - d["mel_post"]).max() <= 1e-4
    assert wav.shape == (Tout * 160,) == d["ac_wav"][:, 0].shape                          # integer: N = Tout * hop
    e_pre = rms(seen["audio"].cpu().numpy() - d["audio"])
    e = rms(wav - d["ac_wav"][:, 0])
    print("CLI vs reference golden: mel_post %.2e, audio rms err %.2e, ac.wav rms err %.2e (rms %.3f)" % (
        np.abs(seen["mel_post"] - d["mel_post"]).max(), e_pre, e, rms(d["ac_wav"])))
    assert e_pre <= 1e-3 and e <= 1e-3


def _config3_lengths(n, seed, scale=1.0):
    """Tin_i = 100 + PCG64(seed).integers(0, 301) frames (1-4 s)."""
    g = np.random.Generator(np.random.PCG64(seed))
    return [max(8, int(round(v * scale))) for v in (100 + g.integers(0, 301, size=n)).tolist()]


def test_config3_batch16_ragged_end_to_end(checkpoints, monkeypatch):
    """BASELINE config 3: 16 variable-length utterances (Tin_i = 100 + PCG64(7).integers(0,301), i.e. 1-4 s),
    max_decoder_steps = Tin_i per utterance, PPG -> mel -> wav -> denoised in ONE padded batch with
    per-utterance random streams.  Every utterance must equal its own batch-1 run bit for bit (same decoder launch
    shape forced for both; across shapes the LSTM sums are cut differently), and the three shortest are checked
    against the CPU oracle of the whole path fed with the very masks / noise the device drew."""
    from common.hparams import create_hparams_stage
    from common.utils import load_waveglow_model
    from facppg import pipeline
    from oracle import dsp, tacotron as otac, waveglow as owg
    from script.train_ppg2mel import load_model
    from waveglow.denoiser import Denoiser
    monkeypatch.setenv("FACPPG_DECODER_MODE", "coop")
    monkeypatch.setenv("FACPPG_DECODER_COOP_U", "20")         # what B = 16 picks by itself (15 workgroups per utterance)
    monkeypatch.setenv("FACPPG_BILSTM_MODE", "single")        # B = 16 is beyond the register-resident BiLSTM (B <= 12)
    lens = _config3_lengths(16, 7)
    assert len(set(lens)) > 8 and min(lens) >= 100 and max(lens) <= 400
    hp = create_hparams_stage(max_decoder_steps=max(lens))
    tsd = synth.tacotron_state_dict(hp, gate_bias=-10.0)
    taco = load_model(hp)
    taco.load_state_dict(tsd)
    taco.eval()
    den = Denoiser(torch.load(checkpoints / "waveglow.pt", weights_only=False)["model"].cuda(), mode="zeros")
    wg = load_waveglow_model(str(checkpoints / "waveglow.pt"))
    cfg = dict(synth.WAVEGLOW_CONFIG)
    wsd = synth.waveglow_state_dict(cfg)
    ppgs = [synth.synthetic_ppg(n, 5816, seed=300 + i, alpha=0.002) for i, n in enumerate(lens)]
    seeds = [9000 + 17 * i for i in range(16)]
    wavs, tout = pipeline.synthesize(ppgs, taco, wg, den, sigma=0.6, strength=0.005, utterance_seeds=seeds, step_limits=lens)
    assert tout == lens and [len(w) for w in wavs] == [n * 160 for n in lens]        # integer facts: Tout_i, N_i = Tout_i*hop
    assert all(np.isfinite(w).all() for w in wavs)
    for b in range(16):
        single, t1 = pipeline.synthesize([ppgs[b]], taco, wg, den, sigma=0.6, strength=0.005, utterance_seeds=[seeds[b]],
                                         step_limits=[lens[b]])
        assert t1 == [lens[b]] and np.array_equal(single[0], wavs[b]), b
    # the same batch three times, then two smaller ones, through the software-pipelined stream (acoustic model of job i+1 on a
    # second HIP stream under the vocoder of job i).  With the decoder left on the whole chip (acoustic_workgroups=0) every
    # job's samples are those of its own synthesize() call ...
    jobs = [{"ppgs": ppgs, "utterance_seeds": seeds, "step_limits": lens}] * 3 + \
           [{"ppgs": ppgs[:5], "utterance_seeds": seeds[:5], "step_limits": lens[:5]}, {"ppgs": ppgs[9:10], "seed": 4, "step_limits": [lens[9]]}]
    got = list(pipeline.synthesize_stream(jobs, taco, wg, den, sigma=0.6, strength=0.005, return_device=False, acoustic_workgroups=0))
    assert len(got) == 5 and [t for _, t in got] == [lens] * 3 + [lens[:5], lens[9:10]]
    for k in range(3):
        assert all(np.array_equal(a, b_) for a, b_ in zip(got[k][0], wavs)), "streamed job %d differs" % k
    assert all(np.array_equal(a, b_) for a, b_ in zip(got[3][0], wavs[:5]))
    alone, _ = pipeline.synthesize(ppgs[9:10], taco, wg, den, sigma=0.6, strength=0.005, seed=4, step_limits=[lens[9]])
    assert np.array_equal(got[4][0][0], alone[0])
    serial = list(pipeline.synthesize_stream(jobs[2:4], taco, wg, den, sigma=0.6, strength=0.005, return_device=True, overlap=False))
    assert all(torch.equal(a, torch.from_numpy(b_).cuda()) for a, b_ in zip(serial[1][0], wavs[:5]))
    assert list(pipeline.synthesize_stream([], taco, wg, den)) == []
    # ... and with the decoder held to 32 CUs (the default while overlapped: 2 workgroups per utterance instead of 15) they are
    # those of synthesize() under the same bound, bit for bit, and those of the unbounded decoder to rounding
    monkeypatch.delenv("FACPPG_DECODER_COOP_U")
    bounded = list(pipeline.synthesize_stream(jobs[1:3], taco, wg, den, sigma=0.6, strength=0.005, return_device=False))
    assert taco.decoder_workgroups == 0
    taco.decoder_workgroups = 32
    ref32, t32 = pipeline.synthesize(ppgs, taco, wg, den, sigma=0.6, strength=0.005, utterance_seeds=seeds, step_limits=lens)
    taco.decoder_workgroups = 0
    assert t32 == lens
    for k in range(2):
        assert all(np.array_equal(a, b_) for a, b_ in zip(bounded[k][0], ref32)), "bounded streamed job %d differs" % k
    worst = max(rms(a - b_) for a, b_ in zip(ref32, wavs))
    print("decoder on 32 CUs vs whole chip: worst wav rms difference %.2e" % worst)
    assert worst <= 1e-3
    with torch.no_grad():
        bias = owg.infer(wsd, cfg, torch.zeros(1, 80, 88), 0.0, [torch.zeros(1, 4, 1760), torch.zeros(1, 2, 1760), torch.zeros(1, 2, 1760)])
    oden = dsp.DenoiserOracle(bias)
    for b in sorted(range(16), key=lambda i: lens[i])[:3]:
        n = lens[b]
        enc, dec = taco.draw_dropout_masks([seeds[b]], n, steps=n)             # device layouts
        em = enc.permute(0, 1, 3, 2).cpu().float()                             # -> reference shape [2, 1, Tin, E]
        dm = dec.cpu().float()
        zflat = wg.draw_noise([seeds[b] + 1], n).cpu()
        L = n * 20
        zs = [zflat[:4 * L].view(1, 4, L), zflat[4 * L:6 * L].view(1, 2, L), zflat[6 * L:].view(1, 2, L)]
        x = torch.from_numpy(ppgs[b]).t().unsqueeze(0)
        hp_b = create_hparams_stage(max_decoder_steps=n)
        mel, mel_post, gate, align = otac.inference(tsd, hp_b, x, em, dm)
        assert mel_post.shape[2] == n
        with torch.no_grad():
            ref = oden(owg.infer(wsd, cfg, mel_post, 0.6, zs), 0.005)[0, 0].numpy()
        e = rms(wavs[b] - ref)
        print("config 3 utt %d: %d frames, wav rms err vs oracle %.2e (rms %.3f)" % (b, n, e, rms(ref)))
        assert e <= 1e-3


def test_config4_corpus_script_world1(tmp_path, monkeypatch):
    """BASELINE config 4 on one GPU: script.synthesize_corpus.main() over 64 ragged synthetic PPGs (length law of
    SURVEY 8d config 4: 100 + PCG64(11).integers(0,301), scaled by 1/4 to keep the fixture files small), batches of 16,
    per-utterance decoder limits.  Every wav it writes must equal that utterance's own batch-1 synthesis with the same
    utterance seed, bit for bit, and re-running with another batch size must write identical files."""
    from common.hparams import create_hparams_stage
    from facppg.pipeline import Synthesizer
    from script import synthesize_corpus
    from waveglow.glow import WaveGlow
    monkeypatch.setenv("FACPPG_DECODER_MODE", "coop")
    monkeypatch.setenv("FACPPG_DECODER_COOP_U", "150")   # 2 workgroups per utterance: fits the 32-CU bound of the overlapped stream at B = 16 and runs at B = 1
    monkeypatch.setenv("FACPPG_BILSTM_MODE", "single")
    cfg = dict(synth.WAVEGLOW_CONFIG)
    wg = WaveGlow(**cfg)
    wg.load_state_dict(weightnorm_state_dict(synth.waveglow_state_dict(cfg)), strict=True)
    torch.save({"model": wg, "iteration": 0, "optimizer": None, "learning_rate": 1e-4}, tmp_path / "waveglow.pt")
    hp = create_hparams_stage()
    torch.save({"state_dict": synth.tacotron_state_dict(hp, gate_bias=-10.0), "iteration": 0}, tmp_path / "tacotron.pt")
    lens = _config3_lengths(64, 11, scale=0.25)
    assert len(set(lens)) > 20
    paths = []
    for i, n in enumerate(lens):
        paths.append(str(tmp_path / ("utt%03d.npy" % i)))
        np.save(paths[-1], synth.synthetic_ppg(n, 5816, seed=500 + i, alpha=0.002))
    (tmp_path / "ppgs.txt").write_text("\n".join(paths) + "\n")
    common = ["--ppg2mel_model", str(tmp_path / "tacotron.pt"), "--waveglow_model", str(tmp_path / "waveglow.pt"),
              "--ppg_list", str(tmp_path / "ppgs.txt"), "--seed", "77", "--limit_steps_to_input"]
    written = synthesize_corpus.main(common + ["--output_dir", str(tmp_path / "out16"), "--batch_size", "16"])
    assert written == ["utt%03d.wav" % i for i in range(64)]
    synthesize_corpus.main(common + ["--output_dir", str(tmp_path / "out5"), "--batch_size", "5", "--no_overlap"])
    one = Synthesizer(str(tmp_path / "tacotron.pt"), str(tmp_path / "waveglow.pt"))
    for i, n in enumerate(lens):
        sr, a = wavfile.read(tmp_path / "out16" / ("utt%03d.wav" % i))
        assert sr == 16000 and a.dtype == np.float32 and a.shape == (n * 160,)             # Tout_i = Tin_i, N = Tout*hop
        assert np.array_equal(a, wavfile.read(tmp_path / "out5" / ("utt%03d.wav" % i))[1]), i
        if i % 4 == 0:                                                                    # 16 of the 64 against their own runs
            single, tout = one([np.load(paths[i])], utterance_seeds=[77 + 2 * i], step_limits=[n])
            assert tout == [n] and np.array_equal(single[0], a), i


def test_config4_corpus_1024_utterances_monophone(tmp_path, monkeypatch):
    """BASELINE config 4 at its size on one GPU (world 1): script.synthesize_corpus.main() over 1024 ragged synthetic
    utterances with the full-length law of SURVEY 8d (Tin_i = 100 + PCG64(11).integers(0, 301) frames, 1-4 s), as 40-dim
    monophone PPGs (config "1m": hparams n_symbols = 40, data_utils.py:253-258; keeps the fixture files at ~40 MB),
    batches of 16, per-utterance decoder limits.  Integer facts for every utterance (one wav each, N_i = Tout_i * hop with
    Tout_i = Tin_i), and 32 utterances spread over the length range equal their own batch-1 synthesis bit for bit."""
    from common.hparams import create_hparams_stage
    from facppg.pipeline import Synthesizer
    from script import synthesize_corpus
    from waveglow.glow import WaveGlow
    monkeypatch.setenv("FACPPG_DECODER_MODE", "coop")
    monkeypatch.setenv("FACPPG_DECODER_COOP_U", "150")   # 2 workgroups per utterance: fits the 32-CU bound of the overlapped stream at B = 16 and runs at B = 1
    monkeypatch.setenv("FACPPG_BILSTM_MODE", "single")
    n_utt, n_sym = 1024, 40
    cfg = dict(synth.WAVEGLOW_CONFIG)
    wg = WaveGlow(**cfg)
    wg.load_state_dict(weightnorm_state_dict(synth.waveglow_state_dict(cfg)), strict=True)
    torch.save({"model": wg, "iteration": 0, "optimizer": None, "learning_rate": 1e-4}, tmp_path / "waveglow.pt")
    hp = create_hparams_stage(n_symbols=n_sym)
    torch.save({"state_dict": synth.tacotron_state_dict(hp, gate_bias=-10.0), "iteration": 0}, tmp_path / "tacotron.pt")
    lens = _config3_lengths(n_utt, 11)
    assert min(lens) >= 100 and max(lens) <= 400 and len(set(lens)) > 250
    paths = []
    for i, n in enumerate(lens):
        paths.append(str(tmp_path / ("utt%04d.npy" % i)))
        np.save(paths[-1], synth.synthetic_ppg(n, n_sym, seed=5000 + i, alpha=0.1))
    (tmp_path / "ppgs.txt").write_text("\n".join(paths) + "\n")
    written = synthesize_corpus.main(["--ppg2mel_model", str(tmp_path / "tacotron.pt"), "--waveglow_model", str(tmp_path / "waveglow.pt"),
                                      "--ppg_list", str(tmp_path / "ppgs.txt"), "--seed", "123", "--limit_steps_to_input",
                                      "--hparams", "n_symbols=%d" % n_sym, "--output_dir", str(tmp_path / "out"), "--batch_size", "16"])
    assert written == ["utt%04d.wav" % i for i in range(n_utt)]
    assert sorted(os.listdir(tmp_path / "out")) == written                                   # one wav per utterance, nothing else
    total = 0
    for i, n in enumerate(lens):
        sr, a = wavfile.read(tmp_path / "out" / ("utt%04d.wav" % i))
        assert sr == 16000 and a.dtype == np.float32 and a.shape == (n * 160,), i           # Tout_i = Tin_i, N_i = Tout_i * hop
        assert np.isfinite(a).all() and float(np.abs(a).max()) > 0.0
        total += a.shape[0]
    assert total == 160 * sum(lens)
    one = Synthesizer(str(tmp_path / "tacotron.pt"), str(tmp_path / "waveglow.pt"), hparams=hp)
    order = sorted(range(n_utt), key=lambda i: (lens[i], i))
    for i in order[::n_utt // 32][:32]:                                                      # 32 utterances across the length range
        single, tout = one([np.load(paths[i])], utterance_seeds=[123 + 2 * i], step_limits=[lens[i]])
        a = wavfile.read(tmp_path / "out" / ("utt%04d.wav" % i))[1]
        assert tout == [lens[i]] and np.array_equal(single[0], a), i


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


def test_hop256_whole_path_matches_oracle():
    """The configuration the metric is quoted on -- hop 256 / 22.05 kHz, PPG -> Tacotron2 -> WaveGlow(hop 256) ->
    Denoiser(hop_length=256), exactly the models bench.py's EndToEnd builds (gate bias -10, max_decoder_steps = Tin_i as in
    SURVEY.md 8d config 3) -- against the oracle of the WHOLE path (oracle.tacotron + oracle.waveglow +
    DenoiserOracle(hop_length=256)) on three ragged utterances with injected dropout masks and noise: same number of frames,
    N = Tout * hop exactly, waveform RMS <= 1e-3; the batch equals its single runs bit for bit."""
    from common.hparams import create_hparams_stage
    from facppg import pipeline
    from oracle import dsp, tacotron as otac, waveglow as owg
    from script.train_ppg2mel import load_model
    from waveglow.denoiser import Denoiser
    from waveglow.glow import WaveGlow
    hop = 256
    lens = [31, 18, 26]
    steps = max(lens)
    hp = create_hparams_stage(max_decoder_steps=steps)
    tsd = synth.tacotron_state_dict(hp, gate_bias=-10.0)
    with contextlib.redirect_stdout(io.StringIO()):
        taco = load_model(hp)
    taco.load_state_dict(tsd)
    taco.eval()
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=hop)
    wsd = synth.waveglow_state_dict(cfg)
    wg = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
    wg.load_state_dict(wsd)
    wg = wg.cuda().eval()
    den = Denoiser(wg, hop_length=hop, mode="zeros")          # as bench.py's EndToEnd does
    ppgs = [synth.synthetic_ppg(n, 5816, seed=160 + i, alpha=0.002) for i, n in enumerate(lens)]
    B, Tin, P = len(lens), max(lens), hop // 8
    em = masks_from_seed(15, (2, B, Tin, 600))
    dm = masks_from_seed(16, (steps, 2, B, 300))
    zs = synth.synthetic_z(B, steps * P, cfg, seed=177)
    with contextlib.redirect_stdout(io.StringIO()):          # "Warning! Reached max decoder steps" (model.py:527)
        wavs, tout = pipeline.synthesize(ppgs, taco, wg, den, sigma=0.6, strength=0.005, dropout_masks=(em, dm), z=zs, step_limits=lens)
    assert tout == lens and [len(w) for w in wavs] == [t * hop for t in tout]      # N = Tout * hop exactly
    nb = 88 * P
    with torch.no_grad():
        bias = owg.infer(wsd, cfg, torch.zeros(1, 80, 88), 0.0, [torch.zeros(1, 4, nb), torch.zeros(1, 2, nb), torch.zeros(1, 2, nb)])
    oden = dsp.DenoiserOracle(bias, hop_length=hop)
    assert np.abs(den.bias_spec.cpu().numpy() - oden.bias_spec.numpy()).max() < 1e-3
    for b, n in enumerate(lens):
        emb, dmb = em[:, b:b + 1, :n], dm[:n, :, b:b + 1]
        zb = [z[b:b + 1, :, :n * P].contiguous() for z in zs]
        hp_n = create_hparams_stage(max_decoder_steps=n)
        with contextlib.redirect_stdout(io.StringIO()):
            taco.decoder.max_decoder_steps = n
            single, t1 = pipeline.synthesize([ppgs[b]], taco, wg, den, sigma=0.6, strength=0.005, dropout_masks=(emb, dmb), z=zb)
            taco.decoder.max_decoder_steps = steps
            x = torch.from_numpy(ppgs[b]).t().unsqueeze(0)
            mel, mel_post, gate, align = otac.inference(tsd, hp_n, x, torch.from_numpy(emb.astype(np.float32)),
                                                        torch.from_numpy(dmb.astype(np.float32)))
        assert t1 == [n] and np.array_equal(single[0], wavs[b])              # batch == independent run, bit-exact
        assert mel_post.shape[2] == n
        with torch.no_grad():
            ref = oden(owg.infer(wsd, cfg, mel_post, 0.6, zb), 0.005)[0, 0].numpy()
        e = wavs[b] - ref
        print("hop 256 utt %d: Tout %d, wav rms err %.2e (rms ref %.3f)" % (b, tout[b], rms(e), rms(ref)))
        assert rms(e) <= 1e-3


def test_metric_shape_matches_reference_golden():
    """The headline's own shape against what the REFERENCE produced for it (tests/golden/e2e_metric.npz, written by
    make_golden.py gen_e2e_metric from the imported reference: generate_synthesis.py:74-98 on PPG [200 x 5816], 200 decoder steps,
    hop 256, sigma 0.6, Denoiser(hop_length=256)) -- the models and the call bench.py's EndToEnd times, default launch selection
    (asserted: the split decoder, the streamed path, 32-frame / 8-wave seeded vocoder tiles), injected dropout masks and z: mel <= 1e-4, N = 51 200
    exactly, waveform RMS <= 1e-3 before and after the denoiser."""
    from test_oracle_golden import metric_case
    from facppg import pipeline
    from script.train_ppg2mel import load_model
    from waveglow.denoiser import Denoiser
    from waveglow.glow import WaveGlow
    d, hp, tsd, ppg, em, dm, cfg, zs = metric_case()
    hop, Tout = int(d["hop"]), int(d["Tout"])
    with contextlib.redirect_stdout(io.StringIO()):
        taco = load_model(hp)
    taco.load_state_dict(tsd)
    taco.eval()
    wg = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
    wg.load_state_dict(synth.waveglow_state_dict(cfg))
    wg = wg.cuda().eval()
    den = Denoiser(wg, hop_length=hop, mode="zeros")
    seen = {}
    inference = taco.inference

    def spy(*a, **kw):
        out = inference(*a, **kw)
        seen["mel_post"] = out[1].detach().cpu().numpy()
        seen["streamed"] = kw.get("frame_consumer") is not None and kw["frame_consumer"].active
        return out

    def den_spy(audio, **kw):
        seen["audio"] = audio.detach().clone()
        return den(audio, **kw)
    taco.inference = spy
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            wavs, tout = pipeline.synthesize([ppg], taco, wg, den_spy, sigma=float(d["sigma"]), strength=float(d["strength"]),
                                             dropout_masks=(em, dm), z=zs)
    finally:
        del taco.inference
    assert seen["streamed"]                     # the default batch-1 path: postnet + conditioning seeds under the decoder
    assert taco.last_decoder_launch()[0] == "split"
    assert wg.last_launch_shape()[:2] == (32, 8)
    assert tout == [Tout] and wavs[0].shape == (Tout * hop,) == (51200,)                      # integer: N = Tout * hop
    assert seen["mel_post"].shape == d["mel_post"].shape
    e_mel = np.abs(seen["mel_post"] - d["mel_post"]).max()
    e_pre = rms(seen["audio"].cpu().numpy() - d["audio"])
    e_post = rms(wavs[0] - d["audio_denoised"][0])
    print("metric shape vs reference golden: mel_post %.2e, audio rms err %.2e, denoised rms err %.2e (rms %.3f)" % (
        e_mel, e_pre, e_post, rms(d["audio_denoised"])))
    assert e_mel <= 1e-4 and e_pre <= 1e-3 and e_post <= 1e-3


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
    # --is_fp16 (inference.py:38-48: module and mels cast to half): the reference's fp16 VALUES on this library's fp32 arithmetic --
    # same shapes, close to the fp32 synthesis (fp16 round-off of weights and mels), not identical to it
    out16 = tmp_path / "syn16"
    torch.manual_seed(3)
    inference.main(str(tmp_path / "mels.txt"), str(checkpoints / "waveglow.pt"), 0.6, str(out16), 16000, True)
    for i in range(len(wavs)):
        a = wavfile.read(out / ("m%d_synthesis.wav" % i))[1].astype(np.float64)
        b = wavfile.read(out16 / ("m%d_synthesis.wav" % i))[1].astype(np.float64)
        assert a.shape == b.shape
        err = np.sqrt(np.mean((a - b) ** 2)) / max(1.0, np.sqrt(np.mean(a ** 2)))
        print("is_fp16 vs fp32 synthesis %d: relative rms difference %.2e" % (i, err))
        assert 0 < err < 0.3          # (12 flows amplify the 2^-11 rounding of every weight: ~5 % with the synthetic weights)
