"""GPU: the WaveGlow training loop surface (script.train_waveglow / waveglow.mel2samp): a few Adam
steps on synthetic wavs reduce the loss, and the pickled-module checkpoint round-trips into
load_waveglow_model and synthesises."""
import os

import numpy as np
import pytest
import torch
from scipy.io import wavfile

pytestmark = pytest.mark.gpu


def test_train_loop_checkpoint_and_resume(tmp_path):
    from common.utils import load_waveglow_model
    from facppg import synth
    from script import train_waveglow as tw
    g = np.random.Generator(np.random.PCG64(9))
    files = []
    for i in range(4):
        t = np.arange(16000 + 800 * i) / 16000.0
        wav = 0.3 * np.sin(2 * np.pi * (180 + 40 * i) * t) + 0.02 * g.standard_normal(t.shape)
        path = tmp_path / ("utt%d.wav" % i)
        wavfile.write(path, 16000, (wav * 32767).astype(np.int16))
        files.append(str(path))
    (tmp_path / "files.txt").write_text("\n".join(files) + "\n")
    cfg = dict(synth.WAVEGLOW_CONFIG, n_flows=4)        # 4 flows keep the test short
    data = dict(training_files=str(tmp_path / "files.txt"), segment_length=4000, sampling_rate=16000, filter_length=1024,
                hop_length=160, win_length=1024, mel_fmin=0.0, mel_fmax=8000.0)
    dist_cfg = dict(dist_backend="nccl", dist_url="tcp://127.0.0.1:54321")
    out = tmp_path / "ckpt"
    losses = []
    orig_step = tw.train_step

    def spy(*a, **k):
        losses.append(orig_step(*a, **k))
        return losses[-1]
    tw.train_step = spy
    try:
        tw.train(1, 0, "", str(out), epochs=100, learning_rate=1e-4, sigma=0.7071, iters_per_checkpoint=5, batch_size=2, seed=16807,
                 checkpoint_path="", data_config=data, dist_config=dist_cfg, waveglow_config=cfg, max_iterations=11)
    finally:
        tw.train_step = orig_step
    assert len(losses) == 11 and all(np.isfinite(losses))
    print("losses", ["%.3f" % v for v in losses])
    assert np.mean(losses[-3:]) < np.mean(losses[:3])       # Adam on the HIP gradients makes progress
    assert os.path.isfile(out / "waveglow_10")
    wg = load_waveglow_model(str(out / "waveglow_10"))       # pickled module -> remove_weightnorm -> cuda
    mel = synth.synthetic_mel(1, 10, seed=1).cuda()
    audio = wg.infer(mel, sigma=0.6, seed=1)
    assert audio.shape == (1, 1600) and torch.isfinite(audio).all()
    # resume: iteration counter continues from the checkpoint
    tw.train(1, 0, "", str(out), epochs=100, learning_rate=1e-4, sigma=0.7071, iters_per_checkpoint=1000, batch_size=2, seed=1,
             checkpoint_path=str(out / "waveglow_10"), data_config=data, dist_config=dist_cfg, waveglow_config=cfg,
             max_iterations=13)


def test_train_script_bf16_runs_as_replayed_graph(tmp_path, capsys):
    """script.train_waveglow with train_precision='bf16': the step runs through waveglow.graphed.GraphedTrainStep (three
    ordinary steps, capture, replays), writes a loadable checkpoint in between, resumes from it (the capturable optimizer
    state loads), and walks the same losses as the launch-by-launch loop (hip_graph=False) on the same data order."""
    from common.utils import load_waveglow_model
    from facppg import synth
    from script import train_waveglow as tw
    g = np.random.Generator(np.random.PCG64(11))
    files = []
    for i in range(4):
        t = np.arange(12000 + 400 * i) / 16000.0
        wav = 0.3 * np.sin(2 * np.pi * (150 + 30 * i) * t) + 0.02 * g.standard_normal(t.shape)
        path = tmp_path / ("utt%d.wav" % i)
        wavfile.write(path, 16000, (wav * 32767).astype(np.int16))
        files.append(str(path))
    (tmp_path / "files.txt").write_text("\n".join(files) + "\n")
    cfg = dict(synth.WAVEGLOW_CONFIG, n_flows=4)
    data = dict(training_files=str(tmp_path / "files.txt"), segment_length=4000, sampling_rate=16000, filter_length=1024,
                hop_length=160, win_length=1024, mel_fmin=0.0, mel_fmax=8000.0)
    dist_cfg = dict(dist_backend="nccl", dist_url="tcp://127.0.0.1:54322")

    def run(out, graph, iters, ckpt=""):
        import random
        random.seed(5)                     # Mel2Samp's segment crops
        tw.train(1, 0, "", str(out), epochs=100, learning_rate=1e-4, sigma=0.7071, iters_per_checkpoint=4, batch_size=2, seed=16807,
                 checkpoint_path=ckpt, data_config=data, dist_config=dist_cfg, waveglow_config=cfg, max_iterations=iters,
                 train_precision="bf16", hip_graph=graph)
        lines = [l for l in capsys.readouterr().out.splitlines() if ":\t" in l]
        return [float(l.split("\t")[1]) for l in lines]

    graphed = run(tmp_path / "g", True, 9)
    eager = run(tmp_path / "e", False, 9)
    print("graphed", ["%.4f" % v for v in graphed])
    print("eager  ", ["%.4f" % v for v in eager])
    assert len(graphed) == 9 and all(np.isfinite(graphed))
    assert np.allclose(graphed, eager, rtol=0, atol=5e-4)
    wg = load_waveglow_model(str(tmp_path / "g" / "waveglow_8"))
    audio = wg.infer(synth.synthetic_mel(1, 10, seed=1).cuda(), sigma=0.6, seed=1)
    assert audio.shape == (1, 1600) and torch.isfinite(audio).all()
    resumed = run(tmp_path / "g", True, 12, ckpt=str(tmp_path / "g" / "waveglow_8"))
    assert len(resumed) == 3 and all(np.isfinite(resumed))
