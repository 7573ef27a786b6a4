"""CPU: the PPG front-end's blob-free part -- Kaldi binary readers against the reference's data files and
the oracle restatement of MFCC -> CMN -> splice -> LDA / resampling / PPG reduction against the KNOWN ANSWERS the
reference's own tests hold for it (test/test_feat.py:55-87, test/test_ppg.py:36-64)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import feat as of

KF = os.path.join(GOLDEN, "kaldi_feats")


def synthetic_wav(n, fs, seed=0):
    g = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n) / fs
    x = 6000 * np.sin(2 * np.pi * 220 * t) + 2500 * np.sin(2 * np.pi * 1370 * t + 1.0) + 300 * g.standard_normal(n)
    return np.round(x).astype(np.int16)


def test_kaldi_readers_on_the_reference_data_files():
    from common import kaldi_io
    lda = kaldi_io.read_matrix(os.path.join(KF, "final.mat"))
    assert lda.shape == (40, 91) and lda.dtype == np.float32 and np.isfinite(lda).all()          # 91 = 13 x (3 + 1 + 3)
    red = kaldi_io.read_sparse_matrix(os.path.join(KF, "reduce_dim.mat"))
    assert red.shape == (40, 5816)
    assert red.sum() == 5816                                   # test_feat.py:85-87 "This is a special matrix"
    assert (np.count_nonzero(red, axis=0) == 1).all()          # every pdf belongs to exactly one monophone
    with pytest.raises(kaldi_io.KaldiFormatError):
        kaldi_io.read_matrix(os.path.join(KF, "reduce_dim.mat"))
    with pytest.raises(kaldi_io.KaldiFormatError):
        kaldi_io.read_matrix(os.path.join(KF, "splice_opts"))


@pytest.mark.parametrize("n", [16000, 51200, 12345, 400, 81])
def test_oracle_mfcc_known_answers(n):
    from common import kaldi_io
    wav = synthetic_wav(n, 16000).astype(np.float32)
    m = of.mfcc(wav, 16000.0)
    assert m.shape == (int(round(n / 160.0 + 1e-9)) if n % 160 != 80 else (n + 80) // 160, 13)      # test_feat.py:59-64
    assert m.shape[0] == (n + 80) // 160
    c = of.cmn(m.astype(np.float64))
    assert abs(c.sum()) < 1e-2                                                                      # test_feat.py:66-72
    lda = kaldi_io.read_matrix(os.path.join(KF, "final.mat"))
    f = of.feat_for_nnet(wav, lda)
    assert f.shape == (m.shape[0], 40)                                                              # test_ppg.py:36-46
    assert np.isfinite(f).all()
    s = of.splice(m, 3, 3)
    assert s.shape == (m.shape[0], 91) and np.array_equal(s[0, :13], m[0]) and np.array_equal(s[0, 39:52], m[0])
    assert np.array_equal(s[-1, -13:], m[-1])


def test_oracle_resample_and_reduction():
    from common import kaldi_io
    fs = 44100
    t = np.arange(44100) / fs
    y = of.resample(10000 * np.sin(2 * np.pi * 1000 * t), fs, 16000)
    assert y.shape == (16000,)                                                       # 141120 samples @44.1k -> 51200: test_feat.py:59-64
    tt = np.arange(16000) / 16000.0
    assert np.abs(y[200:-200] - 10000 * np.sin(2 * np.pi * 1000 * tt)[200:-200]).max() < 60      # pass band: within 0.6 %
    z = of.resample(10000 * np.sin(2 * np.pi * 10000 * t), fs, 16000)
    assert np.abs(z[200:-200]).max() < 100                                           # above the 7.92 kHz cutoff: gone
    assert of.resample(np.ones(141120), 44100, 16000).shape == (51200,)
    red = kaldi_io.read_sparse_matrix(os.path.join(KF, "reduce_dim.mat"))
    g = np.random.Generator(np.random.PCG64(1))
    ppg = g.dirichlet(np.full(5816, 0.01), size=7).astype(np.float32)
    mono = of.reduce_ppg(ppg, red)
    assert mono.shape == (7, 40) and np.allclose(mono.sum(1), 1.0, atol=1e-5)       # test_ppg.py:56-64: mass is kept


def test_dependencies_ppg_loads_what_the_reference_ships(monkeypatch):
    import ppg
    deps = ppg.DependenciesPPG(nnet_path=os.path.join(KF, "final.raw"), lda_path=os.path.join(KF, "final.mat"),
                               reduce_dim_path=os.path.join(KF, "reduce_dim.mat"), splice_opts_path=os.path.join(KF, "splice_opts"))
    assert deps is not None and deps.nnet is None and deps.precomputed_only                 # test_ppg.py:75-77 (the blob is absent)
    assert tuple(deps.lda.shape) == (40, 91) and tuple(deps.monophone_trans.shape) == (40, 5816)
    assert (deps.left_context, deps.right_context) == ("3", "3") and deps.splice_opts.startswith("--left-context=3")
