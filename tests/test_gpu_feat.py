"""GPU: the PPG front-end's blob-free part on the HIP kernels (facppg_resample, facppg_mfcc_*, facppg_cmn_splice_transform,
facppg_reduce_ppg) vs the CPU oracle (oracle/feat.py) and the known answers the reference's tests hold (test/test_feat.py,
test/test_ppg.py).  Tolerances: MFCC 1e-3 absolute on log-domain values of magnitude up to ~95 (measured 6e-5), nnet input features 1e-4
(measured 9e-6), reductions 1e-5."""
import os

import numpy as np
import pytest
import torch
from scipy.io import wavfile

from conftest import GOLDEN
from oracle import feat as of
from test_feat_cpu import KF, synthetic_wav

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,fs", [(51200, 16000), (12345, 16000), (480, 16000), (141120, 44100)])
def test_feat_for_nnet_matches_oracle_and_reference_known_answers(n, fs, tmp_path):
    import ppg
    from common import feat, kaldi_io
    wav = synthetic_wav(n, fs, seed=n)
    path = str(tmp_path / "utt.wav")
    wavfile.write(path, fs, np.stack([wav, wav[::-1]], 1) if n == 12345 else wav)            # one dual-channel case: first channel kept
    wd = feat.read_wav_kaldi(path)
    assert wd.data().shape == (1, n) and wd.samp_freq == fs                                   # test_feat.py:39-53
    opts = feat.MfccOptions()
    opts.frame_opts.allow_downsample = True
    opts.frame_opts.snip_edges = False
    opts.use_energy = False
    mf = feat.compute_mfcc_feats(wd, opts)
    T = int(round(n / (fs * 0.01)))                                                           # test_feat.py:59-64
    assert mf.is_cuda and tuple(mf.shape) == (T, 13)
    ref_wav = wav.astype(np.float32) if fs == 16000 else of.resample(wav.astype(np.float32), fs, 16000)
    ref = of.mfcc(ref_wav, 16000.0)
    e_m = np.abs(mf.cpu().numpy() - ref).max()
    cm = feat.apply_cepstral_mean_norm(mf)
    assert abs(float(cm.sum())) < 5e-2                                                        # test_feat.py:66-72 (assertAlmostEqual(.., 0, 2) at T*13 terms)
    lda = kaldi_io.read_matrix(os.path.join(KF, "final.mat"))
    sp = feat.splice_frames(mf, 3, 3)
    assert tuple(sp.shape) == (T, 91) and np.allclose(sp.cpu().numpy(), of.splice(mf.cpu().numpy(), 3, 3))
    tl = feat.apply_feat_transform(sp, torch.from_numpy(lda))
    assert tuple(tl.shape) == (T, 40)                                                         # test_feat.py:74-83
    out = ppg.compute_feat_for_nnet(path, os.path.join(KF, "final.mat"))
    assert tuple(out.shape) == (T, 40)                                                        # test_ppg.py:32-46
    ref_f = of.feat_for_nnet(wav.astype(np.float32), lda, samp_freq=float(fs))
    e_f = np.abs(out.cpu().numpy() - ref_f).max()
    print("n=%d fs=%d: T=%d, MFCC max err %.2e (|ref| max %.1f), nnet-input max err %.2e (|ref| max %.2f)" % (
        n, fs, T, e_m, np.abs(ref).max(), e_f, np.abs(ref_f).max()))
    assert e_m <= 1e-3 and e_f <= 1e-4
    # energy variant: C0 = log frame energy after DC removal
    opts.use_energy = True
    me = feat.compute_mfcc_feats(wd, opts).cpu().numpy()
    assert np.abs(me - of.mfcc(ref_wav, 16000.0, use_energy=True)).max() <= 1e-3
    with pytest.raises(Exception, match="bad dimension"):
        feat.apply_feat_transform(mf, torch.zeros(40, 91))


def test_resample_and_reduce_ppg_match_oracle():
    import ppg
    from common import feat
    from facppg import lib as flib
    L = flib.load()
    wav = synthetic_wav(44100, 44100, seed=5).astype(np.float32)
    x = torch.from_numpy(wav).cuda()
    n_out = L.facppg_resample_num_samples(x.numel(), 44100, 16000)
    assert n_out == 16000 and L.facppg_resample_num_samples(141120, 44100, 16000) == 51200
    y = torch.empty(n_out, device="cuda")
    flib.check(L.facppg_resample(flib.ptr(x), x.numel(), 44100, 16000, flib.ptr(y), flib.current_stream(x.device)))
    assert np.abs(y.cpu().numpy() - of.resample(wav, 44100, 16000)).max() <= 2e-2          # int16 scale: ~1e-6 relative
    red = feat.read_sparse_mat(os.path.join(KF, "reduce_dim.mat"))
    g = np.random.Generator(np.random.PCG64(2))
    full = g.dirichlet(np.full(5816, 0.01), size=203).astype(np.float32)
    mono = ppg.reduce_ppg_dim(full, red)
    assert mono.is_cuda and tuple(mono.shape) == (203, 40)                                    # test_ppg.py:56-64
    assert abs(float(mono.sum()) - 203) < 1e-2
    assert np.abs(mono.cpu().numpy() - of.reduce_ppg(full, red.numpy())).max() <= 1e-5
    with pytest.raises(flib.FacppgError, match="acoustic model"):
        ppg.compute_full_ppg(None, mono)
    deps = ppg.DependenciesPPG(nnet_path=os.path.join(KF, "final.raw"), lda_path=os.path.join(KF, "final.mat"),
                               reduce_dim_path=os.path.join(KF, "reduce_dim.mat"), splice_opts_path=os.path.join(KF, "splice_opts"))
    f = ppg.compute_feat_for_nnet_internal(feat.read_wav_kaldi_internal(synthetic_wav(8000, 16000), 16000), deps.lda, frame_shift=10)
    assert tuple(f.shape) == (50, 40)
