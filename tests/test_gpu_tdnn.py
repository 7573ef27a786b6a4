"""GPU: nnet3 TDNN acoustic-model inference on the HIP kernels (facppg_tdnn_*, csrc/facppg_tdnn.hip) through
ppg.compute_full_ppg vs the NumPy oracle (oracle/nnet3.py), on models written by common.nnet3's own writer and read back
by common.decode.read_nnet3_model -- f4 of SURVEY.md 8(f); reference: src/ppg/compute_ppg.py:42-70, 161-202.

PARITY UNPINNED at the Kaldi boundary (no model file ships with the reference, no pykaldi): the oracle restates Kaldi's
published component semantics.  Asserted from the reference's own tests (test/test_ppg.py:48-73): one row per frame,
5816 senones, rows are posteriors summing to 1, the 40-dim monophone reduction keeps the mass.
Tolerance: posteriors 2e-6 absolute (fp32 products summed in another order; batch-norm folded in fp64)."""
import os

import numpy as np
import pytest
import torch
from scipy.io import wavfile

from oracle import nnet3 as onnet3
from test_feat_cpu import KF, synthetic_wav

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("norm,output,lda,T", [("batchnorm", "softmax", True, 150), ("renorm", "log-softmax", False, 67), ("batchnorm", "softmax", False, 3)])
def test_full_ppg_matches_oracle_and_reference_known_answers(tmp_path, norm, output, lda, T):
    import ppg
    from common import decode, feat, nnet3
    net = nnet3.synthetic_tdnn(input_dim=40, hidden=256, output_dim=5816, norm=norm, output=output, lda=lda, seed=11)
    path = str(tmp_path / "final.raw")
    nnet3.write_nnet3(path, net)
    model = decode.read_nnet3_model(path)
    assert model.input_dim() == 40 and model.output_dim() == 5816                          # test_decode.py:19-28
    g = np.random.Generator(np.random.PCG64(T))
    feats = (2.0 * g.standard_normal((T, 40))).astype(np.float32)
    out = ppg.compute_full_ppg(model, torch.from_numpy(feats).cuda())
    assert out.is_cuda and tuple(out.shape) == (T, 5816)                                    # test_ppg.py:48-54
    ref = onnet3.forward(model, feats)
    got = out.cpu().numpy()
    post, post_ref = (got, ref) if output == "softmax" else (np.exp(got), np.exp(ref))
    print("T=%d %s/%s: posterior max err %.2e (max posterior %.3f), row-sum err %.1e" % (
        T, norm, output, np.abs(post - post_ref).max(), post_ref.max(), np.abs(post.sum(1) - 1).max()))
    assert np.abs(post - post_ref).max() <= 2e-6
    if output != "softmax":
        assert np.abs(got - ref).max() <= 2e-4                                               # log-posteriors of magnitude ~10
    assert np.abs(post.sum(1) - 1.0).max() <= 1e-4                                          # rows are posteriors
    assert np.array_equal(post.argmax(1), post_ref.argmax(1))
    if output == "softmax":
        red = feat.read_sparse_mat(os.path.join(KF, "reduce_dim.mat"))                     # the reference's own 40 x 5816 map
        mono = ppg.reduce_ppg_dim(out, red)
        assert tuple(mono.shape) == (T, 40) and abs(float(mono.sum()) - T) <= 1e-2        # test_ppg.py:56-73
    again = ppg.compute_full_ppg(model, feats)                                             # numpy in, cached handle, same bits
    assert torch.equal(again, out)
    with pytest.raises(Exception, match="features have"):
        ppg.compute_full_ppg(model, np.zeros((5, 39), np.float32))


def test_wav_to_monophone_ppg_end_to_end(tmp_path):
    """compute_monophone_ppg / DependenciesPPG / get_ppg with a model file present: wav -> MFCC -> CMN -> splice -> LDA ->
    TDNN -> 5816 posteriors -> 40 monophones, every stage on the HIP kernels; vs the oracles of the two halves."""
    import ppg
    from common import data_utils, feat, nnet3
    from oracle import feat as of
    net = nnet3.synthetic_tdnn(input_dim=40, hidden=128, output_dim=5816, seed=4)
    nnet_path = str(tmp_path / "final.raw")
    nnet3.write_nnet3(nnet_path, net)
    deps = ppg.DependenciesPPG(nnet_path=nnet_path, lda_path=os.path.join(KF, "final.mat"),
                               reduce_dim_path=os.path.join(KF, "reduce_dim.mat"), splice_opts_path=os.path.join(KF, "splice_opts"))
    assert deps.nnet is not None and not deps.precomputed_only and (deps.left_context, deps.right_context) == ("3", "3")
    wav = synthetic_wav(24000, 16000, seed=9)
    path = str(tmp_path / "utt.wav")
    wavfile.write(path, 16000, wav)
    full = data_utils.get_ppg(path, deps)                                                  # data_utils.py:55-59
    assert isinstance(full, np.ndarray) and full.shape == (150, 5816)
    ref_feats = of.feat_for_nnet(wav.astype(np.float32), deps.lda.numpy(), samp_freq=16000.0)
    ref = onnet3.forward(deps.nnet, ref_feats)
    assert np.abs(full - ref).max() <= 5e-5 and np.abs(full.sum(1) - 1).max() <= 1e-4
    mono = ppg.compute_monophone_ppg(feat.read_wav_kaldi(path), deps.nnet, deps.lda, deps.monophone_trans)
    assert mono.shape == (150, 40) and abs(float(mono.sum()) - 150) <= 1e-2
