"""CPU: the Kaldi nnet3 raw-model reader / writer (common/nnet3.py), the TDNN computation plan the HIP path runs
(plan_layers: batch-norm and fixed-affine folding, splice grids) and the NumPy oracle (oracle/nnet3.py) -- f4, the acoustic
model of the PPG front-end (reference: src/ppg/compute_ppg.py:42-70, src/common/decode.py:23-38).

PARITY UNPINNED: the reference ships neither the model (data/am/final.raw is a missing blob) nor any Kaldi output, and
pykaldi is absent, so the models here come from this build's own writer.  What IS asserted from the reference: its tests'
known answers (test/test_decode.py:19-28: the model loads, input dim 40; test/test_ppg.py:48-73: one PPG row per frame,
dim = number of senones, rows sum to 1, monophone reduction to 40 dims keeps the mass)."""
import numpy as np
import pytest

from common import decode, nnet3
from oracle import nnet3 as onnet3


def _plan_forward(layers, final, feats):
    """The fused-layer chain as the HIP path evaluates it (frame-shift bookkeeping of csrc/facppg_tdnn.hip), in fp64."""
    T = feats.shape[0]
    L = sum(max(-l["first"], 0) for l in layers)
    R = sum(max(l["first"] + (l["taps"] - 1) * l["dil"], 0) for l in layers)
    Tp = L + T + R
    x = feats[np.clip(np.arange(Tp) - L, 0, T - 1)].astype(np.float64).T          # [C][Tp]
    for l in layers:
        cin = x.shape[0]
        cols = np.zeros((l["taps"] * cin, Tp))
        for j in range(l["taps"]):
            sh = j * l["dil"]
            cols[j * cin:(j + 1) * cin, :Tp - sh] = x[:, sh:]                        # reads past Tp are zero
        y = l["W"] @ cols + l["b"][:, None]
        if l["act"] == "relu":
            y = np.maximum(y, 0)
        if l["renorm"]:
            ss = np.maximum((y ** 2).sum(0, keepdims=True) / (y.shape[0] * l["renorm"] ** 2), 2.0 ** -66)
            y = y / np.sqrt(ss)
        x = y
    y = x[:, :T].T
    if final != "none":
        z = y - y.max(1, keepdims=True)
        lse = np.log(np.exp(z).sum(1, keepdims=True))
        y = np.exp(z - lse) if final == "softmax" else z - lse
    return y


@pytest.mark.parametrize("binary", [True, False])
@pytest.mark.parametrize("norm,output,lda", [("batchnorm", "softmax", True), ("renorm", "log-softmax", False)])
def test_reader_round_trips_writer_and_known_answers(tmp_path, binary, norm, output, lda):
    net = nnet3.synthetic_tdnn(input_dim=40, hidden=48, output_dim=72, norm=norm, output=output, lda=lda, seed=3)
    path = str(tmp_path / "final.raw")
    nnet3.write_nnet3(path, net, binary=binary)
    assert open(path, "rb").read(2) == (b"\x00B" if binary else b"<N")
    back = decode.read_nnet3_model(path)                                          # decode.py:23-38
    assert back.input_dim("input") == 40 and back.output_dim() == 72               # test_decode.py:19-28: input dim 40
    assert [n["name"] for n in back.nodes] == [n["name"] for n in net.nodes]
    for name, c in net.components.items():
        b = back.components[name]
        assert b.type == c.type
        for k, v in c.fields.items():
            if isinstance(v, np.ndarray):
                assert np.allclose(b.fields[k], v, rtol=0, atol=0 if binary else 1e-6), (name, k)
            elif v is not None:
                assert b.fields[k] == pytest.approx(v), (name, k)
    # context of the splices (-2..2), (-1, 2), (-3, 3), (0): 6 left, 7 right
    assert back.context() == (6, 7)
    g = np.random.Generator(np.random.PCG64(1))
    feats = g.standard_normal((23, 40)).astype(np.float32)
    ppg = onnet3.forward(back, feats)
    assert ppg.shape == (23, 72)                                                   # one row per frame, dim = senones
    rows = ppg.sum(1) if output == "softmax" else np.exp(ppg).sum(1)
    assert np.allclose(rows, 1.0, atol=1e-5)                                       # test_ppg.py:54 rows are posteriors
    assert np.array_equal(onnet3.forward(net, feats), ppg)


@pytest.mark.parametrize("norm,lda", [("batchnorm", True), ("batchnorm", False), ("renorm", True)])
def test_plan_layers_equals_the_unfolded_network(norm, lda):
    """The fused chain (BatchNorm folded into the next affine, LDA multiplied into the first one, splices on a uniform grid,
    shifted columns) is the same function as the component-by-component oracle, edges included."""
    net = nnet3.synthetic_tdnn(input_dim=13, hidden=32, output_dim=50, norm=norm, output="softmax", lda=lda, seed=5,
                               splices=((-2, -1, 0, 1, 2), (-1, 2), (-3, 0, 3), (-7, 2), (0,)))
    layers, final = nnet3.plan_layers(net)
    assert final == "softmax" and [l["taps"] for l in layers] == [5, 2, 3, 2, 1, 1] and [l["dil"] for l in layers] == [1, 3, 3, 9, 1, 1]
    assert all(l["act"] == "relu" for l in layers[:-1]) and layers[-1]["act"] == "none"
    assert net.context() == (13, 9)
    g = np.random.Generator(np.random.PCG64(2))
    for T in (1, 4, 31):                                                           # shorter than the context, too
        feats = g.standard_normal((T, 13)).astype(np.float32)
        a, b = _plan_forward(layers, final, feats), onnet3.forward(net, feats)
        assert a.shape == b.shape == (T, 50)
        assert np.abs(a - b).max() <= 2e-6, (T, np.abs(a - b).max())


def test_non_uniform_splice_and_unsupported_graphs():
    net = nnet3.synthetic_tdnn(input_dim=8, hidden=16, output_dim=10, lda=False, splices=((-4, -1, 0, 2),), seed=7)
    layers, final = nnet3.plan_layers(net)
    assert layers[0]["taps"] == 7 and layers[0]["dil"] == 1 and layers[0]["first"] == -4      # embedded in the grid -4..2, zeros between
    feats = np.random.Generator(np.random.PCG64(3)).standard_normal((9, 8)).astype(np.float32)
    assert np.abs(_plan_forward(layers, final, feats) - onnet3.forward(net, feats)).max() <= 2e-6
    # a Sum of two nodes is evaluated by the oracle, refused by the HIP plan
    net.nodes.insert(-1, {"kind": "component", "name": "extra", "component": "tdnn1.relu",
                          "input": ("Sum", ("node", "tdnn1.affine"), ("node", "tdnn1.affine"))})
    net.by_name["extra"] = net.nodes[-2]
    net.by_name["output.affine"]["input"] = ("node", "extra")
    with pytest.raises(nnet3.Nnet3FormatError):
        nnet3.plan_layers(net)
    assert nnet3.parse_descriptor("Append(Offset(a.b, -1), a.b, Offset(a.b, 1))") == \
        ("Append", [("Offset", ("node", "a.b"), -1), ("node", "a.b"), ("Offset", ("node", "a.b"), 1)])
    with pytest.raises(nnet3.Nnet3FormatError):
        nnet3.parse_descriptor("ReplaceIndex(a, t, 0)")


def test_bad_files_fail_loudly(tmp_path):
    p = tmp_path / "x.raw"
    p.write_bytes(b"\x00B<Nnet2> ")
    with pytest.raises(nnet3.Nnet3FormatError):
        nnet3.read_nnet3(str(p))
    net = nnet3.synthetic_tdnn(input_dim=4, hidden=8, output_dim=6, splices=((0,),), lda=False)
    nnet3.write_nnet3(str(p), net)
    raw = p.read_bytes()
    p.write_bytes(raw[:len(raw) // 2])
    with pytest.raises((nnet3.Nnet3FormatError, ValueError, IndexError)):
        nnet3.read_nnet3(str(p))
