"""GPU parity: Tacotron2.inference (encoder GEMMs + BiLSTM kernel + persistent decoder kernel +
postnet GEMMs) vs the reference's golden vectors with the same injected dropout masks.
Tolerances (north_star): mel <= 1e-4 abs; Tout (stop decision incl. the stopping frame) exact."""
import numpy as np
import pytest
import torch

from helpers import tacotron_case

pytestmark = pytest.mark.gpu


def build(hp, sd):
    from script.train_ppg2mel import load_model
    m = load_model(hp)
    m.load_state_dict(sd, strict=True)
    return m.eval()


@pytest.mark.parametrize("mode", ["split", "coop", "coop20", "coop40", "coop75", "coop150", "single"])
@pytest.mark.parametrize("tag", ["nostop", "stop", "mono40"])
def test_inference_matches_reference_golden(tag, mode, monkeypatch):
    """All decoder launch shapes: 'split' (one attention workgroup + 75 register-resident dense-layer
    workers per utterance, B <= 3), 'coop' (38 / 15 / 8 / 4 / 2 cooperating workgroups per utterance for B <= 6 / 16 / 30 / 60 / 120) and
    'single' (one workgroup per utterance, throughput mode)."""
    if mode.startswith("coop") and len(mode) > 4:     # the wider LSTM slices used for larger batches
        monkeypatch.setenv("FACPPG_DECODER_COOP_U", mode[4:])
        mode = "coop"
    monkeypatch.setenv("FACPPG_DECODER_MODE", mode)
    d, hp, sd, ppg, em, dm = tacotron_case(tag)
    m = build(hp, sd)
    x = torch.from_numpy(ppg).t().unsqueeze(0).cuda()
    mel, mel_post, gate, align = m.inference(x, dropout_masks=(em, dm))
    assert mel.shape == d["mel"].shape, (mel.shape, d["mel"].shape)       # Tout exact
    assert gate.shape == d["gate"].shape and align.shape == d["align"].shape
    e_mem = np.abs(m.last_memory.cpu().numpy() - d["memory"]).max()
    e_mel = np.abs(mel.cpu().numpy() - d["mel"]).max()
    e_post = np.abs(mel_post.cpu().numpy() - d["mel_post"]).max()
    e_al = np.abs(align.cpu().numpy() - d["align"]).max()
    e_gate = np.abs(gate.cpu().numpy() - d["gate"]).max()
    print(tag, mode, "memory %.2e mel %.2e mel_post %.2e align %.2e gate %.2e" % (e_mem, e_mel, e_post, e_al, e_gate))
    assert e_mem <= 1e-4 and e_mel <= 1e-4 and e_post <= 1e-4 and e_al <= 1e-4 and e_gate <= 1e-4
    # a17, integer: the masked-out pattern (the reference's masked_fill(-inf) -> softmax == exactly 0) must be the
    # golden's entry for entry -- a window off by one would pass the float check whenever the edge weight is < 1e-4
    assert np.array_equal(align.cpu().numpy() == 0, d["align"] == 0)


def test_attention_window_mask_hip_bit_exact():
    """a17 on the device: facppg_attention_window_mask (the decoder kernels' own attn_window_range) over all 144
    reference masks of tests/golden/attn_masks.npz -- get_mask_from_lengths_window_and_time_step run on the imported
    reference (utils.py:46-78), incl. len 1, ragged batches and steps far past the end (the 'last frame stays
    unmasked' quirk, utils.py:65-69)."""
    import json
    from common.utils import get_mask_from_lengths_window_and_time_step
    from helpers import golden
    d = golden("attn_masks.npz")
    cases = json.loads(bytes(d["cases"]).decode())
    assert len(cases) == 144
    quirk = 0
    for c in cases:
        lengths = torch.tensor(c["lengths"], dtype=torch.int64, device="cuda")
        m = get_mask_from_lengths_window_and_time_step(lengths, c["W"], c["t"])
        assert m.is_cuda and m.dtype == torch.bool
        got = m.cpu().numpy().astype(np.uint8)
        assert got.shape == d[c["key"]].shape and np.array_equal(got, d[c["key"]]), c
        quirk += any(c["t"] - c["W"] > n - 1 for n in c["lengths"])
    assert quirk > 0                      # the fixture does exercise the quirk


@pytest.mark.parametrize("mode", ["split", "coop", "single"])
def test_alignment_support_is_the_reference_window_past_the_end(mode, monkeypatch):
    """a17 through the decoder itself where the quirk bites: Tin = 6 and 40 steps, so from step 26 on t-W has
    passed the last frame and ONLY frame Tin-1 may carry weight (utils.py:65-69).  The non-zero pattern of the
    device alignments must equal the reference mask for every step, and equal the oracle's pattern."""
    from common.hparams import create_hparams_stage
    from common.utils import get_mask_from_lengths_window_and_time_step
    from facppg import synth
    from helpers import masks_from_seed
    from oracle import tacotron as otac
    monkeypatch.setenv("FACPPG_DECODER_MODE", mode)
    Tin, steps, W = 6, 40, 20
    hp = create_hparams_stage(max_decoder_steps=steps)
    sd = synth.tacotron_state_dict(hp, gate_bias=-10.0)
    m = build(hp, sd)
    ppg = synth.synthetic_ppg(Tin, 5816, seed=2)
    em, dm = masks_from_seed(1, (2, 1, Tin, 600)), masks_from_seed(2, (steps, 2, 1, 300))
    x = torch.from_numpy(ppg).t().unsqueeze(0)
    _, _, _, align = m.inference(x.cuda(), dropout_masks=(em, dm))
    al = align[0].cpu().numpy()
    ref = otac.inference(sd, hp, x, torch.from_numpy(em.astype(np.float32)), torch.from_numpy(dm.astype(np.float32)))[3][0].numpy()
    assert al.shape == ref.shape == (steps, Tin)
    for t in range(steps):
        keep = ~get_mask_from_lengths_window_and_time_step([Tin], W, t)[0].numpy()
        assert np.array_equal(al[t] != 0, keep), (t, al[t])
        assert np.array_equal(ref[t] != 0, keep)
    assert np.array_equal(al[30] != 0, np.arange(Tin) == Tin - 1) and abs(al[30, Tin - 1] - 1.0) < 1e-6


def test_bilstm_shapes_agree(monkeypatch):
    """Encoder BiLSTM: the register-resident cooperative kernel (B <= 12) and the one-workgroup kernel give
    the same memory up to the fp32 re-association of the K split, on a ragged batch."""
    d, hp, sd, ppg, em, dm = tacotron_case("stop")
    m = build(hp, sd)
    from facppg import synth
    lens = [24, 9, 17]
    x = torch.zeros(len(lens), ppg.shape[1], max(lens))
    for b, n in enumerate(lens):
        x[b, :, :n] = torch.from_numpy(synth.synthetic_ppg(n, ppg.shape[1], seed=40 + b)).t()
    g = np.random.Generator(np.random.PCG64(5))
    emb = (g.random((2, len(lens), max(lens), 600)) < 0.5).astype(np.uint8)
    dmb = (g.random((int(d["max_steps"]), 2, len(lens), 300)) < 0.5).astype(np.uint8)
    mems = []
    for mode in ("coop", "wide", "single"):     # 32-unit slices, 64-unit slices, one workgroup
        monkeypatch.setenv("FACPPG_BILSTM_MODE", mode)
        m.inference(x.cuda(), lengths=lens, dropout_masks=(emb, dmb))
        mems.append(m.last_memory.clone())
    err = max((mems[0] - mems[2]).abs().max().item(), (mems[1] - mems[2]).abs().max().item())
    print("bilstm coop / wide vs single: max abs diff %.2e" % err)
    assert err <= 1e-5
    for b, n in enumerate(lens):
        assert torch.count_nonzero(mems[0][b, n:]) == 0


def test_get_inference_surface_and_clip():
    from common.utils import get_inference
    d, hp, sd, ppg, em, dm = tacotron_case("mono40")
    m = build(hp, sd)
    torch.manual_seed(0)
    out = get_inference(ppg, m)                     # device-drawn dropout: shape/finite only
    assert out.shape == (1, 80, 16) and torch.isfinite(out).all()
    Tin = ppg.shape[0]
    clipped = get_inference(ppg, m, is_clip=True)   # [10 : Tin-10] on the OUTPUT axis (utils.py:171-172)
    assert clipped.shape[2] == max(0, min(16, Tin - 10) - 10)


@pytest.mark.parametrize("lens", [[24, 9, 17], [24, 9, 17, 13, 20], [11, 24, 9, 17, 13, 20, 22, 15]])
@pytest.mark.parametrize("mode", ["split", "coop", "single"])
def test_padded_batch_equals_independent_runs(mode, lens, monkeypatch):
    """Batched semantics the reference never defined (batch-1 only): identical to B independent
    batch-1 runs, including each utterance's own stop step.  In split mode 3 / 5 / 8 utterances share
    each set of dense-layer workers 1 / 2 / 3 at a time (the last set is partly empty)."""
    monkeypatch.setenv("FACPPG_DECODER_MODE", mode)
    if mode == "coop":      # the slice width follows B; bit-equality is a property of one width (the sums are cut differently)
        monkeypatch.setenv("FACPPG_DECODER_COOP_U", "20")
    d, hp, sd, ppg, em, dm = tacotron_case("stop")
    m = build(hp, sd)
    B, Tin, steps = len(lens), max(lens), int(d["max_steps"])
    g = np.random.Generator(np.random.PCG64(21))
    x = torch.zeros(B, ppg.shape[1], Tin)
    from facppg import synth
    for b, n in enumerate(lens):
        x[b, :, :n] = torch.from_numpy(synth.synthetic_ppg(n, ppg.shape[1], seed=40 + b)).t()
    emb = (g.random((2, B, Tin, 600)) < 0.5).astype(np.uint8)
    dmb = (g.random((steps, 2, B, 300)) < 0.5).astype(np.uint8)
    mel, mel_post, gate, align = m.inference(x.cuda(), lengths=lens, dropout_masks=(emb, dmb))
    out_lens = m.last_output_lengths.tolist()
    for b, n in enumerate(lens):
        sm, sp, sg, sa = m.inference(x[b:b + 1, :, :n].contiguous().cuda(),
                                     dropout_masks=(emb[:, b:b + 1, :n], dmb[:, :, b:b + 1]))
        To = sm.shape[2]
        assert To == out_lens[b]
        assert torch.equal(sm[0], mel[b, :, :To]) and torch.equal(sp[0], mel_post[b, :, :To])
        assert torch.equal(sa[0], align[b, :To, :n])
        assert torch.count_nonzero(mel_post[b, :, To:]) == 0


def test_large_batch_runs_in_chunks_and_matches_single_runs():
    """B = 130 exceeds what one cooperative launch can keep co-resident (120 utterances at 2 workgroups
    each): the decoder runs it in chunks.  Spot-check utterances from both chunks against their own
    batch-1 runs (other launch shape: equal to fp32 round-off, stop step exact)."""
    d, hp, sd, ppg, em, dm = tacotron_case("stop")
    m = build(hp, sd)
    from facppg import synth
    g = np.random.Generator(np.random.PCG64(9))
    B, steps = 130, int(d["max_steps"])
    lens = (8 + g.integers(0, 17, size=B)).tolist()
    Tin = max(lens)
    x = torch.zeros(B, ppg.shape[1], Tin)
    for b, n in enumerate(lens):
        x[b, :, :n] = torch.from_numpy(synth.synthetic_ppg(n, ppg.shape[1], seed=100 + b)).t()
    emb = (g.random((2, B, Tin, 600)) < 0.5).astype(np.uint8)
    dmb = (g.random((steps, 2, B, 300)) < 0.5).astype(np.uint8)
    mel, mel_post, gate, align = m.inference(x.cuda(), lengths=lens, dropout_masks=(emb, dmb))
    out_lens = m.last_output_lengths.tolist()
    assert all(1 <= t <= steps for t in out_lens)
    for b in (0, 57, 119, 120, 129):
        n = lens[b]
        sm, sp, sg, sa = m.inference(x[b:b + 1, :, :n].contiguous().cuda(), dropout_masks=(emb[:, b:b + 1, :n], dmb[:, :, b:b + 1]))
        To = sm.shape[2]
        assert To == out_lens[b]
        assert (sp[0] - mel_post[b, :, :To]).abs().max().item() <= 1e-4
        assert torch.count_nonzero(mel_post[b, :, To:]) == 0


