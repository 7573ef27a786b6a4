"""Oracle (TEST INFRASTRUCTURE ONLY): Tacotron2-style PPG->mel inference on the CPU.

Restates (batch = 1, as the reference's inference is, model.py:524-528):
  * Prenet (dropout p=0.5 ALWAYS on)      src/common/model.py:124-135
  * Encoder.inference                      model.py:237-249
  * LocationLayer / Attention              model.py:44-121
  * Decoder.inference / decode             model.py:489-535, 387-442, 304-335
  * Postnet                                model.py:138-184
  * Tacotron2.inference                    model.py:597-610
  * get_mask_from_lengths_window_and_time_step  src/common/utils.py:46-78
  * get_inference                          utils.py:155-174
The reference draws the prenet dropout masks from torch's RNG; here they are INPUTS
(``enc_masks`` [2][1,Tin,E], ``dec_masks`` [steps][2][1,P], values in {0,1}; kept value is
scaled by 1/(1-p) = 2 exactly as F.dropout does).
"""
import numpy as np
import torch
import torch.nn.functional as F


def window_mask(lengths, window, t):
    """utils.py:46-78: True = masked.  Keeps [min(max(0,t-W), len-1), min(t+W, len-1)]."""
    lengths = [int(x) for x in lengths]
    max_len = max(lengths)
    mask = np.ones((len(lengths), max_len), dtype=bool)
    for i, n in enumerate(lengths):
        max_idx = n - 1
        start = min(max(0, t - window), max_idx)
        end = min(t + window, max_idx)
        if start > end:
            continue
        mask[i, start:end + 1] = False
    return torch.from_numpy(mask)


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".linear_layer.weight"], sd.get(name + ".linear_layer.bias"))


def prenet(sd, prefix, x, masks):
    """model.py:124-135"""
    for j in range(2):
        x = F.relu(_lin(sd, "%s.layers.%d" % (prefix, j), x)) * masks[j] * 2.0
    return x


def _conv_bn(sd, prefix, x, pad):
    y = F.conv1d(x, sd[prefix + "0.conv.weight"], sd[prefix + "0.conv.bias"], padding=pad)
    return F.batch_norm(y, sd[prefix + "1.running_mean"], sd[prefix + "1.running_var"],
                        sd[prefix + "1.weight"], sd[prefix + "1.bias"], training=False, eps=1e-5)


def encoder(sd, hp, x, enc_masks):
    """Encoder.inference model.py:237-249.  x [1, D, Tin] -> memory [1, Tin, E]."""
    x = prenet(sd, "encoder.prenet", x.transpose(1, 2), enc_masks).transpose(1, 2)
    pad = (hp.encoder_kernel_size - 1) // 2
    for j in range(hp.encoder_n_convolutions):
        x = F.relu(_conv_bn(sd, "encoder.convolutions.%d." % j, x, pad))
    x = x.transpose(1, 2)
    E = hp.encoder_embedding_dim
    lstm = torch.nn.LSTM(E, E // 2, 1, batch_first=True, bidirectional=True)
    lstm.load_state_dict({k[len("encoder.lstm."):]: v for k, v in sd.items() if k.startswith("encoder.lstm.")})
    with torch.no_grad():
        out, _ = lstm(x)
    return out


def _lstm_cell(sd, name, x, h, c):
    g = F.linear(x, sd[name + ".weight_ih"], sd[name + ".bias_ih"]) + \
        F.linear(h, sd[name + ".weight_hh"], sd[name + ".bias_hh"])
    i, f, gg, o = g.chunk(4, 1)
    c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    return torch.sigmoid(o) * torch.tanh(c), c


def decoder(sd, hp, memory, dec_masks, max_steps=None):
    """Decoder.inference model.py:489-535 (B = 1).  Returns mel [1,nf,Tout], gate [1,Tout,1],
    align [1,Tout,Tin]."""
    B, Tin, E = memory.shape
    assert B == 1
    nf = hp.n_acoustic_feat_dims
    max_steps = max_steps or hp.max_decoder_steps
    ah = memory.new_zeros(B, hp.attention_rnn_dim)
    ac = memory.new_zeros(B, hp.attention_rnn_dim)
    dh = memory.new_zeros(B, hp.decoder_rnn_dim)
    dc = memory.new_zeros(B, hp.decoder_rnn_dim)
    w = memory.new_zeros(B, Tin)
    wcum = memory.new_zeros(B, Tin)
    ctx = memory.new_zeros(B, E)
    pm = _lin(sd, "decoder.attention_layer.memory_layer", memory)          # model.py:334
    loc_w = sd["decoder.attention_layer.location_layer.location_conv.conv.weight"]
    loc_pad = (hp.attention_location_kernel_size - 1) // 2
    x = memory.new_zeros(B, nf)                                            # go frame model.py:289-302
    mels, gates, aligns = [], [], []
    while True:
        t = len(mels)
        p = prenet(sd, "decoder.prenet", x, dec_masks[t])
        ah, ac = _lstm_cell(sd, "decoder.attention_rnn", torch.cat((p, ctx), -1), ah, ac)
        cat = torch.cat((w.unsqueeze(1), wcum.unsqueeze(1)), 1)
        pq = _lin(sd, "decoder.attention_layer.query_layer", ah.unsqueeze(1))
        pa = F.conv1d(cat, loc_w, padding=loc_pad).transpose(1, 2)
        pa = _lin(sd, "decoder.attention_layer.location_layer.location_dense", pa)
        e = _lin(sd, "decoder.attention_layer.v", torch.tanh(pq + pa + pm)).squeeze(-1)
        if hp.attention_window_size is not None:
            e = e.masked_fill(window_mask([Tin], hp.attention_window_size, t), -float("inf"))
        w = F.softmax(e, dim=1)
        ctx = torch.bmm(w.unsqueeze(1), memory).squeeze(1)
        wcum = wcum + w
        dh, dc = _lstm_cell(sd, "decoder.decoder_rnn", torch.cat((ah, ctx), -1), dh, dc)
        hc = torch.cat((dh, ctx), 1)
        mel = _lin(sd, "decoder.linear_projection", hc)
        gate = _lin(sd, "decoder.gate_layer", hc)
        mels.append(mel)
        gates.append(gate)
        aligns.append(w)
        if torch.sigmoid(gate).item() > hp.gate_threshold:                 # model.py:524-525
            break
        if len(mels) == max_steps:                                         # model.py:526-528
            break
        x = mel
    mel = torch.stack(mels).transpose(0, 1).contiguous().transpose(1, 2)
    gate = torch.stack(gates).transpose(0, 1).contiguous()
    align = torch.stack(aligns).transpose(0, 1)
    return mel, gate, align


def postnet(sd, hp, x):
    """model.py:178-184 in eval mode (dropout off)."""
    n = hp.postnet_n_convolutions
    pad = (hp.postnet_kernel_size - 1) // 2
    for j in range(n - 1):
        x = torch.tanh(_conv_bn(sd, "postnet.convolutions.%d." % j, x, pad))
    return _conv_bn(sd, "postnet.convolutions.%d." % (n - 1), x, pad)


def inference(sd, hp, ppg, enc_masks, dec_masks, max_steps=None):
    """Tacotron2.inference model.py:597-610.  ppg [1, D, Tin] ->
    [mel, mel_post, gate, align]."""
    with torch.no_grad():
        memory = encoder(sd, hp, ppg, enc_masks)
        mel, gate, align = decoder(sd, hp, memory, dec_masks, max_steps)
        mel_post = mel + postnet(sd, hp, mel)
    return [mel, mel_post, gate, align]
