"""Oracle (TEST INFRASTRUCTURE ONLY): WaveGlow on the CPU, functional over a state dict.

Restates src/waveglow/glow.py of the reference (after ``remove_weightnorm``):
  * WN.forward                     glow.py:154-175
  * fused_add_tanh_sigmoid_multiply glow.py:33-40
  * Invertible1x1Conv               glow.py:82-102
  * WaveGlow.infer                  glow.py:252-293   (z injected instead of normal_())
  * WaveGlow.forward                glow.py:208-250
Weights: dict with the reference's state-dict keys.
"""
import torch
import torch.nn.functional as F


def flow_channels(cfg):
    """(n_remaining_channels, n_half) per flow, glow.py:195-206."""
    n_half, n_rem, out = cfg["n_group"] // 2, cfg["n_group"], []
    for k in range(cfg["n_flows"]):
        if k % cfg["n_early_every"] == 0 and k > 0:
            n_half -= cfg["n_early_size"] // 2
            n_rem -= cfg["n_early_size"]
        out.append((n_rem, n_half))
    return out


def wn_forward(sd, k, cfg, audio, spect):
    """glow.py:154-175 for WN[k]."""
    p = "WN.%d." % k
    nc = cfg["WN_config"]["n_channels"]
    nl = cfg["WN_config"]["n_layers"]
    ks = cfg["WN_config"]["kernel_size"]
    audio = F.conv1d(audio, sd[p + "start.weight"], sd[p + "start.bias"])
    output = None
    for i in range(nl):
        d = 2 ** i
        pad = (ks * d - d) // 2
        a = F.conv1d(audio, sd[p + "in_layers.%d.weight" % i], sd[p + "in_layers.%d.bias" % i],
                     dilation=d, padding=pad)
        b = F.conv1d(spect, sd[p + "cond_layers.%d.weight" % i], sd[p + "cond_layers.%d.bias" % i])
        in_act = a + b
        acts = torch.tanh(in_act[:, :nc]) * torch.sigmoid(in_act[:, nc:])      # glow.py:33-40
        rs = F.conv1d(acts, sd[p + "res_skip_layers.%d.weight" % i], sd[p + "res_skip_layers.%d.bias" % i])
        if i < nl - 1:
            audio = rs[:, :nc] + audio
            skip = rs[:, nc:]
        else:
            skip = rs
        output = skip if output is None else skip + output
    return F.conv1d(output, sd[p + "end.weight"], sd[p + "end.bias"])


def upsample_regroup(sd, cfg, spect, trim):
    """glow.py:253-259 (infer: trim kernel-stride) / glow.py:215-222 (forward: crop to audio)."""
    hop, g = cfg["hop_length"], cfg["n_group"]
    s = F.conv_transpose1d(spect, sd["upsample.weight"], sd["upsample.bias"], stride=hop)
    s = s[:, :, :trim]
    s = s.unfold(2, g, g).permute(0, 2, 1, 3)
    return s.contiguous().view(s.size(0), s.size(1), -1).permute(0, 2, 1)


def infer(sd, cfg, spect, sigma, z_list, alternate=False):
    """WaveGlow.infer, glow.py:252-293.  ``z_list`` = the N(0,1) draws in call order:
    [B, n_remaining, L] then one [B, n_early_size, L] per early-output flow (k = 8, then 4)."""
    T = spect.size(2)
    ksz = sd["upsample.weight"].size(2)
    total = (T - 1) * cfg["hop_length"] + ksz
    sp = upsample_regroup(sd, cfg, spect, total - (ksz - cfg["hop_length"]))
    zs = list(z_list)
    audio = sigma * zs.pop(0)
    for k in reversed(range(cfg["n_flows"])):
        n_half = audio.size(1) // 2
        if alternate and k % 2 == 1:           # legacy layout, glow_old.py:224-240: odd flows swap the halves
            a1, a0 = audio[:, :n_half], audio[:, n_half:]
        else:
            a0, a1 = audio[:, :n_half], audio[:, n_half:]
        out = wn_forward(sd, k, cfg, a0, sp)
        s, b = out[:, n_half:], out[:, :n_half]
        a1 = (a1 - b) / torch.exp(s)
        audio = torch.cat([a1, a0], 1) if (alternate and k % 2 == 1) else torch.cat([a0, a1], 1)
        W = sd["convinv.%d.conv.weight" % k].squeeze(-1)
        audio = F.conv1d(audio, W.inverse()[..., None])                         # glow.py:88-97
        if k % cfg["n_early_every"] == 0 and k > 0:
            audio = torch.cat((sigma * zs.pop(0), audio), 1)
    return audio.permute(0, 2, 1).contiguous().view(audio.size(0), -1)


def forward(sd, cfg, spect, audio):
    """WaveGlow.forward, glow.py:208-250 -> (z, log_s_list, log_det_W_list)."""
    g = cfg["n_group"]
    sp = upsample_regroup(sd, cfg, spect, audio.size(1))
    audio = audio.unfold(1, g, g).permute(0, 2, 1)
    outs, log_s_list, log_det_list = [], [], []
    for k in range(cfg["n_flows"]):
        if k % cfg["n_early_every"] == 0 and k > 0:
            outs.append(audio[:, :cfg["n_early_size"]])
            audio = audio[:, cfg["n_early_size"]:]
        W = sd["convinv.%d.conv.weight" % k].squeeze(-1)
        log_det_list.append(audio.size(0) * audio.size(2) * torch.logdet(W))
        audio = F.conv1d(audio, W[..., None])
        n_half = audio.size(1) // 2
        a0, a1 = audio[:, :n_half], audio[:, n_half:]
        out = wn_forward(sd, k, cfg, a0, sp)
        log_s, b = out[:, n_half:], out[:, :n_half]
        a1 = torch.exp(log_s) * a1 + b
        log_s_list.append(log_s)
        audio = torch.cat([a0, a1], 1)
    outs.append(audio)
    return torch.cat(outs, 1), log_s_list, log_det_list


def loss(z, log_s_list, log_det_list, sigma=1.0):
    """WaveGlowLoss, glow.py:43-59."""
    tot = torch.sum(z * z) / (2 * sigma * sigma)
    for ls, ld in zip(log_s_list, log_det_list):
        tot = tot - torch.sum(ls) - ld
    return tot / (z.size(0) * z.size(1) * z.size(2))
