"""Oracle (TEST INFRASTRUCTURE ONLY; parity UNPINNED at the Kaldi boundary): nnet3 TDNN inference in NumPy.

Restates what the reference's compute_full_ppg (src/ppg/compute_ppg.py:42-70) gets from Kaldi's
nnet3::DecodableNnetSimple -- the network's output for every input frame, component by component and WITHOUT any
folding -- from Kaldi's published source:
  * descriptors Append / Offset / Sum / Scale                               nnet3/nnet-descriptor.cc
  * (NaturalGradient|Fixed)AffineComponent   y = W x + b                    nnet3/nnet-simple-component.cc
  * RectifiedLinearComponent, SoftmaxComponent, LogSoftmaxComponent         nnet3/nnet-simple-component.cc
  * BatchNormComponent, test mode            y = (x - mean) (var + eps)^-1/2 target_rms   nnet3/nnet-normalize-component.cc
  * NormalizeComponent                       y = x (max(sum x^2 / (D rms^2), 2^-66))^-1/2  nnet3/nnet-normalize-component.cc
  * frames requested beyond the utterance    the first / last input frame repeated        nnet3/nnet-am-decodable-simple.cc
pykaldi and the acoustic model (data/am/final.raw) are absent, so no output of Kaldi itself pins this file; what the
reference's own tests assert about the result (test/test_ppg.py:48-73: one row per frame, dim = number of senones, rows
are posteriors summing to 1, monophone reduction keeps the mass) is asserted in tests/.  The product never imports it.
"""
import numpy as np


def _component(c, x):
    t = c.type
    if t in ("NaturalGradientAffineComponent", "AffineComponent", "FixedAffineComponent"):
        return x @ c.linear.T.astype(np.float32) + c.bias.astype(np.float32)
    if t == "RectifiedLinearComponent":
        return np.maximum(x, 0.0)
    if t == "NoOpComponent":
        return x
    if t == "BatchNormComponent":
        f = c.fields
        scale = (np.float32(f.get("TargetRms", 1.0)) / np.sqrt(f["StatsVar"].astype(np.float32) + np.float32(f.get("Epsilon", 1e-3)))).astype(np.float32)
        return (x - f["StatsMean"].astype(np.float32)) * scale
    if t == "NormalizeComponent":
        rms = np.float32(c.fields.get("TargetRms", 1.0))
        d = x.shape[1]
        ss = np.maximum((x.astype(np.float32) ** 2).sum(axis=1, keepdims=True) / (d * rms * rms), np.float32(2.0 ** -66))
        return x * (1.0 / np.sqrt(ss)).astype(np.float32)
    if t in ("SoftmaxComponent", "LogSoftmaxComponent"):
        z = x - x.max(axis=1, keepdims=True)
        lse = np.log(np.exp(z).sum(axis=1, keepdims=True))
        return np.exp(z - lse) if t == "SoftmaxComponent" else z - lse
    raise ValueError("component %s" % t)


def forward(nnet, feats, output="output"):
    """feats [T, D] float32 -> the output node for frames 0..T-1, [T, K].  Every node is evaluated on the extended range
    [-L, T-1+R]; reads that leave it (they lie outside every requested frame's dependency cone) are clamped."""
    feats = np.asarray(feats, dtype=np.float32)
    T = feats.shape[0]
    L, R = nnet.context(output)
    E = L + T + R                                        # extended range: index e <-> frame e - L
    memo = {}

    def node(name):
        if name in memo:
            return memo[name]
        n = nnet.by_name[name]
        if n["kind"] == "input":
            v = feats[np.clip(np.arange(E) - L, 0, T - 1)]
        elif n["kind"] == "component":
            v = _component(nnet.components[n["component"]], desc(n["input"]))
        else:
            v = desc(n["input"])
        memo[name] = v.astype(np.float32)
        return memo[name]

    def desc(d):
        if d[0] == "node":
            return node(d[1])
        if d[0] == "Offset":
            return desc(d[1])[np.clip(np.arange(E) + d[2], 0, E - 1)]
        if d[0] == "Append":
            return np.concatenate([desc(a) for a in d[1]], axis=1)
        if d[0] == "Sum":
            return desc(d[1]) + desc(d[2])
        return np.float32(d[1]) * desc(d[2])

    return node(output)[L:L + T]
