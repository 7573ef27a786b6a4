"""The vocoder of ONE streamed 200-frame utterance as a stand-alone workload for rocprofv3 (the end-to-end step itself cannot run
under counter collection: PMC passes serialise kernels, and the frame collectors wait for a decoder that would be queued behind
them): per step the three seed passes of the default plan (frames [0,64), [64,128), [128,160): k_cond_seed) and
WaveGlow.infer_seeded with 160 seeded of 200 frames (k_wn_layer_mixed) -- the launches bench.py's headline times, same shapes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fac-via-ppg_amd")):
    sys.path.insert(0, p)

from facppg import synth  # noqa: E402
from waveglow.glow import WaveGlow  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda", 0)
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=256)
    m = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
    m.load_state_dict(synth.waveglow_state_dict(cfg))
    m = m.to(dev).eval()
    T = 200
    mel = synth.synthetic_mel(1, T, seed=5).to(dev)
    melp = m.mel_pad(mel)
    _, _, nb = m.seed_layout(T, dev)
    seeds = torch.empty(nb // 4, dtype=torch.float32, device=dev)
    for i in range(steps):
        for a, n in ((0, 64), (64, 64), (128, 32)):
            m.cond_seed(melp, T, a, n, seeds, block_tiles=n // 32, layers_per_workgroup=1)
        audio = m.infer_seeded(melp, T, seeds, 160, sigma=0.6, seed=i)
    torch.cuda.synchronize()
    assert torch.isfinite(audio).all()
    print("seeded workload: %d steps, launch shape %s" % (steps, m.last_launch_shape()))


if __name__ == "__main__":
    main()
