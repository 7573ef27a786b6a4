"""Round 5, step 1: the seeded layer kernel (k_wn_layer8<..., SEED>) behind k_cond_seed against the plain per-layer launches:
torch.equal on the audio, vocoder time with / without seeds, producer time per shape (block tiles x layers per workgroup)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fac-via-ppg_amd")):
    sys.path.insert(0, p)

from facppg import synth  # noqa: E402
from waveglow.glow import WaveGlow  # noqa: E402


def timed(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out


def main():
    dev = torch.device("cuda", 0)
    hop = int(os.environ.get("HOP", "256"))
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=hop)
    m = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
    m.load_state_dict(synth.waveglow_state_dict(cfg))
    m = m.to(dev).eval()
    os.environ["FACPPG_WG_PERSIST"] = "0"
    for T in [int(t) for t in os.environ.get("TS", "200,40,100,333").split(",")]:
        mel = synth.synthetic_mel(1, T, seed=5).to(dev)
        zs = [z.to(dev) for z in synth.synthetic_z(1, T * hop // 8, cfg, seed=6)]
        os.environ["FACPPG_WN_TILE"] = "32"
        t_ref, ref = timed(lambda: m.infer(mel, sigma=0.6, z=zs))
        del os.environ["FACPPG_WN_TILE"]
        melp = m.mel_pad(mel)
        tqp, mg, nb = m.seed_layout(T, dev)
        seeds = torch.empty(nb // 4, dtype=torch.float32, device=dev)
        nfr = -(-T // 32) * 32
        m.cond_seed(melp, T, 0, nfr, seeds)
        t_s, out = timed(lambda: m.infer_seeded(melp, T, seeds, nfr, sigma=0.6, z=zs))
        print("T %d: plain 32-frame tiles %.3f ms, seeded %.3f ms, equal %s, shape %s" % (T, t_ref, t_s, torch.equal(ref, out), m.last_launch_shape()))
        s_part = (T - 10) // 32 * 32
        if s_part > 0 and s_part < T:
            t_m, out_m = timed(lambda: m.infer_seeded(melp, T, seeds, s_part, sigma=0.6, z=zs))
            os.environ["FACPPG_WN_TILE"] = "16"
            t_16, out_16 = timed(lambda: m.infer(mel, sigma=0.6, z=zs))
            del os.environ["FACPPG_WN_TILE"]
            print("   mixed (%d seeded + %d unseeded frames) %.3f ms, equal %s, shape %s; plain 16-frame tiles %.3f ms, equal %s" % (
                s_part, T - s_part, t_m, torch.equal(ref, out_m), m.last_launch_shape(), t_16, torch.equal(ref, out_16)))
        if T == 200 and os.environ.get("PRODUCER"):
            for bt, lpw in ((1, 1), (1, 2), (1, 4), (1, 8), (2, 2), (2, 4), (2, 8), (3, 4), (4, 4), (4, 8)):
                seeds.fill_(float("nan"))
                t_p, _ = timed(lambda: m.cond_seed(melp, T, 0, nfr, seeds, block_tiles=bt, layers_per_workgroup=lpw))
                out2 = m.infer_seeded(melp, T, seeds, nfr, sigma=0.6, z=zs)
                t_c, _ = timed(lambda: m.cond_seed(melp, T, 32, 32 * bt, seeds, block_tiles=bt, layers_per_workgroup=lpw))
                print("  producer block_tiles %d lpw %d: all %d frames %.3f ms, one block of %d frames %.3f ms, equal %s" % (
                    bt, lpw, nfr, t_p, 32 * bt, t_c, torch.equal(ref, out2)))


if __name__ == "__main__":
    main()
