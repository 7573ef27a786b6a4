"""Why do the postnet's small launches crawl next to a seed pass?  The one-shot postnet of a 40-frame utterance (10 launches,
~0.15 ms alone) timed on one stream while k_cond_seed runs on another, for several bounds on the pass's workgroups and with /
without non-temporal loads."""
import contextlib
import io
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fac-via-ppg_amd")):
    sys.path.insert(0, p)

from common.hparams import create_hparams_stage  # noqa: E402
from facppg import lib as _lib, synth  # noqa: E402
from script.train_ppg2mel import load_model  # noqa: E402
from waveglow.glow import WaveGlow  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    L = _lib.load()
    hp = create_hparams_stage(max_decoder_steps=200)
    with contextlib.redirect_stdout(io.StringIO()):
        taco = load_model(hp)
    taco.load_state_dict(synth.tacotron_state_dict(hp, gate_bias=-10.0))
    taco.eval()
    h = taco._handle(dev)
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=256)
    wg = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
    wg.load_state_dict(synth.waveglow_state_dict(cfg))
    wg = wg.to(dev).eval()
    T = 200
    mel = synth.synthetic_mel(1, T, seed=5).to(dev)
    melp = wg.mel_pad(mel)
    _, _, nb = wg.seed_layout(T, dev)
    seeds = torch.empty(nb // 4, dtype=torch.float32, device=dev)
    counter = torch.zeros(64, dtype=torch.int32, device=dev)
    Tp = 40
    m40 = mel[:, :, :Tp].contiguous()
    out = torch.zeros_like(m40)
    olen = torch.tensor([Tp], dtype=torch.int32, device=dev)
    ws = torch.empty(L.facppg_taco_postnet_workspace_bytes(h, 1, Tp), dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream(dev)

    def postnet():
        _lib.check(L.facppg_taco_postnet(h, _lib.ptr(m40), _lib.ptr(olen), 1, Tp, Tp, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                         _lib.current_stream(dev)))

    def timed_postnet(n=5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            postnet()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n

    for _ in range(3):
        postnet()
    torch.cuda.synchronize()
    print("postnet of %d frames alone: %.3f ms" % (Tp, timed_postnet()))
    # next to a kernel that does NOTHING (one thread spinning on the clock): is it the neighbour's work or its mere presence?
    for prio in (0, -1):
        hi = torch.cuda.Stream(dev, priority=prio)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            torch.cuda._sleep(int(2.4e9 * 0.02))
        with torch.cuda.stream(hi):
            t = timed_postnet(3)
        torch.cuda.synchronize()
        print("next to a one-thread spin kernel, postnet on a priority %d stream: %.3f ms" % (prio, t))
    with torch.cuda.stream(side):
        torch.cuda._sleep(int(2.4e9 * 0.02))
    t = timed_postnet(3)
    torch.cuda.synchronize()
    print("next to a one-thread spin kernel, postnet on the default stream: %.3f ms" % t)
    # (the round's earlier variants of this probe also ran passes that stored nothing / read one image -- a debug switch of
    #  k_cond_seed that is gone from the shipping kernel; their results are in profiles/r05_experiments.txt section 6)
    for nt in ("0:3:0", "0:3:80", "0:3:160"):
        nt, npass, ldsk = nt.split(":")
        os.environ["FACPPG_SEED_LDS"] = ldsk
        for wgs in (16, 64, 344):
            counter.zero_()
            torch.cuda.synchronize()
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(side):
                s0.record()
                for i in range(int(npass)):
                    wg.cond_seed(melp, T, 0, 32, seeds, block_tiles=1, layers_per_workgroup=1, max_workgroups=wgs, counter=counter[i:i + 1])
                s1.record()
            t = timed_postnet(3)
            torch.cuda.synchronize()
            print("lds>=%s KiB, %s pass(es) bounded to %3d workgroups (%.2f ms): postnet %.3f ms" % (ldsk, npass, wgs, s0.elapsed_time(s1), t))


if __name__ == "__main__":
    main()
