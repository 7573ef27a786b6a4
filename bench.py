#!/usr/bin/env python3
"""Throughput / latency bench of the PPG->wav hot path on MI355X (driver contract: see the task brief).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload e2e|infer|corpus|train]

The metric is BASELINE.json's: 22.05 kHz (hop 256) audio samples per second END TO END (PPG -> wav) and the real-time
factor at batch = 1.  The DEFAULT run therefore times the metric's own configuration: one 200-frame utterance per step,
host PPG [200 x 5816] in -> Tacotron2 -> WaveGlow -> Denoiser -> device wav out (SURVEY.md 8d config 1 at hop 256);
`value` = samples / s of that step, `realtime_factor` = value / 22050, `roofline` = the executed-FLOP fraction of the fp32
MFMA peak reached by the dominant kernel (the fused WaveNet layer) INSIDE that run, from hipEvents around each flow's run of its
launches in the timed steps.  The other BASELINE configs are sub-keys of the same JSON line, each with its own steps /
ms_per_step / roofline: `waveglow_batch8` (configs[1]), `end_to_end_batch16_ragged` (configs[2], one batch at a time and
software-pipelined), plus `reference_rate_config`, `train_step`, `cpu_baseline`.

--workload (what one "step" is, and what `value` counts; every workload names its BASELINE.json config):
  e2e     (default) end-to-end PPG -> mel -> wav (Tacotron2 + WaveGlow + Denoiser), host PPG in, device wav out;
          --e2e-batch 1 (default) = the metric's batch-1 case, --e2e-batch 16 = configs[2] (16 variable-length utterances per
          rank, back-to-back batches software-pipelined).  Weak scaling.
  infer   WaveGlow.infer over one batch of synthetic mels = configs[1]: batch 8, mel 80x1000, fp32, noise generated on the
          device, inputs resident in HBM (--infer-batch / --infer-frames: other shapes, e.g. 1 x 200 for profiling the
          batch-1 vocoder without the decoder's cooperative launches).  Each rank its own batch (weak scaling).
  corpus  configs[3]: offline synthesis of --utterances (1024) ragged monophone-PPG utterances sharded over the ranks
          (facppg.shard, script.synthesize_corpus), INCLUDING the all_gather of lengths and the gather of the audio to
          rank 0.  One step = the whole corpus once.  Strong scaling.
  train   configs[4]: the bf16 WaveGlow training step (fwd + loss + bwd + Adam, segment 10 000) replayed as HIP graphs,
          data parallel with the bucketed RCCL gradient all-reduce of waveglow.distributed; reports the exchange's share.

With --gpus N > 1 it runs one rank per GPU: either launched under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* in the env), or -- when WORLD_SIZE is not set -- it spawns the N ranks itself (the one-process-per-GPU launcher
pattern of the reference's distributed.py:145-170).  The world size and the number of ranks RCCL actually sees are asserted
to equal --gpus.  An N > 1 run of the default workload also carries short `train_dp` and `corpus_dp` entries (the two
configs whose collectives matter), so one 8-GPU invocation measures them too; --no-extra skips them.

  cpu_baseline  the CPU oracle (a port of the reference's PyTorch-CPU path) timed on this host's cores on a bounded sample
                (rank 0, N=1 only): end to end on the SAME config-1 input as `value`, pinned to 16 cores, 3 runs after a
                warm-up (median, min and max stated).
"""
import argparse
import contextlib
import hashlib
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "fac-via-ppg_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0
BATCH, FRAMES, HOP, SR = 8, 1000, 256, 22050
METRIC = "22.05 kHz audio samples/sec end-to-end PPG→wav; real-time factor at batch=1"   # BASELINE.json, verbatim
WG_KERNEL_SOURCE = os.path.join(ROOT, "fac-via-ppg_amd", "csrc", "facppg_wg.hip")


def kernel_source_id(path=None):
    """Identity of the WaveGlow kernel sources a measurement belongs to (csrc/facppg_wg.hip: one launch per layer;
    csrc/facppg_wgp.hip: the persistent launch; their shared header): sha1 of the files with comments and blank space
    removed.  profiles/rNN_pmc.json records it (tools/make_pmc_json.py); a PMC summary taken from another build of the
    kernels is not quoted as this build's traffic."""
    d = os.path.dirname(WG_KERNEL_SOURCE)
    paths = [path] if path else [WG_KERNEL_SOURCE, os.path.join(d, "facppg_wgp.hip"), os.path.join(d, "facppg_wg_internal.h")]
    h = hashlib.sha1()
    for q in paths:
        src = open(q).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"//[^\n]*", "", src)
        h.update(re.sub(r"\s+", "", src).encode())
    return h.hexdigest()[:16]


def layer_flops_per_position(n_layers=8, C=256, ncond=None, edge_fold=None, n_flows=12, n_early_every=4, n_early_size=2,
                             n_group=8, hop=HOP):
    """Algorithmic FLOPs of one k_wn_layer launch per group position, averaged over all layers of all flows
    (SURVEY.md Appendix D): in_layer 2*C*2C*3 + cond 2*ncond*2C + res_skip 2*C*2C (C in a flow's last layer).
    The FLOPs counted are those of the formulation that actually runs:
    * the upsampling ConvTranspose1d is folded into the conditioning conv, whose reduction shrinks from 640
      (= 80 mel x 8 group) to ceil(1024/hop)*80 = 320 rows (FACPPG_WG_UNFOLDED=1 runs, and counts, the reference's 640);
    * the flow edges are folded into the layers (FACPPG_WG_EDGE_FOLD=0 runs, and counts, the unfolded form): the end
      conv is applied through every layer's skip rows, so res_skip is 256 res rows (none in the last layer) plus
      2*n_half end rows, and the first layer's three taps act on the n_half conditioning channels + 1 through the start
      conv (K = 3*(n_half+1) instead of 768).  Zero padding of these small operands to MFMA tile sizes is NOT counted.
    ncond=640 with edge_fold=False is the reference's own formulation."""
    unfolded = os.environ.get("FACPPG_WG_UNFOLDED", "0") not in ("", "0")
    if ncond is None:
        ncond = 640 if unfolded else -(-1024 // hop) * 80
    if edge_fold is None:
        edge_fold = not unfolded and os.environ.get("FACPPG_WG_EDGE_FOLD", "1") not in ("0",)
    g1 = 2 * (3 * C + ncond) * 2 * C
    if not edge_fold:
        return ((n_layers - 1) * (g1 + 2 * C * 2 * C) + (g1 + 2 * C * C)) / n_layers
    total, hh = 0, n_group // 2
    for k in range(n_flows):
        if k % n_early_every == 0 and k > 0:
            hh -= n_early_size // 2
        for i in range(n_layers):
            taps = 2 * (3 * (hh + 1) + ncond) * 2 * C if i == 0 else g1
            res = 0 if i == n_layers - 1 else 2 * C * C
            total += taps + res + 2 * (2 * hh) * C
    return total / (n_flows * n_layers)


# ---------------------------------------------------------------------------------------------- CPU baseline (oracle)
def config3_lengths(n=16, seed=7):
    """Tin_i = 100 + PCG64(seed).integers(0, 301) frames (1-4 s)."""
    import numpy as np
    return (100 + np.random.Generator(np.random.PCG64(seed)).integers(0, 301, size=n)).tolist()


def cpu_baseline_worker(threads, mode):
    """The oracle (a port of the reference's PyTorch-CPU path, validated against the reference's golden vectors) on
    `threads` host cores; prints one JSON line.  Modes: wg600 = WaveGlow.infer alone, B = 1 x 600 frames, one run;
    e2e1 = config 1 end to end (PPG [200 x 5816] -> Tacotron2 -> WaveGlow -> Denoiser), e2e3 = the two shortest
    utterances of config 3 one after the other (the reference synthesises batch 1 only): warm-up + median of 3."""
    import numpy as np
    from common.hparams import create_hparams_stage
    from facppg import synth
    from oracle import dsp, tacotron as otac, waveglow as owg
    try:        # pin to the first `threads` CPUs this process may use (with OMP_PROC_BIND=close from the parent): round 3's
        cpus = sorted(os.sched_getaffinity(0))[:threads]      # unpinned runs moved by +-25 % from box to box
        os.sched_setaffinity(0, set(cpus))
    except (AttributeError, OSError):
        pass
    torch.set_num_threads(threads)
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=HOP)
    sd = synth.waveglow_state_dict(cfg)

    def z_for(frames, seed):
        return synth.synthetic_z(1, frames * HOP // 8, cfg, seed=seed)

    with torch.no_grad():
        owg.infer(sd, cfg, synth.synthetic_mel(1, 20, seed=1), 0.6, z_for(20, 2))   # warm-up (thread pool, oneDNN primitives)
        if mode == "wg600":
            frames = 600
            mel, zs = synth.synthetic_mel(1, frames, seed=1234), z_for(frames, 4321)
            t0 = time.perf_counter()
            owg.infer(sd, cfg, mel, 0.6, zs)
            t = time.perf_counter() - t0
            print(json.dumps({"threads": threads, "mode": mode, "seconds": t, "samples": frames * HOP, "runs": 1}))
            return
        lens = [200] if mode == "e2e1" else sorted(config3_lengths())[:2]
        hp = create_hparams_stage(max_decoder_steps=max(lens))
        tsd = synth.tacotron_state_dict(hp, gate_bias=-10.0)
        nb = 88 * HOP // 8
        bias = owg.infer(sd, cfg, torch.zeros(1, 80, 88), 0.0, [torch.zeros(1, 4, nb), torch.zeros(1, 2, nb), torch.zeros(1, 2, nb)])
        den = dsp.DenoiserOracle(bias, hop_length=HOP)
        g = np.random.Generator(np.random.PCG64(3))
        utts = []
        for i, n in enumerate(lens):
            x = torch.from_numpy(synth.synthetic_ppg(n, 5816, seed=i)).t().unsqueeze(0)
            em = torch.from_numpy((g.random((2, 1, n, hp.symbols_embedding_dim)) < 0.5).astype(np.float32))
            dm = torch.from_numpy((g.random((n, 2, 1, hp.prenet_dim)) < 0.5).astype(np.float32))
            utts.append((n, x, em, dm, z_for(n, 50 + i)))

        def run():
            out = 0
            for n, x, em, dm, zs in utts:
                hp_n = create_hparams_stage(max_decoder_steps=n)
                _, mel_post, _, _ = otac.inference(tsd, hp_n, x, em, dm)
                wav = den(owg.infer(sd, cfg, mel_post, 0.6, zs), 0.005)
                out += wav.shape[-1]
            return out
        run()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            samples = run()
            ts.append(time.perf_counter() - t0)
    print(json.dumps({"threads": threads, "mode": mode, "seconds": sorted(ts)[1], "seconds_all": ts, "samples": samples, "runs": 3, "frames": lens}))


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(log):
    """Bounded CPU sample in subprocesses (configs 1-3 end to end, median of 3 after a warm-up, core count
    and CPU model stated).  A 256-thread oneDNN run of these small convolutions is pathologically slow; the thread count is swept
    over 16 / 32 / 64 / 128 on config 1 in every run and the best is what is reported."""
    import subprocess

    def run(mode, threads, limit):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(threads), mode],
                           capture_output=True, text=True, timeout=limit,
                           env=dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="close", OMP_PLACES="cores"))
        d = json.loads(r.stdout.strip().splitlines()[-1])
        d["value"] = d["samples"] / d["seconds"]
        log("cpu baseline %s: %d threads -> %.0f samples/s (%.1f s per run)" % (mode, threads, d["value"], d["seconds"]))
        return d
    # the thread count is SWEPT on the headline's own case (config 1, ~1 s per run) and the best is reported with its count:
    # 16 threads were the best of {16, 32} in round 2; whether more of the host's cores help is measured here, every run
    ncpu = os.cpu_count() or 1
    sweep, res = {}, {}
    for t in sorted(set(min(v, ncpu) for v in (16, 32, 64, 128))):
        try:
            sweep[t] = run("e2e1", t, 240)
        except Exception as e:  # noqa: BLE001  (timeout or parse failure: report what we have)
            log("cpu baseline e2e1 with %d threads failed: %r" % (t, e))
    if sweep:
        threads = max(sweep, key=lambda t: sweep[t]["value"])
        res["e2e1"] = sweep[threads]
    else:
        threads = min(16, ncpu)
    for mode, limit in (("e2e3", 240), ("wg600", 120)):
        try:
            res[mode] = run(mode, threads, limit)
        except Exception as e:  # noqa: BLE001
            log("cpu baseline %s failed: %r" % (mode, e))
    if not res:
        return None
    head = res.get("e2e1") or res.get("e2e3") or res["wg600"]
    out = {"value": head["value"], "unit": "samples/s", "cores": threads, "kind": "port", "host_cpus": os.cpu_count(),
           "cpu_model": cpu_model_name(),
           "sample": "oracle (torch-CPU fp32 port of the reference path, pinned to reference-generated golden vectors) on %d of %d "
                     "host threads; value = config 1 end to end: PPG [200 x 5816] -> Tacotron2 -> WaveGlow -> Denoiser at hop %d = "
                     "%d samples, median of 3 runs after a warm-up (%.1f s per run)" % (
                         threads, os.cpu_count() or 0, HOP, head["samples"], head["seconds"]),
           "realtime_factor": head["value"] / SR, "pinned": "sched_setaffinity to the first %d allowed CPUs, OMP_PROC_BIND=close" % threads,
           "thread_sweep": {str(t): d["value"] for t, d in sorted(sweep.items())},
           "thread_sweep_note": "config 1 end to end at each thread count (median of 3 after a warm-up); `value` / `cores` are the best of them"}
    if head.get("seconds_all"):
        out["spread"] = {"runs_s": head["seconds_all"], "value_min": head["samples"] / max(head["seconds_all"]),
                         "value_max": head["samples"] / min(head["seconds_all"])}
    if "e2e3" in res:
        d = res["e2e3"]
        out["config3_subset"] = {"value": d["value"], "seconds": d["seconds"], "samples": d["samples"],
                                 "sample": "the two shortest utterances of config 3 (%s frames), synthesised one after the other as "
                                           "the reference does (batch 1), median of 3 after a warm-up" % d.get("frames")}
    if "wg600" in res:
        d = res["wg600"]
        out["waveglow_only"] = {"value": d["value"], "seconds": d["seconds"], "samples": d["samples"],
                                "sample": "oracle WaveGlow.infer alone, B=1, mel 80x600, hop=%d, one run after a warm-up" % HOP}
    return out


# ---------------------------------------------------------------------------------------------- secondary figures
def train_step(log):
    """BASELINE configs[4]'s step (WaveGlow fwd + loss + bwd + Adam, segment 10 000, bf16 MFMA operands) on ONE GPU at the
    reference's per-GPU batch 3 and at 12, in a child process; secondary figure.  Returns None if the child fails."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--train-worker"], capture_output=True, text=True, timeout=600)
        out = json.loads(r.stdout.strip().splitlines()[-1])
        for k, v in out.items():
            log("training step %s: %.1f ms" % (k, v["ms_per_step"]))
        return out
    except Exception as e:   # noqa: BLE001
        log("training-step measurement failed: %r" % (e,))
        return None


def train_batch(dev, B, seed=1):
    """A synthetic training batch: audio [B, 10000] ~ N(0, 0.1^2) clipped to +-1 and its mel
    from the build's own GPU mel analysis (hop 160, 16 kHz: the reference's training config, config.json)."""
    import numpy as np
    from common.layers import TacotronSTFT
    stft = TacotronSTFT(1024, 160, 1024, 80, 16000, 0.0, 8000.0).to(dev)
    g = np.random.Generator(np.random.PCG64(seed))
    audio = torch.from_numpy(np.clip(g.standard_normal((B, 10000), dtype=np.float32) * 0.1, -1, 1)).to(dev)
    with torch.no_grad():
        mel = stft.mel_spectrogram(audio)
    return mel, audio


def make_train_model(dev, prec="bf16"):
    """WaveGlow in training mode (weight-normed, non-trivial coupling: WN.end ~ N(0, 0.02^2)) and its loss."""
    from facppg import synth
    from waveglow.glow import WaveGlow, WaveGlowLoss
    torch.manual_seed(16807)
    m = WaveGlow(**dict(synth.WAVEGLOW_CONFIG)).to(dev).train()
    with torch.no_grad():
        for wn in m.WN:
            wn.end.weight.normal_(0, 0.02)
    m.train_precision = prec
    return m, WaveGlowLoss(0.7071)


def train_worker():
    from waveglow.graphed import GraphedTrainStep
    from waveglow.optim import Adam
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    out = {}
    m, crit = make_train_model(dev)
    for prec, B, graphed in (("bf16", 3, True), ("bf16", 12, True), ("bf16", 3, False), ("bf16", 12, False), ("fp32", 3, False)):
        m.train_precision = prec
        mel, audio = train_batch(dev, B)
        # graphed: the step replayed as one captured HIP graph, as script.train_waveglow runs it by default under bf16
        opt = Adam(m.parameters(), lr=1e-5)
        stepper = GraphedTrainStep(m, crit, opt, warmup=2) if graphed else None
        ts = []
        for i in range(9 if graphed else 6):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            if graphed:
                loss = stepper(mel, audio)
            else:
                m.zero_grad()
                loss = crit(m((mel, audio)))
                loss.backward()
                opt.step()
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        steady = ts[4:] if graphed else ts[1:]              # graphed: 2 warm-up steps, the capturing step, one more
        t = sorted(steady)[len(steady) // 2]
        flops = 3 * 20.26e6 * B * 10000                      # fwd 20.3 MFLOP/sample, fwd + bwd ~ 3x
        peak = PEAK_BF16_MFMA_TFLOPS if prec == "bf16" else PEAK_F32_MFMA_TFLOPS
        out["%s_B%d%s" % (prec, B, "_graph" if graphed else "")] = {
            "workload": "WaveGlow training step (fwd + WaveGlowLoss + bwd + fused Adam), segment 10000 @16 kHz / hop 160, per-GPU batch %d, "
                        "%s MFMA operands, fp32 accumulation / master weights / gradients, 1 GPU, %s" % (
                            B, prec, "one replayed HIP graph per step" if graphed else "launch by launch"),
            "ms_per_step": t * 1e3, "samples_per_s": B * 10000 / t, "tflops": flops / t / 1e12,
            "frac_of_mfma_peak": flops / t / 1e12 / peak, "mfma_peak_tflops": peak, "loss_finite": bool(torch.isfinite(loss))}
        if prec == "bf16":                                    # which launch structure the library picked for this batch (host-side query)
            from facppg import lib as _flib
            pl = _flib.load().facppg_wn_bf16_launch_plan(8, B, 10000 // 8)
            out["%s_B%d%s" % (prec, B, "_graph" if graphed else "")]["launch_plan"] = {
                "fused_forward_layer": bool(pl & 1), "fused_backward_layer": bool(pl & 2), "wgrad_tile": 256 if pl & 4 else 128, "tile_positions": pl >> 8}
        try:                                                  # (outside the timed steps) every gradient of the last step is finite
            gs = [p.grad for p in m.parameters() if p.grad is not None]
            out["%s_B%d%s" % (prec, B, "_graph" if graphed else "")]["grads_finite"] = bool(
                len(gs) > 0 and all(bool(x) for x in torch.isfinite(torch.stack(torch._foreach_norm(gs))).cpu()))
        except Exception as e:                                # never let a diagnostic take the measurement down
            out["%s_B%d%s" % (prec, B, "_graph" if graphed else "")]["grads_finite"] = "not checked: %r" % (e,)
        del stepper, opt
    print(json.dumps(out))


def stage_rooflines(stages, frames_in, frames_out, batch, hop, seeded_frames=0):
    """Per-stage achieved rates next to the roofline that bounds each (SURVEY.md 8d per-unit figures): the MFMA
    stages in TFLOP/s of the fp32 MFMA peak, the streaming denoiser in GB/s of HBM peak, the autoregressive decoder as
    microseconds per frame (latency-bound: neither roofline applies).
    seeded_frames: frames of the (single, streamed) utterance whose conditioning chunks k_cond_seed executed under the decoder
    -- those FLOPs did not run in the vocoder stage and are not credited to it."""
    out = {}
    P = hop // 8
    wg_flops = 96 * P * (seeded_frames * layer_flops_per_position(ncond=0, hop=hop) + (frames_out - seeded_frames) * layer_flops_per_position(hop=hop))
    flop = {"encoder": 22.8e6 * frames_in, "postnet": 8.68e6 * frames_out, "waveglow": wg_flops}
    for k, f in flop.items():
        if stages.get(k):
            t = f / (stages[k] * 1e-3) / 1e12
            out[k] = {"bound": "mfma", "achieved": t, "unit": "TFLOP/s", "frac": t / PEAK_F32_MFMA_TFLOPS}
    if stages.get("denoiser"):
        # compulsory 4 B in + 4 B out per sample plus the unfused STFT intermediates (frames, spectrum, frames back)
        nbytes = frames_out * hop * 8 + frames_out * (2 * 1026 + 2 * 1024) * 4
        g = nbytes / (stages["denoiser"] * 1e-3) / 1e9
        out["denoiser"] = {"bound": "hbm", "achieved": g, "unit": "GB/s", "frac": g / 8000.0}
    if stages.get("decoder"):
        out["decoder"] = {"bound": "latency", "us_per_frame": stages["decoder"] * 1e3 / max(1, frames_out // batch)}
    return out


class EndToEnd(object):
    """PPG -> mel -> wav (Tacotron2.inference + WaveGlow.infer + Denoiser) on synthetic utterances, inputs on the host (the
    PPG upload is part of the path).  SURVEY.md 8d: config 1 = one 200-frame utterance, config 3 = 16 utterances of
    Tin_i = 100 + PCG64(7).integers(0, 301) frames with max_decoder_steps = Tin_i."""

    def __init__(self, dev, lens, n_symbols=5816, seed0=0, waveglow=None, denoiser=None):
        from common.hparams import create_hparams_stage
        from facppg import synth
        from script.train_ppg2mel import load_model
        from waveglow.denoiser import Denoiser
        from waveglow.glow import WaveGlow
        self.dev, self.lens = dev, list(lens)
        if waveglow is None:
            cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=HOP)
            waveglow = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
            waveglow.load_state_dict(synth.waveglow_state_dict(cfg))
            waveglow = waveglow.to(dev).eval()
        self.waveglow = waveglow
        self.denoiser = denoiser if denoiser is not None else Denoiser(waveglow, hop_length=HOP, mode="zeros")
        with contextlib.redirect_stdout(sys.stderr):
            hp = create_hparams_stage(max_decoder_steps=max(self.lens), n_symbols=n_symbols)
            self.tacotron = load_model(hp)
            self.tacotron.load_state_dict(synth.tacotron_state_dict(hp, gate_bias=-10.0))
            self.tacotron.eval()
        self.ppgs = [synth.synthetic_ppg(n, n_symbols, seed=seed0 + i, alpha=0.002 if n_symbols > 100 else 0.1)
                     for i, n in enumerate(self.lens)]
        self.samples = sum(self.lens) * HOP

    def step(self, i, timer=None):
        from facppg import pipeline
        with contextlib.redirect_stdout(sys.stderr):     # the model prints the reference's "Reached max decoder steps"
            return pipeline.synthesize(self.ppgs, self.tacotron, self.waveglow, self.denoiser, sigma=0.6, strength=0.005, seed=i,
                                       return_device=True, step_limits=self.lens if len(self.lens) > 1 else None, timer=timer)

    def stream(self, jobs, **kw):
        """facppg.pipeline.Synthesizer.stream over this object's models (script.synthesize_corpus.synthesize_shard calls it)."""
        from facppg import pipeline
        gen = pipeline.synthesize_stream(jobs, self.tacotron, self.waveglow, self.denoiser, **kw)
        while True:
            with contextlib.redirect_stdout(sys.stderr):
                try:
                    item = next(gen)
                except StopIteration:
                    return
            yield item

    def steady_state(self, overlap=True, **kw):
        """An endless stream of this batch (seed = step number), software-pipelined unless overlap=False."""
        def jobs():
            i = 0
            while True:
                yield {"ppgs": self.ppgs, "seed": i, "step_limits": self.lens if len(self.lens) > 1 else None}
                i += 1
        return self.stream(jobs(), sigma=0.6, strength=0.005, return_device=True, overlap=overlap, **kw)

    def describe(self):
        n = len(self.lens)
        return ("PPG [%s x %d] -> mel -> wav, hop=%d (%d Hz), batch=%d%s (Tacotron2 + WaveGlow + Denoiser), fp32, host PPG in, "
                "device wav out; seeded synthetic weights" % ("200" if n == 1 else "100..400", self.ppgs[0].shape[1], HOP, SR, n,
                                                              "" if n == 1 else " ragged, max_decoder_steps = Tin_i"))


def time_end_to_end(dev, lens, steps, warmup, log, name, **kw):
    """K steps between two synchronisations on the host clock (the driver-checkable figure) + one more step under a
    hipEvent StageTimer for the per-stage breakdown."""
    from facppg import pipeline
    e = EndToEnd(dev, lens, **kw)
    for i in range(warmup):
        e.step(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        wavs, tout = e.step(warmup + i)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    timer = pipeline.StageTimer()
    e.step(warmup + steps, timer=timer)
    st = timer.stages_ms()
    assert [int(t) for t in tout] == e.lens and all(torch.isfinite(w).all() for w in wavs)
    stream_ms = None
    if len(lens) > 1:           # the same batches back to back, software-pipelined (acoustic model of i+1 under vocoder of i)
        gen = e.steady_state()
        for i in range(warmup):
            next(gen)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            next(gen)
        torch.cuda.synchronize(dev)
        stream_ms = (time.perf_counter() - t0) / steps * 1e3
        gen.close()
        log("end-to-end %s, back-to-back batches overlapped: %.2f ms/step" % (name, stream_ms))
    ms = el / steps * 1e3
    log("end-to-end %s: %d samples in %.2f ms/step = %.0fx real time" % (name, e.samples, ms, e.samples / (ms * 1e-3) / SR))
    return {"workload": e.describe(), "timing": "host clock around %d steps between device synchronisations, after %d warm-ups; "
            "stage_ms: hipEvents on the launch stream, one further step" % (steps, warmup),
            "steps": steps, "warmup": warmup, "ms_per_step": ms, "ms": ms, "samples": e.samples, "samples_per_s": e.samples / (ms * 1e-3),
            "realtime_factor": e.samples / (ms * 1e-3) / SR, "frames": sum(e.lens), "stage_ms_total": st["total"],
            **({} if stream_ms is None else {"steady_state_overlapped": {
                "what": "the same batch back to back through facppg.pipeline.synthesize_stream: step i+1's PPG upload + Tacotron2 on a second "
                        "HIP stream under step i's WaveGlow + denoiser; host clock around %d steps after %d" % (steps, warmup),
                "ms_per_step": stream_ms, "samples_per_s": e.samples / (stream_ms * 1e-3), "realtime_factor": e.samples / (stream_ms * 1e-3) / SR}}),
            "stage_ms": {k: v for k, v in st.items() if k != "total"},
            "stage_roofline": stage_rooflines(st, sum(e.lens), sum(e.lens), len(e.lens), HOP)}


def reference_rate_config(dev, mel, log):
    """Secondary figure (SURVEY.md 8d asks for both): the same WaveGlow.infer batch at the reference's own
    rate, hop 160 / 16 kHz (config.json) -- only the upsampling stride, hence the number of group positions
    per frame and the folded conditioning width, differs.  2 steps after 1 warm-up."""
    from facppg import synth
    from waveglow.glow import WaveGlow
    hop, sr = 160, 16000
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=hop)
    m = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
    m.load_state_dict(synth.waveglow_state_dict(cfg))
    m = m.to(dev).eval()
    m.infer(mel, sigma=0.6, seed=1)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(2):
        m.infer(mel, sigma=0.6, seed=2 + i)
    torch.cuda.synchronize(dev)
    t = (time.perf_counter() - t0) / 2
    n = BATCH * FRAMES * hop
    log("reference-rate config (hop 160): %.1f ms/step" % (t * 1e3))
    return {"workload": "WaveGlow.infer batch=%d, mel 80x%d, hop=%d (%d Hz)" % (BATCH, FRAMES, hop, sr), "ms_per_step": t * 1e3,
            "samples_per_s": n / t, "realtime_factor": n / t / sr}


def pmc_traffic(key="k_wn_layer"):
    """HBM bytes per launch of the dominant kernel, READ FROM the committed rocprofv3 PMC passes of this same kernel source
    (newest profiles/rNN_pmc.json; FETCH_SIZE doubled per the gfx950 correction, calibrated on k_flow_end) -- PMC counters
    cannot be collected from inside the timed run.  `key` names the entry: "k_wn_layer" = the launches of configs[1]
    (B = 8 x 1000), "k_wn_layer_b1_t200" = those of the metric's batch-1 utterance.  The summary records the identity of the
    kernel source it was taken from (tools/make_pmc_json.py); one taken from ANOTHER build of the kernels is refused (traffic
    = null and the reason in traffic_source) instead of being passed off as this build's.  Returns (bytes or None, provenance)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        try:
            with open(path) as f:
                doc = json.load(f)
            have, want = doc["k_wn_layer"].get("kernel_source_id"), kernel_source_id()
            if have != want:
                return None, "STALE: %s was taken from kernel source %s, this build is %s -- re-run tools/profile_bench.sh" % (
                    os.path.basename(path), have, want)
            if key not in doc:
                return None, "%s holds no entry %s" % (os.path.basename(path), key)
            d = doc[key]
            return d["hbm_bytes_per_launch"], "static: %s[%s] (%s)" % (os.path.basename(path), key, d.get("build", "separate rocprofv3 --pmc passes"))
        except Exception:   # noqa: BLE001
            continue
    return None, None


def wn_layer_roofline(model, layer_ms, layer_n, positions, pmc_key, flops=None, kernel="k_wn_layer"):
    """`roofline` of the fused WaveNet-layer kernel from the library's own hipEvents (facppg_wg_set_profiling: a pair around
    each flow's eight back-to-back launches on the launch stream, divided by eight -- a pair per launch costs a 78 us launch
    ~9 us of its own, and rocprofv3's per-kernel average would not agree): executed FLOPs of one launch (layer_flops_per_position x the group positions one
    launch processes) / average launch duration.  The same launch is also priced at the reference formulation's FLOPs
    (2*(3*256+640)*512 + res_skip per position), which can exceed the fp32 MFMA peak because the folded
    kernels execute a third fewer FLOPs."""
    if flops is None:
        flops = layer_flops_per_position() * positions
    achieved = flops / (layer_ms * 1e-3) / 1e12 if layer_ms > 0 else 0.0
    flops_ref = layer_flops_per_position(ncond=640, edge_fold=False) * positions
    achieved_ref = flops_ref / (layer_ms * 1e-3) / 1e12 if layer_ms > 0 else 0.0
    traffic, traffic_src = pmc_traffic(pmc_key)
    tile = model.last_launch_shape()
    return {"bound": "mfma", "kernel": kernel, "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS,
            "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
            "avg_launch_ms": layer_ms, "launches_timed": layer_n, "flops_per_launch": flops, "positions_per_launch": positions,
            "tile_frames": tile[0], "waves_per_workgroup": tile[1], "workgroups_per_launch": tile[2],
            "kernel_source_id": kernel_source_id(),
            "reference_formulation": {"flops_per_launch": flops_ref, "achieved": achieved_ref, "frac": achieved_ref / PEAK_F32_MFMA_TFLOPS}}


# ---------------------------------------------------------------------------------------------- launch plumbing
def spawn_ranks(n, argv):
    """--gpus N without a launcher: start N copies of this script, one per GPU, with the env torch.distributed.run
    would give them (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR=127.0.0.1, a free MASTER_PORT), wait for all, and pass
    rank 0's JSON line through.  Any rank failing fails the run."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # watch all ranks: one that dies must take the others down with it (they would sit in a barrier until RCCL's timeout)
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        time.sleep(0.2)
        failed = next((r for r, p in enumerate(procs) if p.poll() not in (None, 0)), None)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.kill()
    rcs = [p.wait() for p in procs]
    reader.join(5)
    out0 = "".join(c for c in chunks if c)
    if any(rcs):
        sys.stderr.write(out0)
        raise SystemExit("bench.py: rank exit codes %s" % rcs)
    sys.stdout.write(out0)
    return 0


def rank_census(dist, dev, backend):
    """The number of ranks the collective backend really connects (an all_reduce of ones: RCCL over xGMI on the GPUs)
    and each rank's device index (all_gather)."""
    on = dev if backend == "nccl" else "cpu"
    one = torch.ones(1, device=on, dtype=torch.int32)
    dist.all_reduce(one)
    mine = torch.tensor([dev.index if dev.type == "cuda" else -1], device=on, dtype=torch.int32)
    all_dev = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(all_dev, mine)
    return int(one.item()), [int(t.item()) for t in all_dev]


class FakeCorpusSynthesizer(object):
    """CPU stand-in for facppg.pipeline.Synthesizer in --launch-check (the HIP path needs a GPU): utterance i of n frames
    -> a ramp of n * HOP samples whose values encode (seed, n)."""

    def __call__(self, ppgs, utterance_seeds=None, step_limits=None, return_device=False, **kw):
        wavs = [torch.arange(p.shape[0] * HOP, dtype=torch.float32) * 1e-6 + float(s % 1000) for p, s in zip(ppgs, utterance_seeds)]
        return wavs, [p.shape[0] for p in ppgs]


def launch_check(args):
    """CPU-runnable check of the N-rank launch path (tests/test_bench_launch.py): rendezvous, census, barrier, one JSON
    line from rank 0 -- everything bench.py does around the timed region, without the GPU work.  With --workload corpus /
    train the workload's own data path runs too, on CPU stand-ins: the sharding + all_gather + padded gather of the
    corpus, the bucketed gradient exchange of the training step."""
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "WORLD_SIZE %d != --gpus %d" % (world, args.gpus)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    n, devs = rank_census(dist, torch.device("cpu"), args.dist_backend)
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out = {"launch_check": True, "workload": args.workload, "n_gpus": world, "world_size": dist.get_world_size(), "ranks_connected": n,
           "max_over_ranks": float(t.item())}
    if args.workload == "corpus":
        corpus = CorpusWorkload(None, rank, world, args, dist, synthesizer=FakeCorpusSynthesizer())
        got = corpus.step(0)
        if rank == 0:
            out["utterances"] = len(got)
            out["samples"] = int(sum(v.numel() for v in got.values()))
            out["expected_samples"] = corpus.samples
            out["lengths_ok"] = all(got[i].numel() == n_ * HOP for i, n_ in enumerate(corpus.lengths))
    if args.workload == "train":
        from waveglow.distributed import GradientExchange, broadcast_parameters
        torch.manual_seed(rank)
        m = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Tanh(), torch.nn.Linear(8, 2))
        broadcast_parameters(m, 0)
        ex = GradientExchange(m, n_buckets=2)
        m(torch.full((4, 8), float(rank + 1))).sum().backward()
        local = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
        ex.exchange()
        mean = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        out["exchange_ok"] = bool(torch.allclose(mean, sum(gathered) / world, atol=1e-6))
        out["exchange_bytes"] = ex.bytes_per_exchange()
    if rank == 0:
        print(json.dumps(out))
    dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------- workloads
class InferWorkload(object):
    """configs[1]: WaveGlow.infer batch 8 x 80x1000 per rank."""
    name, scaling, dtype = "infer", "weak", "f32"

    def __init__(self, dev, rank, world, args, dist, model=None):
        from facppg import lib as flib, synth
        from waveglow.glow import WaveGlow
        self.dev, self.world, self.flib = dev, world, flib
        if model is None:
            cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=HOP)
            model = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
            model.load_state_dict(synth.waveglow_state_dict(cfg))
            model = model.to(dev).eval()
        self.model = model
        self.B, self.T = int(getattr(args, "infer_batch", BATCH)), int(getattr(args, "infer_frames", FRAMES))
        self.mel = synth.synthetic_mel(self.B, self.T, seed=1234 + rank).to(dev)
        self.samples = world * self.B * self.T * HOP
        self.audio = None

    def step(self, i):
        self.audio = self.model.infer(self.mel, sigma=0.6, seed=1000 + i)

    def start_timed(self):
        L = self.flib.load()
        self.flib.check(L.facppg_wg_set_profiling(self.model._handle(self.dev), 1))

    def finish(self, out, elapsed, steps):
        flib, L = self.flib, self.flib.load()
        ms, n = flib.ctypes.c_float(), flib.ctypes.c_int()
        flib.check(L.facppg_wg_last_layer_ms(self.model._handle(self.dev), flib.ctypes.byref(ms), flib.ctypes.byref(n)))
        layer_ms, layer_n = ms.value, n.value
        flib.check(L.facppg_wg_set_profiling(self.model._handle(self.dev), 0))
        assert os.environ.get("FACPPG_BENCH_NO_CHECK") or torch.isfinite(self.audio).all()
        standard = (self.B, self.T) == (BATCH, FRAMES)
        out["config"] = {"workload": "%sWaveGlow.infer batch=%d, mel 80x%d, hop=%d (%d Hz), fp32, sigma=0.6, device "
                                     "Philox noise; the vocoder stage (98.8 %% of the FLOPs) of the PPG->wav path; seeded synthetic weights "
                                     "(no checkpoints ship with the reference)" % ("BASELINE configs[1]: " if standard else "", self.B, self.T, HOP, SR),
                         "per_gpu_batch": self.B, "global_batch": self.B * self.world, "parallelism": "dp%d" % self.world}
        out["roofline"] = wn_layer_roofline(self.model, layer_ms, layer_n, self.B * self.T * HOP // 8,
                                            "k_wn_layer" if standard else "k_wn_layer_b%d_t%d" % (self.B, self.T))


class E2EWorkload(object):
    """The metric's own configuration (--e2e-batch 1, the default: one 200-frame utterance per step) or configs[2]
    (--e2e-batch 16: variable-length utterances): end-to-end PPG -> wav per rank, host PPG in, device wav out."""
    name, scaling, dtype = "e2e", "weak", "f32"

    def __init__(self, dev, rank, world, args, dist, waveglow=None):
        lens = [200] if args.e2e_batch == 1 else config3_lengths(args.e2e_batch, 7 + rank)
        self.e = EndToEnd(dev, lens, seed0=1000 * rank, waveglow=waveglow)
        self.world, self.dev, self.steps = world, dev, args.steps
        # back-to-back batches, software-pipelined (facppg.pipeline.synthesize_stream): the acoustic model of step i+1 runs
        # under the vocoder of step i.  The batch-1 case is the metric's latency figure: one utterance at a time, no overlap.
        self.overlap = bool(args.e2e_overlap) and len(lens) > 1
        self.gen = self.e.steady_state() if self.overlap else None
        # every rank has the same number of utterances but its own lengths: count what was really synthesised
        n = torch.tensor([float(self.e.samples)], dtype=torch.float64)
        if dist is not None:
            n = n.to(dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(n)
        self.samples = int(n.item())

    def step(self, i):
        self.out = next(self.gen) if self.overlap else self.e.step(i)

    def start_timed(self):
        # hipEvent pairs around every flow's fused-WN-layer launches of the timed steps (created here, recorded on the launch stream)
        from facppg import lib as flib
        if len(self.e.lens) == 1:
            flib.check(flib.load().facppg_wg_set_profiling(self.e.waveglow._handle(self.dev), max(2, self.steps)))

    def finish(self, out, elapsed, steps):
        from facppg import lib as flib, pipeline
        out["config_overlap"] = ("steady state of back-to-back batches: step i+1's PPG upload + Tacotron2 run on a second HIP stream under "
                                 "step i's WaveGlow + denoiser; every step's acoustic model and vocoder are inside the timed region "
                                 "(stage_ms below: one further, un-overlapped step)") if self.overlap else "none: one batch at a time"
        b1 = len(self.e.lens) == 1
        if b1:
            L, h = flib.load(), self.e.waveglow._handle(self.dev)
            ms, n = flib.ctypes.c_float(), flib.ctypes.c_int()
            flib.check(L.facppg_wg_last_layer_ms(h, flib.ctypes.byref(ms), flib.ctypes.byref(n)))
            flib.check(L.facppg_wg_set_profiling(h, 0))
            # The streamed utterance (facppg.pipeline.ConditioningStream): the layer launches are k_wn_layer_mixed -- seeded 32-frame
            # tiles (no conditioning chunks: those FLOPs were executed by k_cond_seed under the decoder) + unseeded 16-frame
            # tiles for the frames behind the last seed pass.  Executed FLOPs of a launch are counted accordingly.
            T, P = self.e.lens[0], HOP // 8
            cs = self.e.waveglow.__dict__.get("_facppg_cond_stream")
            seeded = min(int(getattr(cs, "seeded", 0)), T) if cs is not None and os.environ.get("FACPPG_STREAM", "1") != "0" else 0
            flops = P * (seeded * layer_flops_per_position(ncond=0) + (T - seeded) * layer_flops_per_position())
            out["roofline"] = wn_layer_roofline(self.e.waveglow, ms.value, n.value, T * P, "k_wn_layer_b1_t200", flops=flops,
                                                kernel="k_wn_layer_mixed" if 0 < seeded < T else "k_wn_layer8")
            out["roofline"]["seeded_frames"] = seeded
            out["roofline"]["measured"] = ("hipEvents around each flow's 8 launches of the fused WN-layer kernel in the %d timed end-to-end steps "
                                           "(%d launches); frames [0, %d) start from the seeds k_cond_seed formed under the decoder (their "
                                           "conditioning FLOPs are that kernel's, see roofline_seed_pass), the rest run unseeded" % (steps, n.value, seeded))
        # (two untimed steps first: the roofline bookkeeping above left the GPU idle for tens of milliseconds, and a step behind idle
        #  time runs ~5 % slower than the back-to-back steps of the timed region -- measured since round 4, cause not identified)
        for i in range(2):
            self.e.step(10 ** 6 - 2 + i)
        timer = pipeline.StageTimer()
        self.e.step(10 ** 6, timer=timer)
        st = timer.stages_ms()
        cs = self.e.waveglow.__dict__.get("_facppg_cond_stream") if b1 else None
        if cs is not None:
            cs.profile = True                 # (a step of its own: the events around the seed passes must not sit in the stage timings)
            self.e.step(10 ** 6 + 1)
            torch.cuda.synchronize(self.dev)
            cs.profile = False
            passes = cs.pass_ms() if cs.pass_events else []
            if passes:
                # k_cond_seed: per pass, frames x P positions x (n_flows * wn_layers) layers x 2 * 512 * K conditioning FLOPs; every pass
                # streams all (layer, phase) weight images once (HBM-side bytes: images + the seeds it writes)
                wg = self.e.waveglow
                P, layers = HOP // 8, wg.n_flows * wg.WN[0].n_layers
                kc = -(-1024 // HOP) * 80
                flops = sum(n * P * (layers * nf // wg.n_flows) * 2.0 * 512 * kc for n, nf, _ in passes)
                byts = sum((layers * nf // wg.n_flows) * P * (512 * (-(-kc // 64) * 64) * 4.0 + n * 512 * 4.0) for n, nf, _ in passes)
                ms = sum(t for _, _, t in passes)
                n_cu = torch.cuda.get_device_properties(self.dev).multi_processor_count
                dec_wgs = self.e.tacotron.last_decoder_launch()[1]
                out["roofline_seed_pass"] = {
                    "kernel": "k_cond_seed", "bound": "mfma (64-frame blocks) / hbm (32-frame blocks): 16 FLOP per weight byte per 32 frames",
                    "passes": [{"frames": n, "flows": nf, "ms": t} for n, nf, t in passes],
                    "achieved": flops / (ms * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": flops / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                    "cus_available": n_cu - dec_wgs,
                    "frac_of_available_cus": flops / (ms * 1e-3) / 1e12 / (PEAK_F32_MFMA_TFLOPS * (n_cu - dec_wgs) / n_cu),
                    "hbm_GBps": byts / (ms * 1e-3) / 1e9, "hbm_frac": byts / (ms * 1e-3) / 1e9 / 8000.0,
                    "measured": "hipEvents around every seed pass of one further end-to-end step (not the one of stage_ms), on the seed stream, while the decoder "
                                "holds %d of the %d CUs (passes that start after the decoder's end have the chip)" % (dec_wgs, n_cu)}
        out["config"] = {"workload": "BASELINE configs[2]: " + self.e.describe() if not b1 else
                         "the metric's own case, real-time factor at batch = 1 (SURVEY.md 8d config 1 at the metric's 22.05 kHz / hop 256): "
                         + self.e.describe() + "; one utterance per step, nothing overlapped across steps",
                         "per_gpu_batch": len(self.e.lens), "global_batch": len(self.e.lens) * self.world, "parallelism": "dp%d" % self.world}
        out["stage_ms"] = {k: v for k, v in st.items() if k != "total"}
        out["stage_ms_total"] = st["total"]
        cs = self.e.waveglow.__dict__.get("_facppg_cond_stream") if b1 else None
        seeded_now = min(int(getattr(cs, "seeded", 0)), self.e.lens[0]) if cs is not None and os.environ.get("FACPPG_STREAM", "1") != "0" else 0
        out["stage_roofline"] = stage_rooflines(st, sum(self.e.lens), sum(self.e.lens), len(self.e.lens), HOP, seeded_frames=seeded_now)
        if seeded_now:
            out["stage_roofline"]["waveglow"]["executed"] = ("%d of %d frames start from seeds: their conditioning chunks ran in k_cond_seed under "
                                                             "the decoder and are not counted here" % (seeded_now, self.e.lens[0]))


class CorpusWorkload(object):
    """configs[3]: --utterances ragged monophone-PPG utterances (Tin_i = 100 + PCG64(11).integers(0, 301)
    frames, 40 symbols = config "1m") sharded over the ranks by facppg.shard.partition; every rank synthesises its shard
    in batches of --corpus-batch (64) through script.synthesize_corpus.synthesize_shard; the audio comes back to rank 0 through
    script.synthesize_corpus.collect (an all_gather of lengths + one padded gather over RCCL).  No files: the PPGs are
    generated in memory and the waveforms stay in memory on rank 0."""
    name, scaling, dtype = "corpus", "strong", "f32"

    def __init__(self, dev, rank, world, args, dist, synthesizer=None):
        import numpy as np
        from facppg import synth
        self.rank, self.world, self.dev = rank, world, dev
        self.lengths = config3_lengths(args.utterances, 11)
        self.samples = sum(self.lengths) * HOP
        self.args = argparse.Namespace(batch_size=getattr(args, "corpus_batch", 64), sigma=0.6, denoiser_strength=0.005, seed=0, limit_steps_to_input=True)
        from facppg import shard
        mine = set(shard.partition(self.lengths, world)[rank])
        nsym = 40
        # only this rank's utterances are materialised (the others are never touched by synthesize_shard)
        self.ppgs = [synth.synthetic_ppg(n, nsym, seed=5000 + i, alpha=0.1) if i in mine else np.zeros((n, 0), np.float32)
                     for i, n in enumerate(self.lengths)]
        if synthesizer is None:
            e = EndToEnd(dev, [max(self.lengths)], n_symbols=nsym)
            synthesizer = e
        self.synthesizer = synthesizer
        self.gathered = None

    def step(self, i):
        from script import synthesize_corpus as sc
        wavs, ids = sc.synthesize_shard(self.synthesizer, self.ppgs, self.lengths, self.rank, self.world, self.args)
        self.gathered = sc.collect(wavs, ids, self.world)
        return self.gathered

    def start_timed(self):
        pass

    def finish(self, out, elapsed, steps):
        if self.rank == 0:
            assert sorted(self.gathered) == list(range(len(self.lengths))), "an utterance was lost or duplicated in the gather"
            assert all(self.gathered[i].numel() == n * HOP for i, n in enumerate(self.lengths))
        out["config"] = {"workload": "BASELINE configs[3]: offline corpus synthesis, %d ragged utterances (100..400 frames, 40-symbol "
                                     "monophone PPGs) -> wav at hop=%d, sharded length-sorted round-robin over %d rank(s), batches of %d, "
                                     "per-utterance seeds and decoder limits; one step = the whole corpus incl. the all_gather of lengths "
                                     "and the padded gather of the audio to rank 0" % (len(self.lengths), HOP, self.world, self.args.batch_size),
                         "utterances": len(self.lengths), "per_gpu_utterances": -(-len(self.lengths) // self.world),
                         "global_batch": len(self.lengths), "parallelism": "dp%d" % self.world}


class TrainWorkload(object):
    """configs[4]: the bf16 WaveGlow training step, one replayed HIP graph per step, data parallel."""
    name, scaling, dtype = "train", "weak", "bf16"

    def __init__(self, dev, rank, world, args, dist):
        from waveglow.distributed import GradientExchange, broadcast_parameters
        from waveglow.graphed import GraphedTrainStep
        self.dev, self.world, self.B = dev, world, args.train_batch
        self.model, crit = make_train_model(dev, "bf16")
        self.mel, self.audio = train_batch(dev, self.B, seed=1 + rank)
        self.exchange = None
        self.baseline_ms = None
        if dist is not None:
            # what the exchange COSTS a step: the same graphed step without it, timed first on this rank (a throw-away
            # model of the same shape; 3 steps to capture, then the median of 6)
            from waveglow.optim import Adam as _Adam
            bm, bcrit = make_train_model(dev, "bf16")
            bstep = GraphedTrainStep(bm, bcrit, _Adam(bm.parameters(), lr=1e-5), warmup=2)
            ts = []
            for i in range(10):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                bstep(self.mel, self.audio)
                torch.cuda.synchronize(dev)
                ts.append(time.perf_counter() - t0)
            self.baseline_ms = sorted(ts[4:])[3] * 1e3 if bstep.graph is not None else None
            del bstep, bm, bcrit
            torch.cuda.empty_cache()
            broadcast_parameters(self.model, 0)
            self.exchange = GradientExchange(self.model, n_buckets=args.grad_buckets,
                                             grad_dtype=torch.bfloat16 if args.grad_dtype == "bf16" else None)
        from waveglow.optim import Adam
        opt = Adam(self.model.parameters(), lr=1e-5)
        self.stepper = GraphedTrainStep(self.model, crit, opt, warmup=2, exchange=self.exchange)
        self.dp = dist is not None
        self.samples = world * self.B * 10000
        self.loss = None

    def step(self, i):
        self.loss = self.stepper(self.mel, self.audio)

    def start_timed(self):
        assert self.stepper.graph is not None or self.stepper.capture_failed, "need >= 3 warm-up steps before the timed region (capture)"

    def finish(self, out, elapsed, steps):
        assert bool(torch.isfinite(self.loss))
        ms = elapsed / steps * 1e3
        flops = 3 * 20.26e6 * self.B * 10000 * self.world
        out["config"] = {"workload": "BASELINE configs[4]: WaveGlow training step (fwd + WaveGlowLoss + bwd + fused Adam), segment 10000 "
                                     "@16 kHz / hop 160, per-GPU batch %d, bf16 MFMA operands, fp32 accumulation / master weights / "
                                     "gradients, one replayed HIP graph per step%s" % (
                                         self.B, "" if not self.dp else (
                                             ", data parallel: the bucketed gradient all-reduce is PART of the graph (each bucket on the links "
                                             "under the backward pass of the earlier flows)" if self.stepper.graph_holds_step else
                                             ", data parallel: bucketed gradient all-reduce after each replay")),
                         "per_gpu_batch": self.B, "global_batch": self.B * self.world, "parallelism": "dp%d" % self.world,
                         "graph_captured": self.stepper.graph is not None}
        out["tflops"] = flops / (ms * 1e-3) / 1e12
        out["frac_of_mfma_peak"] = out["tflops"] / (PEAK_BF16_MFMA_TFLOPS * self.world)
        if self.exchange is not None:
            # the exchange on its own, after the timed region (every rank gets here): all buckets launched, then waited for
            self.exchange.exchange()
            ex_ms = self.exchange.exchange_ms()
            nbytes = self.exchange.bytes_per_exchange()
            out["gradient_exchange"] = {
                "mode": "captured in the step graph, overlapped with backward" if self.stepper.graph_holds_step else "after each replay",
                "buckets": len(self.exchange.buckets), "bytes": nbytes, "dtype": str(self.exchange.comm_dtype).replace("torch.", ""),
                "ms": ex_ms, "fraction_of_step": ex_ms / ms if ex_ms else None,
                # bus bandwidth in the NCCL-tests convention: algorithm bytes x 2 (N - 1) / N per second
                "bus_GBps": (nbytes * 2 * (self.world - 1) / self.world / (ex_ms * 1e-3) / 1e9) if ex_ms else None,
                "step_without_exchange_ms": self.baseline_ms,
                # what data parallelism adds to a step: (step with the exchange) - (the same graphed step without it, this rank)
                "exposed_ms": (ms - self.baseline_ms) if self.baseline_ms else None,
                "timing": "ms: hipEvents on the compute stream around ONE stand-alone exchange after the timed region (all_reduce of every "
                          "bucket + scale); exposed_ms: ms_per_step minus the median graphed step without any exchange"}


WORKLOADS = {"infer": InferWorkload, "e2e": E2EWorkload, "corpus": CorpusWorkload, "train": TrainWorkload}


def isolated_extra(args, rank, world, dist, log, name, steps, warmup, timeout=600):
    """A secondary workload (`train` / `corpus`) of an N > 1 headline run, measured by a second set of N ranks that rank 0
    spawns (this script with --workload <name>) while the ranks of this run wait, host-side, on the process group's store.
    Returns the entry for `<name>_dp` (rank 0) or None."""
    import datetime
    import subprocess
    store = dist.distributed_c10d._get_default_store()
    key = "facppg_%s_dp_done" % name
    entry = None
    if rank == 0:
        try:
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "GROUP_RANK",
                                                                     "LOCAL_WORLD_SIZE", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
            cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--workload", name, "--steps", str(steps),
                   "--warmup", str(warmup), "--no-extra", "--train-batch", str(args.train_batch), "--grad-buckets", str(args.grad_buckets),
                   "--grad-dtype", args.grad_dtype, "--dist-backend", args.dist_backend, "--utterances", str(args.utterances),
                   "--corpus-batch", str(args.corpus_batch)]
            if args.share_gpu:
                cmd.append("--share-gpu")
            if world == 1:
                cmd.append("--force-dist")
                env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29511")) + (17 if name == "train" else 18))
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode == 0 and lines:
                child = json.loads(lines[-1])
                entry = {k: child[k] for k in ("steps", "warmup", "ms_per_step", "value", "unit", "scaling", "dtype", "config", "tflops",
                                               "frac_of_mfma_peak", "gradient_exchange", "ranks_connected", "rank_devices") if k in child}
                entry["measured_by"] = "a second set of %d ranks spawned by rank 0 (bench.py --workload %s) while this run's ranks waited" % (world, name)
                log("%s_dp: %.1f ms/step" % (name, entry["ms_per_step"]))
            else:
                log("%s_dp: the spawned run failed (rc %s): %s" % (name, r.returncode, r.stderr[-600:]))
        except Exception as e:   # noqa: BLE001
            log("%s_dp: %r" % (name, e))
        store.set(key, "1")
    else:
        store.wait([key], datetime.timedelta(seconds=timeout + 120))
    return entry


def timed_run(wl, steps, warmup, fence, dist, dev, backend):
    for i in range(warmup):
        wl.step(i)
    torch.cuda.synchronize(dev)
    wl.start_timed()
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        wl.step(warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 20 for e2e batch 1, 5 otherwise)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default: 3 for e2e batch 1, 2 otherwise)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="e2e")
    ap.add_argument("--infer-batch", type=int, default=BATCH, help="infer workload: utterances per batch (configs[1]: 8)")
    ap.add_argument("--infer-frames", type=int, default=FRAMES, help="infer workload: mel frames per utterance (configs[1]: 1000)")
    ap.add_argument("--utterances", type=int, default=1024, help="corpus workload: utterances in the corpus")
    ap.add_argument("--corpus-batch", type=int, default=64, help="corpus workload: utterances per synthesis batch")
    ap.add_argument("--e2e-batch", type=int, default=1, help="e2e workload: utterances per rank (1 = the metric's batch-1 case, 16 = configs[2])")
    ap.add_argument("--e2e-overlap", type=int, default=1,
                    help="e2e workload, batch > 1: 1 = back-to-back steps software-pipelined (Tacotron2 of step i+1 under WaveGlow of step i), "
                         "0 = strictly one batch after the other")
    ap.add_argument("--train-batch", type=int, default=3, help="train workload: per-GPU batch (config.json: 3)")
    ap.add_argument("--grad-buckets", type=int, default=3)
    ap.add_argument("--grad-dtype", choices=("fp32", "bf16"), default="fp32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the secondary end-to-end measurements (they use cooperative launches, "
                    "which rocprofv3 on this stack does not survive)")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary training-step measurement")
    ap.add_argument("--no-extra", action="store_true", help="N > 1 infer runs: skip the train_dp / corpus_dp entries")
    ap.add_argument("--dist-backend", default="nccl", help=argparse.SUPPRESS)   # "gloo" + --share-gpu: 1-GPU dry run of the N>1 path
    ap.add_argument("--share-gpu", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--force-dist", action="store_true", help=argparse.SUPPRESS)   # a process group even at N = 1: the RCCL code paths on ONE GPU
    ap.add_argument("--cpu-baseline-worker", nargs=2, metavar=("THREADS", "MODE"), help=argparse.SUPPRESS)
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--train-worker", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    headline = args.workload == "e2e" and args.e2e_batch == 1      # the metric's own configuration
    if args.steps is None:
        args.steps = 20 if headline else 5
    if args.warmup is None:
        args.warmup = 3 if headline else 2
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(int(args.cpu_baseline_worker[0]), args.cpu_baseline_worker[1])
    if args.train_worker:
        return train_worker()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus, sys.argv[1:])
    if args.launch_check:
        return launch_check(args)

    t_start = time.perf_counter()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "WORLD_SIZE %d != --gpus %d: launch one rank per GPU (or drop WORLD_SIZE and let bench.py spawn them)" % (
        world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    if args.share_gpu:
        local_rank = 0
    assert local_rank < torch.cuda.device_count(), "rank %d has no GPU: %d visible, --gpus %d" % (rank, torch.cuda.device_count(), args.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29511")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    ranks_connected, rank_devices = rank_census(dist, dev, args.dist_backend) if dist is not None else (1, [dev.index])
    assert ranks_connected == args.gpus, "%d ranks connected, --gpus %d" % (ranks_connected, args.gpus)
    if not args.share_gpu:
        assert len(set(rank_devices)) == world, "ranks share a GPU: %s" % rank_devices

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def log(msg):
        if rank == 0:
            print("[bench %.1fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    if args.workload == "train":
        args.warmup = max(args.warmup, 4)      # 2 eager warm-ups + the capturing step + one replay before the clock starts
    wl = WORKLOADS[args.workload](dev, rank, world, args, dist)
    log("%s workload ready; warmup" % args.workload)
    elapsed = timed_run(wl, args.steps, args.warmup, fence, dist, dev, args.dist_backend)
    log("timed region done: %.3f s" % elapsed)
    value = wl.samples * args.steps / elapsed
    train = args.workload == "train"
    out = {
        "metric": METRIC if not train else "16 kHz audio samples/sec through the WaveGlow training step (BASELINE configs[4])",
        "value": value, "unit": "samples/s", "n_gpus": world, "world_size": world, "ranks_connected": ranks_connected,
        "rank_devices": rank_devices,
        "collective_backend": ("RCCL (torch.distributed nccl)" if args.dist_backend == "nccl" else args.dist_backend) if dist is not None else None,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": wl.scaling, "vs_baseline": None, "dtype": wl.dtype, "data": "synthetic", "workload": args.workload,
        "realtime_factor": value / (16000 if train else SR),
    }
    wl.finish(out, elapsed, args.steps)
    secondary_ok = rank == 0 and world == 1 and headline and not args.force_dist
    if secondary_ok and not args.no_e2e:
        # the other BASELINE configs next to the headline, each timed in this process with its own steps / ms_per_step / roofline
        try:
            sub_args = argparse.Namespace(**vars(args))
            sub_args.infer_batch, sub_args.infer_frames = BATCH, FRAMES
            sub = InferWorkload(dev, rank, world, sub_args, None, model=wl.e.waveglow)
            el = timed_run(sub, 5, 2, fence, None, dev, args.dist_backend)
            entry = {"steps": 5, "warmup": 2, "ms_per_step": el / 5 * 1e3, "value": sub.samples * 5 / el, "unit": "samples/s",
                     "realtime_factor": sub.samples * 5 / el / SR, "dtype": sub.dtype}
            sub.finish(entry, el, 5)
            out["waveglow_batch8"] = entry
            log("configs[1] WaveGlow.infer batch 8 x 1000: %.1f ms/step = %.2f M samples/s, k_wn_layer at %.3f of fp32 MFMA peak" % (
                entry["ms_per_step"], entry["value"] / 1e6, entry["roofline"]["frac"]))
            mel8 = sub.mel
            del sub
        except Exception as e:   # noqa: BLE001  (secondary figure: report its absence, never fail the bench)
            log("waveglow_batch8 failed: %r" % (e,))
            out["waveglow_batch8"], mel8 = None, None
        try:
            out["end_to_end_batch16_ragged"] = time_end_to_end(dev, config3_lengths(), 5, 2, log, "end_to_end_batch16_ragged", waveglow=wl.e.waveglow)
        except Exception as e:   # noqa: BLE001
            log("end_to_end_batch16_ragged failed: %r" % (e,))
            out["end_to_end_batch16_ragged"] = None
        if mel8 is not None:
            out["reference_rate_config"] = reference_rate_config(dev, mel8, log)
            del mel8
    if dist is not None and (headline or args.workload == "infer") and not args.no_extra:
        # the two configs whose collectives matter, measured in the same multi-GPU invocation (short runs; a failure
        # is reported as null and never takes the primary figure down)
        del wl
        torch.cuda.empty_cache()
        # train_dp and corpus_dp run in processes of their own (rank 0 starts `bench.py --gpus N --workload train|corpus`, the
        # other ranks wait on the store, GPUs idle): the training step's graph holds RCCL collectives, the corpus gather is
        # point-to-point RCCL traffic, and what can go wrong there on a stack that has never run them across GPUs -- c10d's
        # watchdog meeting a capturing stream, a collective timing out -- ABORTS a process instead of raising; the headline of
        # this run must survive that
        for key, name, steps, warm in (("train_dp", "train", 10, 4), ("corpus_dp", "corpus", 1, 1)):
            try:
                out[key] = isolated_extra(args, rank, world, dist, log, name, steps, warm)
            except Exception as e:   # noqa: BLE001
                log("%s failed: %r" % (key, e))
                out[key] = None
    if secondary_ok and not args.no_train:
        wl = None
        torch.cuda.empty_cache()
        out["train_step"] = train_step(log)
    if secondary_ok and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(log)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
