#!/usr/bin/env python3
"""Throughput bench of the PPG->wav hot path on MI355X (driver contract: see the task brief).

A "step" is one WaveGlow.infer over one batch of synthetic mels = BASELINE.json configs[1]:
batch 8, mel 80x1000, fp32, noise generated on the device, inputs resident in HBM.  The metric is
BASELINE.json's: 22.05 kHz audio samples per second (hop 256), whole job over all ranks.
With --gpus N > 1 it runs one rank per GPU: either launched under torch.distributed.run (RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* in the env), or -- when WORLD_SIZE is not set -- it spawns the N
ranks itself (the one-process-per-GPU launcher pattern of the reference's distributed.py:145-170).
Utterance batches are independent, so each rank synthesises its own batch (weak scaling, no
data-path collective; RCCL is used only for the barrier and the max-over-ranks of the elapsed
time).  The world size and the number of ranks RCCL actually sees are asserted to equal --gpus.

Adds to the JSON line:
  roofline      fp32-MFMA roofline of the dominant kernel (k_wn_layer): algorithmic FLOPs per
                launch / its average launch duration measured live with hipEvents on the launch
                stream during the timed steps (facppg_wg_last_layer_ms).
  cpu_baseline  the CPU oracle (a port of the reference's PyTorch-CPU path) timed on this host's
                cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "fac-via-ppg_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BATCH, FRAMES, HOP, SR = 8, 1000, 256, 22050


def layer_flops_per_position(n_layers=8, C=256, ncond=None, edge_fold=None, n_flows=12, n_early_every=4, n_early_size=2,
                             n_group=8):
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
        ncond = 640 if unfolded else -(-1024 // HOP) * 80
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


def cpu_baseline_worker(threads, frames):
    """Oracle WaveGlow.infer (a port of the reference's PyTorch-CPU path, validated against the
    reference's golden vectors) on `threads` host cores; prints one JSON line."""
    from facppg import synth
    from oracle import waveglow as owg
    torch.set_num_threads(threads)
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=HOP)
    sd = synth.waveglow_state_dict(cfg)
    mel = synth.synthetic_mel(1, frames, seed=1234)
    zs = synth.synthetic_z(1, frames * HOP // 8, cfg, seed=4321)
    with torch.no_grad():
        owg.infer(sd, cfg, mel[:, :, :20], 0.6, [z[:, :, :20 * HOP // 8] for z in zs])   # warm-up
        t0 = time.perf_counter()
        owg.infer(sd, cfg, mel, 0.6, zs)
        t = time.perf_counter() - t0
    print(json.dumps({"threads": threads, "frames": frames, "seconds": t, "value": frames * HOP / t}))


def cpu_baseline(log):
    """Bounded CPU sample in subprocesses (a 256-thread oneDNN run of these small convs is
    pathologically slow, so a few thread counts are tried under a timeout and the best is reported)."""
    import subprocess
    frames, best = 600, None
    for threads in (16, 32):
        if threads > (os.cpu_count() or 1):
            continue
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(threads), str(frames)],
                               capture_output=True, text=True, timeout=120)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            log("cpu baseline: %d threads -> %.0f samples/s (%.1f s)" % (threads, d["value"], d["seconds"]))
            if best is None or d["value"] > best["value"]:
                best = d
        except Exception as e:  # timeout or parse failure: report what we have
            log("cpu baseline with %d threads failed: %r" % (threads, e))
    if best is None:
        return None
    return {"value": best["value"], "unit": "samples/s", "cores": best["threads"], "kind": "port",
            "host_cpus": os.cpu_count(),
            "sample": "oracle (torch-CPU fp32 port of the reference path) WaveGlow.infer B=1 mel 80x%d hop=%d = %d samples, "
                      "%.1f s on %d threads (best of 16/32 threads)" % (frames, HOP, frames * HOP, best["seconds"], best["threads"])}


def end_to_end(log):
    """The metric's own configurations (PPG -> mel -> wav: Tacotron2 + WaveGlow + Denoiser), batch=1 (its
    "real-time factor at batch=1") and BASELINE configs[2] (16 variable-length utterances), measured in a child
    process (e2e_worker) so that nothing it does -- it uses cooperative launches, which rocprofv3 on this stack
    does not survive -- can take the primary measurement down with it.  Returns None if the child fails."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--e2e-worker"], capture_output=True, text=True, timeout=600)
        out = json.loads(r.stdout.strip().splitlines()[-1])
        for k, v in out.items():
            log("end-to-end %s: %d samples in %.2f ms = %.0fx real time" % (k, v["samples"], v["ms"], v["realtime_factor"]))
        return out
    except Exception as e:   # noqa: BLE001  (secondary figure: report its absence, never fail the bench)
        log("end-to-end measurement failed: %r" % (e,))
        return None


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


def train_worker():
    import numpy as np
    from common.layers import TacotronSTFT
    from facppg import synth
    from waveglow.glow import WaveGlow, WaveGlowLoss
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = dict(synth.WAVEGLOW_CONFIG)                      # the reference's training config: hop 160, 16 kHz (config.json)
    m = WaveGlow(**cfg).to(dev).train()
    with torch.no_grad():
        for wn in m.WN:
            wn.end.weight.normal_(0, 0.02)
    stft = TacotronSTFT(1024, 160, 1024, 80, 16000, 0.0, 8000.0).to(dev)
    crit = WaveGlowLoss(0.7071)
    out = {}
    from waveglow.graphed import GraphedTrainStep
    for prec, B, graphed in (("bf16", 3, True), ("bf16", 12, True), ("bf16", 3, False), ("bf16", 12, False), ("fp32", 3, False)):
        m.train_precision = prec
        # graphed: the step replayed as one captured HIP graph, as script.train_waveglow runs it by default under bf16
        opt = torch.optim.Adam(m.parameters(), lr=1e-5, fused=True, capturable=graphed)
        g = np.random.Generator(np.random.PCG64(1))
        audio = torch.from_numpy(np.clip(g.standard_normal((B, 10000), dtype=np.float32) * 0.1, -1, 1)).to(dev)
        with torch.no_grad():
            mel = stft.mel_spectrogram(audio)
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
        peak = 2500.0 if prec == "bf16" else PEAK_F32_MFMA_TFLOPS
        out["%s_B%d%s" % (prec, B, "_graph" if graphed else "")] = {
            "workload": "WaveGlow training step (fwd + WaveGlowLoss + bwd + fused Adam), segment 10000 @16 kHz / hop 160, per-GPU batch %d, "
                        "%s MFMA operands, fp32 accumulation / master weights / gradients, 1 GPU, %s" % (
                            B, prec, "one replayed HIP graph per step" if graphed else "launch by launch"),
            "ms_per_step": t * 1e3, "samples_per_s": B * 10000 / t, "tflops": flops / t / 1e12,
            "frac_of_mfma_peak": flops / t / 1e12 / peak, "mfma_peak_tflops": peak, "loss_finite": bool(torch.isfinite(loss))}
        del stepper, opt
    print(json.dumps(out))


def stage_rooflines(stages, frames_in, frames_out, batch, hop):
    """Per-stage achieved rates next to the roofline that bounds each (SURVEY.md 8d per-unit figures): the MFMA
    stages in TFLOP/s of the fp32 MFMA peak, the streaming denoiser in GB/s of HBM peak, the autoregressive decoder as
    microseconds per frame (latency-bound: neither roofline applies)."""
    out = {}
    pos = frames_out * hop // 8
    flop = {"encoder": 22.8e6 * frames_in, "postnet": 8.68e6 * frames_out, "waveglow": 96 * layer_flops_per_position() * pos}
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


def e2e_worker():
    """PPG -> mel -> wav (Tacotron2.inference + WaveGlow.infer + Denoiser) with inputs on the host (the PPG upload is
    part of the path), timed with hipEvents on the launch stream (facppg.pipeline.StageTimer): total and per stage.
    Median of 5 after 2 warm-ups.  Prints one JSON line {config: {...}}."""
    import contextlib
    import numpy as np
    from common.hparams import create_hparams_stage
    from facppg import pipeline, synth
    from script.train_ppg2mel import load_model
    from waveglow.denoiser import Denoiser
    from waveglow.glow import WaveGlow
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=HOP)
    waveglow = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
    waveglow.load_state_dict(synth.waveglow_state_dict(cfg))
    waveglow = waveglow.to(dev).eval()
    den = Denoiser(waveglow, hop_length=HOP, mode="zeros")
    g = np.random.Generator(np.random.PCG64(7))
    configs = {"batch1": [200], "batch16_ragged": (100 + g.integers(0, 301, size=16)).tolist()}
    out = {}
    with contextlib.redirect_stdout(sys.stderr):     # the model prints the reference's "Reached max decoder steps"
        for name, lens in configs.items():
            hp = create_hparams_stage(max_decoder_steps=max(lens))
            taco = load_model(hp)
            taco.load_state_dict(synth.tacotron_state_dict(hp, gate_bias=-10.0))
            taco.eval()
            ppgs = [synth.synthetic_ppg(n, 5816, seed=i) for i, n in enumerate(lens)]
            runs = []
            for i in range(7):
                timer = pipeline.StageTimer()
                wavs, tout = pipeline.synthesize(ppgs, taco, waveglow, den, sigma=0.6, strength=0.005, seed=i, return_device=True,
                                                 step_limits=lens if len(lens) > 1 else None, timer=timer)
                runs.append(timer.stages_ms())
            runs = sorted(runs[2:], key=lambda r: r["total"])
            st = runs[len(runs) // 2]
            n = sum(tout) * HOP
            out[name] = {"workload": "PPG [%s x 5816] -> mel -> wav, hop=%d, batch=%d%s (Tacotron2 + WaveGlow + Denoiser), host PPG in, "
                                     "device wav out" % ("200" if len(lens) == 1 else "100..400", HOP, len(lens),
                                                         "" if len(lens) == 1 else " ragged, max_decoder_steps = Tin_i"),
                         "timing": "hipEvents on the launch stream, median of 5 after 2 warm-ups",
                         "ms": st["total"], "samples": n, "samples_per_s": n / (st["total"] * 1e-3),
                         "realtime_factor": n / (st["total"] * 1e-3) / SR, "frames": sum(tout),
                         "stage_ms": {k: v for k, v in st.items() if k != "total"},
                         "stage_roofline": stage_rooflines(st, sum(lens), sum(tout), len(lens), HOP)}
            del taco
    print(json.dumps(out))


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


def pmc_traffic():
    """HBM bytes per k_wn_layer launch, READ FROM the committed rocprofv3 PMC passes of this same command
    (newest profiles/rNN_pmc.json; FETCH_SIZE doubled per the gfx950 correction, calibrated on k_flow_end) -- PMC
    counters cannot be collected from inside the timed run.  Returns (bytes, provenance) or (None, None)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)["k_wn_layer"]
            return d["hbm_bytes_per_launch"], "static: %s (%s)" % (os.path.basename(path), d.get("build", "separate rocprofv3 --pmc passes"))
        except Exception:
            continue
    return None, None


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


def launch_check(args):
    """CPU-runnable check of the N-rank launch path (tests/test_bench_launch.py): rendezvous, census, barrier, one JSON
    line from rank 0 -- everything bench.py does around the timed region, without the GPU work."""
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "WORLD_SIZE %d != --gpus %d" % (world, args.gpus)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    n, devs = rank_census(dist, torch.device("cpu"), args.dist_backend)
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "world_size": dist.get_world_size(), "ranks_connected": n,
                          "max_over_ranks": float(t.item())}))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the secondary batch-1 end-to-end measurement")
    ap.add_argument("--dist-backend", default="nccl", help=argparse.SUPPRESS)   # "gloo" + --share-gpu: 1-GPU dry run of the N>1 path
    ap.add_argument("--share-gpu", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-worker", nargs=2, type=int, metavar=("THREADS", "FRAMES"), help=argparse.SUPPRESS)
    ap.add_argument("--e2e-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--train-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-train", action="store_true", help="skip the secondary training-step measurement")
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(*args.cpu_baseline_worker)
    if args.e2e_worker:
        return e2e_worker()
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
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    ranks_connected, rank_devices = rank_census(dist, dev, args.dist_backend) if dist is not None else (1, [dev.index])
    assert ranks_connected == args.gpus, "%d ranks connected, --gpus %d" % (ranks_connected, args.gpus)
    if not args.share_gpu:
        assert len(set(rank_devices)) == world, "ranks share a GPU: %s" % rank_devices

    from facppg import lib as flib, synth
    from waveglow.glow import WaveGlow
    cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=HOP)
    sd = synth.waveglow_state_dict(cfg)
    model = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    mel = synth.synthetic_mel(BATCH, FRAMES, seed=1234 + rank).to(dev)
    L = flib.load()

    def step(i):
        return model.infer(mel, sigma=0.6, seed=1000 + i)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def log(msg):
        if rank == 0:
            print("[bench %.1fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    log("model ready; warmup")
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    log("warmup done")
    handle = model._handle(dev)
    flib.check(L.facppg_wg_set_profiling(handle, 1))
    layer_ms, layer_n = [], 0
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        audio = step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    log("timed region done: %.3f s" % elapsed)
    ms = flib.ctypes.c_float()
    n = flib.ctypes.c_int()
    flib.check(L.facppg_wg_last_layer_ms(handle, flib.ctypes.byref(ms), flib.ctypes.byref(n)))
    layer_ms, layer_n = ms.value, n.value
    assert os.environ.get("FACPPG_BENCH_NO_CHECK") or torch.isfinite(audio).all()
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    samples_per_step = world * BATCH * FRAMES * HOP
    value = samples_per_step * args.steps / elapsed
    positions = BATCH * FRAMES * HOP // 8
    flops = layer_flops_per_position() * positions
    achieved = flops / (layer_ms * 1e-3) / 1e12 if layer_ms > 0 else 0.0
    # the same launch priced at the reference formulation's FLOPs (2*(3*256+640)*512 + res_skip per
    # position); it can exceed the fp32 MFMA peak because the folded kernels execute a third fewer FLOPs
    flops_ref = layer_flops_per_position(ncond=640, edge_fold=False) * positions
    achieved_ref = flops_ref / (layer_ms * 1e-3) / 1e12 if layer_ms > 0 else 0.0
    traffic, traffic_src = pmc_traffic()
    out = {
        "metric": "22.05 kHz audio samples/sec, WaveGlow.infer (mel->wav) of the PPG->wav path",
        "value": value, "unit": "samples/s", "n_gpus": world, "world_size": world, "ranks_connected": ranks_connected,
        "rank_devices": rank_devices, "collective_backend": ("RCCL (torch.distributed nccl)" if args.dist_backend == "nccl" else args.dist_backend) if world > 1 else None, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "WaveGlow.infer batch=%d, mel 80x%d, hop=%d (%d Hz), fp32, sigma=0.6, device Philox noise; "
                               "seeded synthetic weights (no checkpoints ship with the reference)" % (BATCH, FRAMES, HOP, SR),
                   "per_gpu_batch": BATCH, "global_batch": BATCH * world, "parallelism": "dp%d" % world},
        "realtime_factor": value / SR,
        "roofline": {"bound": "mfma", "kernel": "k_wn_layer", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                     "avg_launch_ms": layer_ms, "launches_timed": layer_n,
                     "flops_per_launch": flops,
                     "reference_formulation": {"flops_per_launch": flops_ref, "achieved": achieved_ref,
                                               "frac": achieved_ref / PEAK_F32_MFMA_TFLOPS}},
    }
    if rank == 0 and world == 1 and not args.no_e2e:
        e2e = end_to_end(log) or {}
        out["end_to_end_batch1"] = e2e.get("batch1")
        out["end_to_end_batch16_ragged"] = e2e.get("batch16_ragged")
        out["reference_rate_config"] = reference_rate_config(dev, mel, log)
    if rank == 0 and world == 1 and not args.no_train:
        del model
        torch.cuda.empty_cache()
        out["train_step"] = train_step(log)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(log)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
