"""Stage-by-stage timing of pipeline.synthesize for one 200-frame utterance (each stage synchronised)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import numpy as np, torch
from common.hparams import create_hparams_stage
from facppg import pipeline, synth
from script.train_ppg2mel import load_model
from waveglow.denoiser import Denoiser
from waveglow.glow import WaveGlow
import contextlib, io
hop=256
cfg = dict(synth.WAVEGLOW_CONFIG, hop_length=hop)
wg = WaveGlow.remove_weightnorm(WaveGlow(**cfg)); wg.load_state_dict(synth.waveglow_state_dict(cfg)); wg = wg.cuda().eval()
den = Denoiser(wg, hop_length=hop, mode="zeros")
hp = create_hparams_stage(max_decoder_steps=200)
taco = load_model(hp); taco.load_state_dict(synth.tacotron_state_dict(hp, gate_bias=-10.0)); taco.eval()
ppgs = [synth.synthetic_ppg(200, 5816, seed=0)]
def sync(): torch.cuda.synchronize()
with contextlib.redirect_stdout(io.StringIO()):
    for i in range(3):
        pipeline.synthesize(ppgs, taco, wg, den, seed=3, return_device=True)
    sync()
    for rep in range(2):
        t0=time.perf_counter(); x, lens = pipeline.pad_ppgs(ppgs, device="cuda"); sync(); t1=time.perf_counter()
        with torch.no_grad():
            out = taco.inference(x, seed=1); sync(); t2=time.perf_counter()
            tout = [int(v) for v in taco.last_output_lengths]; t3=time.perf_counter()
            a = wg.infer(out[1].contiguous(), sigma=0.6, seed=1); sync(); t4=time.perf_counter()
            d = den(a, strength=0.005); sync(); t5=time.perf_counter()
        print("pad %.2f taco %.2f lens %.2f wg %.2f den %.2f ms" % ((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3,(t4-t3)*1e3,(t5-t4)*1e3), file=sys.stderr)
