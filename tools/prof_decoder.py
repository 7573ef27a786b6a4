"""Phase-level cycle profile of the cooperative decoder (FACPPG_DECODER_PROF=1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
os.environ["FACPPG_DECODER_PROF"] = "1"
import torch
from common.hparams import create_hparams_stage
from facppg import synth, pipeline
from script.train_ppg2mel import load_model
Tin = 200
hp = create_hparams_stage(max_decoder_steps=Tin)
m = load_model(hp); m.load_state_dict(synth.tacotron_state_dict(hp)); m.eval()
x, _ = pipeline.pad_ppgs([synth.synthetic_ppg(Tin)])
x = x.cuda()
for i in range(3):
    torch.cuda.synchronize(); t = time.perf_counter(); m.inference(x, seed=1); torch.cuda.synchronize()
    print("inference %.2f ms" % ((time.perf_counter() - t) * 1e3))
